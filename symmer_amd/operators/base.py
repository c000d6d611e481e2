"""``PauliwordOp`` — drop-in for the reference class on the symplectic hot path
(``symmer/operators/base.py:33-1561``): same constructor, attributes, methods, exceptions and results for
construction, ``+ - *``, ``cleanup``, commutation/adjacency, single-Pauli rotations and GF(2) generator
routines.  The data-parallel work runs in hand-written HIP kernels (``libsymgpu.so``); host code is NumPy
glue only.  Out of scope (not on the path): ``from_matrix``, ``to_sparse_matrix``, graph colouring,
openfermion/qiskit converters, ``QuantumState``.
"""
import warnings
from copy import deepcopy
from functools import reduce, cached_property
from numbers import Number
from typing import Dict, List, Tuple, Union

import numpy as np

from .. import kernels, multi, packing
from .utils import (string_to_symplectic, symplectic_to_string, random_symplectic_matrix, check_independent,
                    cref_binary, _rref_binary, check_adjmat_noncontextual, check_jordan_independent)

warnings.simplefilter('always', UserWarning)


def _warn_large_angle(angle: float, threshold: float) -> None:
    """base.py:1156-1157: the warning belongs to the non-Clifford branch of a rotation that acts non-trivially."""
    multiple = angle * 2 / np.pi
    if abs(round(multiple) - multiple) > threshold and abs(angle) > 1e6:
        warnings.warn('Large angle can lead to precision errors: recommend using high-precision math library '
                      'such as mpmath or redefine angle in range [-pi, pi]')


class PauliwordOp:
    """Weighted sum of n-qubit Pauli strings in the symplectic representation
    (``symp_matrix`` bool ``[T, 2n]`` = ``[X | Z]``, ``coeff_vec`` complex128 ``[T]``)."""
    sigfig = 3

    def __init__(self, symp_matrix, coeff_vec) -> None:
        # validation mirrors base.py:56-72 (AssertionError on bad input; TypeError for a scalar coefficient via len())
        symp_matrix = np.asarray(symp_matrix)
        if symp_matrix.dtype == int:
            assert set(np.unique(symp_matrix)).issubset({0, 1}), 'symplectic matrix not defined with 0 and 1 only'
            symp_matrix = symp_matrix.astype(bool)
        assert symp_matrix.dtype == bool, 'Symplectic matrix must be defined over bools'
        if len(symp_matrix.shape) == 1:
            symp_matrix = symp_matrix.reshape([1, len(symp_matrix)])
        assert symp_matrix.shape[-1] % 2 == 0, 'symplectic matrix must have even number of columns'
        assert len(symp_matrix.shape) == 2, 'symplectic matrix must be 2 dimensional only'
        self._symp = symp_matrix
        self.n_qubits = symp_matrix.shape[1] // 2
        self._coeff = np.array(coeff_vec, dtype=complex)          # OUR copy (the reference aliases a complex ndarray argument): see the note below
        self._coeff_private = True       # nobody outside this object has been handed the host coefficient array
        self.n_terms = symp_matrix.shape[0]
        assert self.n_terms == len(self._coeff), 'coeff list and Pauliwords not same length'
        self._packed_cache = None
        self._dev = None                 # kernels.DeviceOp: the operator resident on the GPU (rows + coefficients)
        self._dev_coeff_valid = False    # the handle's coefficients are this operator's coefficients AND nobody outside can change them

    # ---- three layouts of one operator: the reference's bool matrix, packed rows on the host, packed rows on the device --------------
    # Results of device kernels STAY on the device (`_dev`) and come to the host when somebody asks for `symp_matrix`, `packed` or
    # `coeff_vec`; operands are uploaded once and the handle is kept, so a multi-step caller (rotate -> project -> cleanup,
    # symmer/projection/base.py:44-124) moves its operator over PCIe once in and once out.  `symp_matrix` is treated as immutable, as
    # in the reference; `coeff_vec` is not (`op.coeff_vec *= -1`, `op.coeff_vec[i] = x` are reference idioms).  Ownership is explicit
    # (`_coeff_private`): the constructor COPIES its coefficient argument (a deliberate divergence — the reference's np.asarray aliases
    # a complex ndarray, so a caller who later writes into the array he passed changes the reference's operator; here mutations reach
    # the operator through `op.coeff_vec` only), and from the moment `coeff_vec` has been handed out or assigned the device copy of
    # the coefficients is refreshed (16 bytes per term) before every device call that reads them.
    @property
    def symp_matrix(self) -> np.ndarray:
        """bool[T, 2n] = [X | Z] as in the reference; expanded to one byte per bit only when somebody asks (a 2.5e7-term, 1000-qubit
        product is 6 GB packed but 50 GB as bools) — on the device when the operator is resident there."""
        if self._symp is None:
            if self._packed_cache is None and self._dev is not None and self.n_qubits > 0:
                self._symp = self._dev.download_bool(self.n_qubits)
            else:
                self._symp = packing.unpack_rows(self.packed, self.n_qubits)
        return self._symp

    @property
    def X_block(self) -> np.ndarray:
        return self.symp_matrix[:, :self.n_qubits]

    @property
    def Z_block(self) -> np.ndarray:
        return self.symp_matrix[:, self.n_qubits:]

    @property
    def packed(self) -> np.ndarray:
        """uint64[T, 2*Wq] rows of the C-ABI on the host; cached."""
        if self._packed_cache is None:
            if self._symp is not None:
                self._packed_cache = packing.pack_rows(self._symp)
            elif self._coeff is None:
                self._packed_cache, self._coeff = self._dev.download()          # one transfer for both halves of the operator
            else:
                self._packed_cache = self._dev.download(with_coeff=False)
        return self._packed_cache

    @property
    def coeff_vec(self) -> np.ndarray:
        """complex128[T] (whatever was assigned, for subclasses that keep ints).  Handing the array out ends our knowledge of its
        contents: see the note above."""
        self._dev_coeff_valid = False
        c = self._c()
        self._coeff_private = False
        return c

    @coeff_vec.setter
    def coeff_vec(self, value) -> None:
        self._coeff = value if isinstance(value, np.ndarray) else np.asarray(value)
        self._coeff_private = False                               # the caller may keep (and write through) what he assigned
        self._dev_coeff_valid = False

    def _c(self) -> np.ndarray:
        """The coefficients for READING inside this package (never mutated, never passed on by reference)."""
        if self._coeff is None:
            self._coeff = self._dev.download_coeff()
            self._coeff_private = True
        return self._coeff

    _UPLOAD_BOOL_MIN_BYTES = 1 << 20

    def _device(self, rows_only: bool = False) -> "kernels.DeviceOp":
        """The operator as a device handle: uploaded on first use, kept for the life of the object.  ``rows_only``: the caller reads
        the rows only (commutation, Y counts, GF(2) kernels), so stale device coefficients need no refresh."""
        assert self.n_qubits > 0, 'a 0-qubit operator has no packed rows'
        if self._dev is None:
            coeff = np.asarray(self._c(), dtype=complex)
            if self._packed_cache is None and self._symp.size >= self._UPLOAD_BOOL_MIN_BYTES and self._symp.flags.c_contiguous:
                self._dev = kernels.DeviceOp.upload_bool(self._symp, coeff)       # packed by a ballot kernel: 10x np.packbits
            else:
                self._dev = kernels.DeviceOp.upload(self.packed, coeff)
            self._dev_coeff_valid = self._coeff_private
        elif not rows_only and not self._dev_coeff_valid and self._coeff is not None:
            if self._dev.shared:
                self._dev = self._dev.clone()
            self._dev.set_coeff(np.asarray(self._coeff, dtype=complex))
            self._dev_coeff_valid = self._coeff_private
        return self._dev

    def _multi_source(self, with_coeff: bool = True):
        """The operator as symmer_amd.multi takes it: the resident handle if there is one (shards then travel device to device), else
        the host arrays."""
        if self._dev is not None:
            return self._device(rows_only=not with_coeff)
        return (self.packed, np.asarray(self._c(), dtype=complex) if with_coeff else None)

    @classmethod
    def _from_device(cls, dev: "kernels.DeviceOp", n_qubits: int, n_terms: int = None) -> "PauliwordOp":
        """A kernel's result, left where it is (``n_terms``: the caller knows it already)."""
        op = cls.__new__(cls)
        op._symp = op._packed_cache = op._coeff = None
        op._coeff_private = True
        op._dev, op._dev_coeff_valid = dev, True
        op.n_qubits = n_qubits
        op.n_terms = dev.n_terms if n_terms is None else n_terms
        return op

    @classmethod
    def _from_packed(cls, packed: np.ndarray, n_qubits: int, coeff_vec) -> "PauliwordOp":
        op = cls.__new__(cls)
        packed = np.ascontiguousarray(packed, dtype='<u8')
        assert packed.ndim == 2 and packed.shape[1] == 2 * packing.words_per_block(n_qubits)
        op._symp = None
        op._packed_cache = packed
        op.n_qubits = n_qubits
        op.n_terms = packed.shape[0]
        op._coeff = np.array(coeff_vec, dtype=complex)
        op._coeff_private = True
        op._dev, op._dev_coeff_valid = None, False
        assert op.n_terms == len(op._coeff), 'coeff list and Pauliwords not same length'
        return op

    def _host_has_rows(self) -> bool:
        return self._symp is not None or self._packed_cache is not None

    def _derive(self, index=None, coeff_vec=None) -> "PauliwordOp":
        """Row selection / new coefficients without touching layouts that have not been materialised: an operator that lives on the
        device only — or is resident and large — is indexed THERE (``symgpu_op_gather``), so the selection needs no upload of its own."""
        if index is not None and self._dev is not None and (not self._host_has_rows() or self.n_terms * self.n_qubits >= (1 << 22)):
            assert coeff_vec is None
            picked = np.arange(self.n_terms)[index].astype(np.int64).reshape(-1)
            return PauliwordOp._from_device(kernels.op_gather(self._device(), picked), self.n_qubits)
        op = PauliwordOp.__new__(PauliwordOp)
        op._symp = None if self._symp is None else (self._symp if index is None else self._symp[index])
        op._packed_cache = None if self._packed_cache is None else (self._packed_cache if index is None else
                                                                    np.ascontiguousarray(self._packed_cache[index]))
        op.n_qubits = self.n_qubits
        coeff = self._c() if coeff_vec is None else coeff_vec
        op._coeff = np.array(coeff if index is None or coeff_vec is not None else coeff[index], dtype=complex)   # a copy: never an alias of ours
        op.n_terms = len(op._coeff)
        op._coeff_private = True
        op._dev, op._dev_coeff_valid = None, False
        if index is None and self._dev is not None:
            op._dev = self._dev                                   # same rows: the handle is shared, its coefficients are not ours
            self._dev.shared = True
        return op

    # ---- constructors --------------------------------------------------------------------------------
    @classmethod
    def random(cls, n_qubits: int, n_terms: int, diagonal: bool = False, complex_coeffs: bool = True,
               density: float = 0.3) -> "PauliwordOp":
        """base.py:82-107: Bernoulli(density) bits, standard-normal coefficients (same draws, same order, from NumPy's
        global generator as the reference, so seeded scripts see the same operators)."""
        bits = random_symplectic_matrix(n_qubits, n_terms, diagonal, density=density)
        coeffs = np.random.randn(n_terms) + 0j
        if complex_coeffs:
            coeffs = coeffs + 1j * np.random.randn(n_terms)
        return cls(bits, coeffs)

    @classmethod
    def from_list(cls, pauli_terms: List[str], coeff_vec: List[complex] = None) -> "PauliwordOp":
        """base.py:199-232: strings over I/X/Y/Z, all of one length; coefficients default to 1, may be (re, im) pairs."""
        count = len(pauli_terms)
        if coeff_vec is None:
            weights = np.ones(count)
        else:
            weights = np.array(coeff_vec)
            if weights.ndim == 2:
                assert weights.shape[1] == 2, 'Only tuples of size two allowed (real and imaginary components)'
                weights = weights[:, 0] + 1j * weights[:, 1]
        if count == 0:
            return cls(np.array([[]], dtype=bool), weights)
        width = len(pauli_terms[0])
        table = np.stack([string_to_symplectic(term, width) for term in pauli_terms]) if count else None
        return cls(table.astype(int), weights)

    @classmethod
    def from_dictionary(cls, operator_dict: Dict[str, complex]) -> "PauliwordOp":
        pauli_terms, coeff_vec = zip(*operator_dict.items())
        return cls.from_list(list(pauli_terms), coeff_vec)

    @classmethod
    def empty(cls, n_qubits: int) -> "PauliwordOp":
        return cls.from_dictionary({'I' * n_qubits: 0})

    # ---- printing / copying / ordering ---------------------------------------------------------------
    def __str__(self) -> str:
        fmt = f' .{self.sigfig}f'
        if self.n_qubits == 0:
            return format(self._c()[0], fmt)                      # a bare scalar
        return ' +\n'.join(f'{format(c, fmt)} {symplectic_to_string(row)}' for row, c in zip(self.symp_matrix, self._c()))

    def __repr__(self) -> str:
        return str(self)

    def copy(self) -> "PauliwordOp":
        return deepcopy(self)

    def __deepcopy__(self, memo):
        # member by member (a device handle is shared, kernels.DeviceOp.__deepcopy__) — NOT through __getstate__ below, which brings the
        # operator to the host
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for key, value in self.__dict__.items():
            if key != '_row_addr':                                # (an address into THIS object's packed rows)
                new.__dict__[key] = deepcopy(value, memo)
        return new

    # The reference's objects are plain NumPy and are pickled by its process pool (symmer/process_handler.py) and by users who save
    # results; a device handle is a pointer into THIS process.  Pickling brings the operator to the host (packed rows + coefficients,
    # 1/8 of the bool matrix) and the copy starts life without a handle.
    def __getstate__(self):
        state = {k: v for k, v in self.__dict__.items() if k not in ('_dev', '_dev_coeff_valid', '_symp', '_packed_cache', '_coeff', '_row_addr')}
        if self.n_qubits > 0:
            state['_packed_cache'] = self.packed
            state['_symp'] = None
        else:
            state['_packed_cache'], state['_symp'] = None, self.symp_matrix
        state['_coeff'] = np.array(self._c())
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self._dev, self._dev_coeff_valid, self._coeff_private = None, False, True

    def _xz_score(self, wx: int, wz: int) -> np.ndarray:
        """Per term: wx * (number of X bits) + wz * (number of Z bits)."""
        return wx * self.X_block.sum(axis=1, dtype=int) + wz * self.Z_block.sum(axis=1, dtype=int)

    def _support_order(self) -> np.ndarray:
        occupied = np.ascontiguousarray(self.X_block | self.Z_block)
        as_bytes = occupied.view(np.dtype((np.void, occupied.shape[1] * occupied.dtype.itemsize))).ravel()
        return np.argsort(as_bytes)[::-1]

    # the orderings of the reference's ``sort`` (base.py:455-492) as a table: name -> term order for key='decreasing'.  The same
    # NumPy sorts on the same score vectors as the reference, so ties fall the same way.
    _ORDERINGS = {
        'magnitude': lambda P: np.argsort(-abs(P._c())),
        'lex': lambda P: P._lex_order(),
        'weight': lambda P: np.argsort(-P.symp_matrix.sum(axis=1, dtype=int)),
        'support': lambda P: P._support_order(),
        'Z': lambda P: np.argsort(P._xz_score(P.n_qubits + 1, 1)),
        'X': lambda P: np.argsort(P._xz_score(1, P.n_qubits + 1)),
        'Y': lambda P: np.argsort(np.abs(P.X_block.astype(int) - P.Z_block.astype(int)).sum(axis=1)),
    }

    def _lex_order(self) -> np.ndarray:
        """``np.lexsort(symp_matrix.T)`` (last column = primary key) from the packed rows: column j is bit j % 64 of word j // 64 of its
        half, so the 2 Wq words taken as sort keys in storage order (last word primary) order the rows exactly as the 2n columns do."""
        if not self.n_terms:
            return np.zeros(0, dtype=int)
        if not self.n_qubits:
            return np.lexsort(self.symp_matrix.T)
        return np.lexsort(self.packed.T)

    def sort(self, by: str = 'magnitude', key: str = 'decreasing') -> "PauliwordOp":
        """Terms re-ordered by one of ``_ORDERINGS`` (``__eq__`` relies on ``'lex'``); same names and errors as base.py:455-492."""
        if by not in self._ORDERINGS:
            raise ValueError('Only permitted sort by values are magnitude, weight, X, Y or Z')
        if key not in ('increasing', 'decreasing'):
            raise ValueError('Only permitted sort by values are increasing or decreasing')
        order = self._ORDERINGS[by](self)
        return self._derive(index=order[::-1] if key == 'increasing' else order)

    def reindex(self, qubit_map) -> "PauliwordOp":
        """base.py:493-521: relabel qubits; ``{0: 2, 2: 3, 3: 0}`` or the list ``[2, 3, 0]`` (sorted values -> listed values):
        column ``old`` of the result is column ``new`` of this operator."""
        if isinstance(qubit_map, list):
            old_indices, new_indices = sorted(qubit_map), qubit_map
        elif isinstance(qubit_map, dict):
            old_indices, new_indices = zip(*qubit_map.items())
        else:
            raise TypeError('qubit_map must be a list or a dictionary')
        unmapped = set(old_indices).difference(new_indices)
        assert len(new_indices) == len(set(new_indices)), 'Duplicated index'
        assert len(unmapped) == 0, f'Assignment conflict: indices {unmapped} cannot be mapped.'
        perm = np.arange(self.n_qubits)
        perm[list(old_indices)] = list(new_indices)
        return PauliwordOp(np.hstack([self.X_block[:, perm], self.Z_block[:, perm]]), self._c().copy())

    def set_processing_method(self, method):
        """base.py:76-80 selects mp / ray / single_thread for the reference's host fan-out; the device path has no such pool."""
        if method not in ('mp', 'ray', 'single_thread'):
            raise ValueError('Invalid processing method, must be one of mp, single_thread or ray.')

    def to_dataframe(self):
        """base.py:1419-1434."""
        import pandas as pd
        frame = pd.DataFrame.from_dict({'Pauli terms': list(self.to_dictionary.keys()), 'Coefficients (real)': self._c().real})
        if np.any(self._c().imag):
            frame['Coefficients (imaginary)'] = self._c().imag
        return frame

    def conjugate_op(self, R: "PauliwordOp") -> "PauliwordOp":
        """base.py:1512-1561 is a stub in the reference as well."""
        raise NotImplementedError('not done yet. Full function at: from symmer.operators.anticommuting_op.conjugate_Pop_with_R')

    def jordan_generator_reconstruction(self, generators: "PauliwordOp"):
        """base.py:562-602: reconstruction under the Jordan product — the symmetry generators plus ONE clique of the
        pairwise anticommuting generators at a time, each through ``generator_reconstruction`` (device GF(2) kernel)."""
        assert check_jordan_independent(generators), 'The non-symmetry elements do not pairwise anticommute.'
        symmetry_mask = np.all(generators.commutes_termwise(generators), axis=1)
        if np.all(symmetry_mask):
            return self.generator_reconstruction(generators)
        reconstruction = np.zeros([self.n_terms, generators.n_terms])
        reconstructed = np.zeros(self.n_terms, dtype=bool)
        for clique in generators[~symmetry_mask].clique_cover(edge_relation='C').values():
            member = [int(np.where(np.all(generators.symp_matrix == row, axis=1))[0][0]) for row in clique.symp_matrix]
            use = symmetry_mask.copy()
            use[member] = True
            part, ok = self.generator_reconstruction(generators[use])
            rows, cols = np.ix_(ok, use)
            reconstruction[rows, cols] = part[ok]
            reconstructed |= ok
        return reconstruction.astype(int), reconstructed

    # ---- a2 ----------------------------------------------------------------------------------------------
    @cached_property
    def Y_count(self) -> np.ndarray:
        """base.py:604-615: per-term count of Pauli Y (popcount of X & Z on the packed rows, on device)."""
        if self.n_terms == 0 or self.n_qubits == 0:
            return np.zeros(self.n_terms, dtype=np.int64)
        return self._device(rows_only=True).ycount()

    # ---- a5 ----------------------------------------------------------------------------------------------
    def cleanup(self, zero_threshold: float = 1e-15) -> "PauliwordOp":
        """base.py:617-638.  Edge cases as the reference: no terms -> one identity row with coefficient 0;
        0 qubits -> the scalar term (the reference raises there, SURVEY §8a'; we return the sum)."""
        if self.n_qubits == 0:
            return PauliwordOp(np.zeros((1, 0), dtype=bool), [np.sum(self._c())])
        if self.n_terms == 0:
            return PauliwordOp(np.zeros((1, 2 * self.n_qubits), dtype=bool), [0])
        return PauliwordOp._from_device(kernels.cleanup_dev(self._device(), zero_threshold), self.n_qubits)

    def __eq__(self, Pword: "PauliwordOp") -> bool:
        """base.py:640-662: cleanup + lexicographic sort on both sides, exact rows, ``np.allclose`` coefficients."""
        check_1 = self.cleanup()
        check_2 = Pword.cleanup()
        if check_1.n_qubits != check_2.n_qubits:
            raise ValueError('Operators defined over differing numbers of qubits.')
        if check_1.n_terms != check_2.n_terms:
            return False
        if not check_1.n_qubits:
            return bool(np.allclose(check_1._c(), check_2._c()))
        # both sides in lexicographic order, compared on the packed rows (the bool matrices are never formed)
        order_1, order_2 = check_1._lex_order(), check_2._lex_order()
        return bool(np.array_equal(check_1.packed[order_1], check_2.packed[order_2]) and
                    np.allclose(check_1._c()[order_1], check_2._c()[order_2]))

    def __hash__(self) -> int:
        return hash(tuple(self.to_dictionary.items()))

    def append(self, PwordOp: "PauliwordOp") -> "PauliwordOp":
        assert self.n_qubits == PwordOp.n_qubits, 'Pauliwords defined for different number of qubits'
        if self.n_qubits and (self._dev is not None or PwordOp._dev is not None) and self.n_terms and PwordOp.n_terms:
            # an operand already lives on the device: stacked there (device-to-device), nothing comes back
            return PauliwordOp._from_device(kernels.concat_dev([self._device(), PwordOp._device()]), self.n_qubits)
        coeff = np.hstack((self._c(), PwordOp._c()))
        if (self._symp is None or PwordOp._symp is None) and self.n_qubits:
            return PauliwordOp._from_packed(np.vstack((self.packed, PwordOp.packed)), self.n_qubits, coeff)
        return PauliwordOp(np.vstack((self.symp_matrix, PwordOp.symp_matrix)), coeff)

    def __add__(self, PwordOp: "PauliwordOp") -> "PauliwordOp":
        return self.append(PwordOp).cleanup()

    def __radd__(self, add_obj) -> "PauliwordOp":
        if isinstance(add_obj, Number) and add_obj == 0:
            return self
        return self + add_obj

    def __sub__(self, PwordOp: "PauliwordOp") -> "PauliwordOp":
        return self + PwordOp.multiply_by_constant(-1)        # base.py:742-748 negates a copy's coefficients: the same values

    def _scaled(self, const: complex, conjugate_first: bool = False) -> "PauliwordOp":
        """coefficients -> (conj?) * const with the rows untouched; on the device when the coefficients live there only."""
        if self._dev is not None and self._coeff is None:
            scaled = self._dev.clone()
            scaled.scale(const, conjugate_first)
            return PauliwordOp._from_device(scaled, self.n_qubits)
        coeff = self._c().conjugate() if conjugate_first else self._c()
        return self._derive(coeff_vec=coeff if (conjugate_first and const == 1) else coeff * const)

    def multiply_by_constant(self, const: complex) -> "PauliwordOp":
        return self._scaled(const)

    # ---- a3 / a4 -----------------------------------------------------------------------------------------
    def _product(self, other: "PauliwordOp", self_is_inner: bool, zero_threshold: float) -> "PauliwordOp":
        """Fused product + cleanup ``self * other`` on resident operands; the result stays resident.  ``self_is_inner``: self's terms
        are the fast index of the reference's pair order (base.py:783-792)."""
        if self.n_terms == 0 or other.n_terms == 0:
            return PauliwordOp._from_packed(np.empty((0, 2 * packing.words_per_block(self.n_qubits)), dtype='<u8'), self.n_qubits, [])
        n_pairs = self.n_terms * other.n_terms
        if (n_pairs >= multi.MIN_PAIRS_PRODUCT
                and multi.product_uses_devices(n_pairs, 16 * packing.words_per_block(self.n_qubits) + 16, kernels.device_memory_bytes())
                and multi.group() is not None):
            # on request (SYMGPU_DEVICES_PRODUCT=1) or when one MI355X cannot hold the product: the outer index in contiguous blocks over the
            # devices of this process (symmer_amd/multi.py)
            inner, outer = (self, other) if self_is_inner else (other, self)
            res = multi.group().mul_cleanup(inner._multi_source(), None if other is self else outer._multi_source(), self_is_inner, zero_threshold,
                                            same=other is self)
            return PauliwordOp._from_device(res, self.n_qubits)
        a = self._device()
        b = a if other is self else other._device()
        inner, outer = (a, b) if self_is_inner else (b, a)
        return PauliwordOp._from_device(kernels.mul_cleanup_handles(inner, outer, self_is_inner, zero_threshold), self.n_qubits)

    def _multiply_by_operator(self, PwordOp: "PauliwordOp", zero_threshold: float = 1e-15) -> "PauliwordOp":
        """base.py:764-794: ``self`` is the inner (fast) index and the LEFT factor; fused product + cleanup on device."""
        assert self.n_qubits == PwordOp.n_qubits, 'PauliwordOps defined for different number of qubits'
        return self._product(PwordOp, True, zero_threshold)

    def __mul__(self, mul_obj, zero_threshold: float = 1e-15) -> "PauliwordOp":
        """base.py:821-859.  The operand with fewer terms is the outer index; the reference does that through
        ``(B^+ A^+)^+`` (base.py:847-849), which equals the direct phase with the roles swapped."""
        if isinstance(mul_obj, Number):
            return self.multiply_by_constant(mul_obj)
        from .quantum_state import QuantumState
        is_state = isinstance(mul_obj, QuantumState)
        if is_state:
            # applying an operator to a ket == multiplying by its state_op (|0> -> Z, |1> -> X), base.py:838-842
            assert mul_obj.vec_type == 'ket', 'cannot multiply a bra from the left'
            other = mul_obj.state_op
        else:
            other = mul_obj
        assert isinstance(other, PauliwordOp), f'cannot multiply PauliwordOp by {type(mul_obj)}'
        assert self.n_qubits == other.n_qubits, 'PauliwordOps defined for different number of qubits'
        out = self._product(other, self.n_terms >= other.n_terms, zero_threshold)
        if is_state:
            # identities were mapped to Z: II == ZZ as states, so fold i^Y into the coefficients and merge again (base.py:854-857)
            return QuantumState(out.X_block.astype(int), out._c() * (1j ** out.Y_count)).cleanup()
        return out

    def expval(self, psi) -> complex:
        """base.py:796-819: <psi|H|psi>; many-term operators use one bra x op x ket product, otherwise term by term."""
        from .quantum_state import single_term_expval
        if self.n_terms > psi.n_terms and psi.n_terms > 10:
            return (psi.dagger * self * psi).real
        expvals = np.array([single_term_expval(P, psi) for P in self]) if self.n_terms > 1 else np.array(single_term_expval(self, psi))
        return np.sum(expvals * self._c()).real

    def __rmul__(self, const):
        if isinstance(const, Number):
            return self.multiply_by_constant(const)
        return NotImplemented

    def __imul__(self, PwordOp: "PauliwordOp") -> "PauliwordOp":
        return self.__mul__(PwordOp)

    def __pow__(self, exponent: int) -> "PauliwordOp":
        assert isinstance(exponent, int), 'the exponent is not an integer'
        if exponent == 0:
            return PauliwordOp.from_list(['I' * self.n_qubits], [1])
        return reduce(lambda x, y: x * y, [self.copy()] * exponent)

    def __getitem__(self, key) -> "PauliwordOp":
        if isinstance(key, (int, np.integer)):
            key = int(key)
            if key < 0:
                key += self.n_terms
            assert key < self.n_terms, 'Index out of range'
            mask = [key]
        elif isinstance(key, slice):
            start, stop = key.start, key.stop
            if start is None:
                start = 0
            if stop is None:
                stop = self.n_terms
            mask = np.arange(start, stop, key.step)
        elif isinstance(key, (list, np.ndarray)):
            mask = np.asarray(key)
        else:
            raise ValueError(f'Unrecognised input {type(key)}, must be an integer, slice, list or np.array')
        return self._derive(index=mask)

    def __iter__(self):
        return iter([self[i] for i in range(self.n_terms)])

    # ---- a6 ----------------------------------------------------------------------------------------------
    def commutes_termwise(self, PwordOp: "PauliwordOp") -> np.ndarray:
        """base.py:938-971: bool ``[N, M]``, True where terms commute."""
        assert self.n_qubits == PwordOp.n_qubits, 'Pauliwords defined for different number of qubits'
        if self.n_qubits == 0:
            return np.ones((self.n_terms, PwordOp.n_terms), dtype=bool)
        if self.n_terms == 0 or PwordOp.n_terms == 0:
            return np.empty((self.n_terms, PwordOp.n_terms), dtype=bool)
        if self.n_terms * PwordOp.n_terms >= multi.MIN_PAIRS_COMMUTES and multi.group() is not None:
            # more than one MI355X in this process: the left term axis in contiguous blocks over the devices, the right operand all-gathered
            # over xGMI, the blocks of the table read back over every device's own PCIe link (symmer_amd/multi.py)
            return multi.group().commutes(self._multi_source(False), None if PwordOp is self else PwordOp._multi_source(False))
        a = self._device(rows_only=True)
        return kernels.commutes_handles(a, a if PwordOp is self else PwordOp._device(rows_only=True))

    def anticommutes_termwise(self, PwordOp: "PauliwordOp") -> np.ndarray:
        return ~self.commutes_termwise(PwordOp)

    @cached_property
    def adjacency_matrix(self) -> np.ndarray:
        return self.commutes_termwise(self)

    @cached_property
    def is_noncontextual(self) -> bool:
        """base.py:1074-1088 / utils.py:567-589, all of it on the device (csrc/project.hip): bit-packed adjacency rows, the rows of the
        terms that do not commute with everything restricted to those terms, unique rows by the cleanup kernels, disjointness of the
        unique rows as a popcount identity — the M x M matrix never exists as bytes and never leaves the GPU."""
        if self.n_terms < 4:
            return True
        return kernels.noncontextual_dev(self._device(rows_only=True))

    # ---- graph glue on the device-computed adjacency matrix (reference base.py:985-1364; networkx, host) ------------
    def qubitwise_commutes_termwise(self, PwordOp: "PauliwordOp") -> np.ndarray:
        """base.py:985-1009: True where terms commute qubit by qubit (host NumPy; not a hot-path kernel)."""
        assert self.n_qubits == PwordOp.n_qubits, 'Pauliwords defined for different number of qubits'
        xa, za = self.X_block[:, None, :], self.Z_block[:, None, :]
        xb, zb = PwordOp.X_block[None, :, :], PwordOp.Z_block[None, :, :]
        both = (xa | za) & (xb | zb)
        return np.all(~both | ((xa == xb) & (za == zb)), axis=2)

    @cached_property
    def adjacency_matrix_qwc(self) -> np.ndarray:
        return self.qubitwise_commutes_termwise(self)

    def get_graph(self, edge_relation: str = 'C', label_nodes: bool = False):
        """base.py:1192-1230: networkx graph whose edges join terms that commute ('C'), anticommute ('AC') or commute
        qubit-wise ('QWC'); the relation matrices come from the device kernel."""
        import networkx as nx
        relations = {'C': lambda: self.adjacency_matrix, 'AC': lambda: ~self.adjacency_matrix, 'QWC': lambda: self.adjacency_matrix_qwc}
        if edge_relation not in relations:
            raise TypeError('Unrecognised edge relation, must be one of C (commuting), AC (anticommuting) or QWC (qubitwise commuting).')
        edges = np.array(relations[edge_relation](), dtype=bool, copy=True)
        np.fill_diagonal(edges, False)                            # no self loops
        graph = nx.from_numpy_array(edges)
        if label_nodes:
            graph = nx.relabel_nodes(graph, {k: symplectic_to_string(row) for k, row in enumerate(self.symp_matrix)})
        return graph

    def largest_clique(self, edge_relation: str = 'C') -> "PauliwordOp":
        import networkx as nx
        graph = self.get_graph(edge_relation=edge_relation)
        pauli_indices = sorted(nx.find_cliques(graph), key=lambda x: -len(x))[0]
        return sum([self[i] for i in pauli_indices])

    def clique_cover(self, edge_relation: str = 'C', strategy: str = 'largest_first', colouring_interchange: bool = False
                     ) -> Dict[int, "PauliwordOp"]:
        """base.py:1266-1364: clique partition by greedy colouring of the complement graph, or 'sorted_insertion'."""
        if strategy == 'sorted_insertion':
            if colouring_interchange is not False:
                warnings.warn(f'{strategy} is not a graph colouring method, so colouring_interchange flag is ignored')
            ops = list(self.sort(by='magnitude', key='decreasing'))
            check = {'C': lambda x, y: np.all(x.commutes_termwise(y)), 'AC': lambda x, y: np.all(~x.commutes_termwise(y)),
                     'QWC': lambda x, y: np.all(x.qubitwise_commutes_termwise(y))}[edge_relation]
            cliques = {0: ops[0]}
            for op in ops[1:]:
                for key in cliques:
                    if check(op, cliques[key]):
                        cliques[key] += op
                        break
                else:
                    cliques[len(cliques)] = op
            return cliques
        import networkx as nx
        graph = self.get_graph(edge_relation=edge_relation)
        col_map = nx.greedy_color(nx.complement(graph), strategy=strategy, interchange=colouring_interchange)
        cliques = {}
        for p_index, colour in col_map.items():
            cliques[colour] = cliques.get(colour, PauliwordOp.from_list(['I' * self.n_qubits], [0])) + self[p_index]
        return cliques

    def commutator(self, PwordOp: "PauliwordOp") -> "PauliwordOp":
        return self * PwordOp - PwordOp * self

    def anticommutator(self, PwordOp: "PauliwordOp") -> "PauliwordOp":
        return self * PwordOp + PwordOp * self

    def commutes(self, PwordOp: "PauliwordOp") -> bool:
        commutator = self.commutator(PwordOp).cleanup()
        return bool(commutator.n_terms == 0 or np.all(commutator._c()[0] == 0))

    # ---- a7 ----------------------------------------------------------------------------------------------
    def _rotate_by_single_Pword(self, Pword: "PauliwordOp", angle: float = None, threshold: float = 1e-18
                                ) -> "PauliwordOp":
        """base.py:1090-1161 as one fused device pass; ``threshold`` decides Clifford vs non-Clifford as in base.py:1146.
        Operators with duplicate rows are detected on the device and take the merging path (``csrc/rotate.hip``)."""
        if angle is None:
            angle = np.pi / 2
        if angle.imag != 0:
            warnings.warn('Complex component in angle: this will be ignored.')
        angle = angle.real
        assert Pword.n_terms == 1, 'Only rotation by single Pauliword allowed here'
        assert Pword.n_qubits == self.n_qubits, 'Pauliwords defined for different number of qubits'
        if Pword._c()[0] != 1:
            warnings.warn(f'Pword coefficient {Pword._c()[0]: .8f} has been set to 1')
        if self.n_terms == 0:
            return self
        q_addr = Pword.__dict__.get('_row_addr')                # the generator's packed row, pinned by its owner: a Trotter circuit reuses its generators
        if q_addr is None:
            q_addr = Pword.__dict__['_row_addr'] = Pword.packed.ctypes.data
        res, all_commute, n_out = kernels.rotate_single_resident(self._device(), q_addr, angle, 1e-15, threshold)
        if all_commute:
            return self                                         # identity action: the SAME object (base.py:1131-1133)
        if abs(angle) > 1e6:
            _warn_large_angle(angle, threshold)                 # only on the non-Clifford, non-commuting branch (base.py:1156-1157)
        if n_out == 0 and kernels.rotation_args(angle, threshold)[2] < 0 and not np.any(self.commutes_termwise(Pword)):
            # non-Clifford and nothing left: the reference returns `commute_self + anticom_part` (base.py:1159-1161), and the sum of
            # two operators without terms is 0 * I (append, then cleanup(): base.py:631-632)
            return PauliwordOp(np.zeros((1, 2 * self.n_qubits), dtype=bool), [0])
        return PauliwordOp._from_device(res, self.n_qubits, n_out)     # the result stays on the device until somebody reads it

    def perform_rotations(self, rotations: List[Tuple["PauliwordOp", float]]) -> "PauliwordOp":
        """base.py:1163-1186: rotations applied left to right, each followed by ``cleanup()``; the operator stays
        device-resident across the whole chain (one upload, one download)."""
        if rotations == []:
            return self.copy().cleanup()
        wq = packing.words_per_block(self.n_qubits)
        # pack the generators of the whole sequence in ONE vectorised call (a depth-2,000 circuit handed over as bool matrices spent
        # 10 ms packing 2,000 single rows one by one: more than the 10 ms of its device run)
        fresh = [r for r, _ in rotations if isinstance(r, PauliwordOp) and r._packed_cache is None and r._symp is not None
                 and r.n_terms == 1 and r.n_qubits == self.n_qubits]
        if len(fresh) > 4:
            for r, row in zip(fresh, packing.pack_rows(np.concatenate([r._symp for r in fresh], axis=0))):
                r._packed_cache = row.reshape(1, -1)
        # every rotation is checked (and warned about) before the first one runs; angles, generators and Clifford multiples of the
        # whole sequence go to the device library in ONE call (symgpu_perform_rotations_dev: no Python between the rotations)
        K = len(rotations)
        q_rows = np.zeros((K, 2 * wq), dtype='<u8')
        cos_t, sin_t = np.zeros(K), np.zeros(K)
        ks = np.zeros(K, dtype=np.int32)
        angles = []
        for r, (pauli_rotation, angle) in enumerate(rotations):
            assert pauli_rotation.n_terms == 1, 'Only rotation by single Pauliword allowed here'
            assert pauli_rotation.n_qubits == self.n_qubits, 'Pauliwords defined for different number of qubits'
            if angle is None:
                angle = np.pi / 2
            if pauli_rotation._c()[0] != 1:
                warnings.warn(f'Pword coefficient {pauli_rotation._c()[0]: .8f} has been set to 1')
            if getattr(angle, 'imag', 0) != 0:
                warnings.warn('Complex component in angle: this will be ignored.')
            angle = float(np.real(angle))
            angles.append(angle)
            q_rows[r] = pauli_rotation.packed[0]
            cos_t[r], sin_t[r], ks[r] = kernels.rotation_args(angle)
        dev = start = self._device()                             # resident already, or uploaded once and kept
        # The reference calls ``.cleanup()`` after every rotation (base.py:1185).  On an operator that has no duplicate rows and
        # no coefficient with |c| <= 1e-15 that cleanup is the identity, and the rotation kernels preserve both properties
        # (a rotated row P*Q can only coincide with an input row, which the kernels merge themselves; they also apply the
        # strict threshold).  So the device cleanup runs until the operator is known to be in that state — i.e. once, after
        # the first rotation of a user-supplied operator — and is skipped for the rest of the chain; a run of Clifford rotations
        # of a clean operator is one call of the chain entry point (rows in registers, csrc/rotate_chain.hip).
        # An operator without terms and 0 * I alternate under the reference's cleanup() (base.py:631-632, utils.py:275-278); the library
        # call follows that state machine itself, including WHY an operator is empty (emptied by a rotation: cleanup() gives 0 * I;
        # emptied by the cleanup: it stays without terms), so nothing is patched up here.
        clean = False
        step = 0
        while step < K:
            res, n_done, acted, clean = kernels.perform_rotations_dev(dev, q_rows[step:], cos_t[step:], sin_t[step:], ks[step:], clean)
            for r in np.flatnonzero(acted[:n_done]):
                _warn_large_angle(angles[step + int(r)], 1e-18)
            if res is not None:
                if dev is not start:
                    dev.free()                                  # an intermediate of this chain; `start` belongs to self
                dev = res
            assert n_done > 0, 'symgpu_perform_rotations_dev made no progress'
            step += n_done
        if dev is start:
            start.shared = True                                 # nothing acted and the operator was clean: a new object on the same rows
        if dev.n_terms == 0:
            # cleanup() of an operator whose terms all cancelled has shape (0, 2n) (test_base.py:124-130)
            return PauliwordOp(np.zeros((0, 2 * self.n_qubits), dtype=bool), [])
        return PauliwordOp._from_device(dev, self.n_qubits)

    def tensor(self, right_op: "PauliwordOp") -> "PauliwordOp":
        id_r = np.zeros([right_op.n_terms, self.n_qubits], dtype=bool)
        id_l = np.zeros([self.n_terms, right_op.n_qubits], dtype=bool)
        left = PauliwordOp(np.hstack([self.X_block, id_l, self.Z_block, id_l]), self._c().copy())
        right = PauliwordOp(np.hstack([id_r, right_op.X_block, id_r, right_op.Z_block]), right_op._c().copy())
        return left * right

    @cached_property
    def dagger(self) -> "PauliwordOp":
        return self._scaled(1, conjugate_first=True)

    @cached_property
    def to_dictionary(self) -> Dict[str, complex]:
        op = self.cleanup()
        return {symplectic_to_string(v): c for v, c in zip(op.symp_matrix, op._c())}

    # ---- a10 / f2: GF(2) generator routines (same device kernel as a8) ----------------------------------------
    @cached_property
    def generators(self) -> "PauliwordOp":
        """base.py:1436-1456: non-zero rows of ``_rref_binary(symp_matrix)`` — the packed rows are reduced on the device as they are (the
        column order x_0 .. x_{n-1}, z_0 .. z_{n-1} is the packed bit order; padding bits never become pivots): no bool matrix."""
        if self.n_terms == 0 or self.n_qubits == 0:
            row_red = _rref_binary(self.symp_matrix)
            non_zero_rows = row_red[np.sum(row_red, axis=1).astype(bool)]
            gens = PauliwordOp(non_zero_rows, np.ones(non_zero_rows.shape[0], dtype=complex))
        else:
            gens = PauliwordOp._from_device(kernels.generators_dev(self._device(rows_only=True)), self.n_qubits)
        assert check_independent(gens), 'generators are not independent'
        assert gens.n_terms <= 2 * self.n_qubits, 'cannot have an independent generating set of size greaterthan 2 time num qubits'
        return gens

    def generator_reconstruction(self, generators: "PauliwordOp", override_independence_check: bool = False):
        """base.py:523-560: ``cref_binary(vstack([G, M]))`` -> (R int[T, g], mask bool[T])."""
        if not override_independence_check:
            assert check_independent(generators), 'Supplied generators are algebraically dependent'
        dim = generators.n_terms
        assert self.n_qubits == generators.n_qubits, 'Pauliwords defined for different number of qubits'
        if dim == 0 or self.n_terms == 0 or self.n_qubits == 0 or dim > 2 * self.n_qubits:
            reduced = cref_binary(np.vstack([generators.symp_matrix, self.symp_matrix]))     # degenerate shapes: host glue around the device rref
            mask = np.all(~reduced[dim:, dim:], axis=1)
            return reduced[dim:, :dim].astype(int), mask
        # the transposed stack is built, reduced and read out in pivot order on the device (csrc/genrec.hip): neither operand is expanded
        return kernels.generator_reconstruction_dev(generators._device(rows_only=True), self._device(rows_only=True), self.n_qubits)
