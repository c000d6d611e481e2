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

from .. import kernels, packing
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
        self.coeff_vec = np.asarray(coeff_vec, dtype=complex)
        self.n_terms = symp_matrix.shape[0]
        assert self.n_terms == len(self.coeff_vec), 'coeff list and Pauliwords not same length'
        self._packed_cache = None

    # ---- the two layouts: reference bool matrix (host glue) and packed rows (the C-ABI operand) -----------
    @property
    def symp_matrix(self) -> np.ndarray:
        """bool[T, 2n] = [X | Z] as in the reference.  Results of device kernels arrive packed and are only expanded
        to one byte per bit when somebody asks (a 2.5e7-term, 1000-qubit product is 6 GB packed but 50 GB as bools)."""
        if self._symp is None:
            self._symp = packing.unpack_rows(self._packed_cache, self.n_qubits)
        return self._symp

    @property
    def X_block(self) -> np.ndarray:
        return self.symp_matrix[:, :self.n_qubits]

    @property
    def Z_block(self) -> np.ndarray:
        return self.symp_matrix[:, self.n_qubits:]

    @property
    def packed(self) -> np.ndarray:
        """uint64[T, 2*Wq] rows of the C-ABI; cached (``symp_matrix`` is treated as immutable, as in the reference)."""
        if self._packed_cache is None:
            self._packed_cache = packing.pack_rows(self._symp)
        return self._packed_cache

    @classmethod
    def _from_packed(cls, packed: np.ndarray, n_qubits: int, coeff_vec) -> "PauliwordOp":
        op = cls.__new__(cls)
        packed = np.ascontiguousarray(packed, dtype='<u8')
        assert packed.ndim == 2 and packed.shape[1] == 2 * packing.words_per_block(n_qubits)
        op._symp = None
        op._packed_cache = packed
        op.n_qubits = n_qubits
        op.n_terms = packed.shape[0]
        op.coeff_vec = np.asarray(coeff_vec, dtype=complex)
        assert op.n_terms == len(op.coeff_vec), 'coeff list and Pauliwords not same length'
        return op

    def _derive(self, index=None, coeff_vec=None) -> "PauliwordOp":
        """Row selection / new coefficients without touching layouts that have not been materialised."""
        op = PauliwordOp.__new__(PauliwordOp)
        op._symp = None if self._symp is None else (self._symp if index is None else self._symp[index])
        op._packed_cache = None if self._packed_cache is None else (self._packed_cache if index is None else
                                                                    np.ascontiguousarray(self._packed_cache[index]))
        op.n_qubits = self.n_qubits
        coeff = self.coeff_vec if coeff_vec is None else coeff_vec
        op.coeff_vec = np.asarray(coeff if index is None or coeff_vec is not None else coeff[index], dtype=complex)
        op.n_terms = len(op.coeff_vec)
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
            return format(self.coeff_vec[0], fmt)                 # a bare scalar
        return ' +\n'.join(f'{format(c, fmt)} {symplectic_to_string(row)}' for row, c in zip(self.symp_matrix, self.coeff_vec))

    def __repr__(self) -> str:
        return str(self)

    def copy(self) -> "PauliwordOp":
        return deepcopy(self)

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
        'magnitude': lambda P: np.argsort(-abs(P.coeff_vec)),
        'lex': lambda P: np.lexsort(P.symp_matrix.T) if P.n_terms else np.zeros(0, dtype=int),    # last column = primary key
        'weight': lambda P: np.argsort(-P.symp_matrix.sum(axis=1, dtype=int)),
        'support': lambda P: P._support_order(),
        'Z': lambda P: np.argsort(P._xz_score(P.n_qubits + 1, 1)),
        'X': lambda P: np.argsort(P._xz_score(1, P.n_qubits + 1)),
        'Y': lambda P: np.argsort(np.abs(P.X_block.astype(int) - P.Z_block.astype(int)).sum(axis=1)),
    }

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
        return PauliwordOp(np.hstack([self.X_block[:, perm], self.Z_block[:, perm]]), self.coeff_vec)

    def set_processing_method(self, method):
        """base.py:76-80 selects mp / ray / single_thread for the reference's host fan-out; the device path has no such pool."""
        if method not in ('mp', 'ray', 'single_thread'):
            raise ValueError('Invalid processing method, must be one of mp, single_thread or ray.')

    def to_dataframe(self):
        """base.py:1419-1434."""
        import pandas as pd
        frame = pd.DataFrame.from_dict({'Pauli terms': list(self.to_dictionary.keys()), 'Coefficients (real)': self.coeff_vec.real})
        if np.any(self.coeff_vec.imag):
            frame['Coefficients (imaginary)'] = self.coeff_vec.imag
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
        return kernels.ycount(self.packed)

    # ---- a5 ----------------------------------------------------------------------------------------------
    def cleanup(self, zero_threshold: float = 1e-15) -> "PauliwordOp":
        """base.py:617-638.  Edge cases as the reference: no terms -> one identity row with coefficient 0;
        0 qubits -> the scalar term (the reference raises there, SURVEY §8a'; we return the sum)."""
        if self.n_qubits == 0:
            return PauliwordOp(np.zeros((1, 0), dtype=bool), [np.sum(self.coeff_vec)])
        if self.n_terms == 0:
            return PauliwordOp(np.zeros((1, self.symp_matrix.shape[1]), dtype=bool), [0])
        rows, coeff = kernels.cleanup(self.packed, self.coeff_vec, zero_threshold)
        return PauliwordOp._from_packed(rows, self.n_qubits, coeff)

    def __eq__(self, Pword: "PauliwordOp") -> bool:
        """base.py:640-662: cleanup + lexicographic sort on both sides, exact rows, ``np.allclose`` coefficients."""
        check_1 = self.cleanup().sort('lex')
        check_2 = Pword.cleanup().sort('lex')
        if check_1.n_qubits != check_2.n_qubits:
            raise ValueError('Operators defined over differing numbers of qubits.')
        if check_1.n_terms != check_2.n_terms:
            return False
        return bool(not np.sum(np.logical_xor(check_1.symp_matrix, check_2.symp_matrix)) and
                    np.allclose(check_1.coeff_vec, check_2.coeff_vec))

    def __hash__(self) -> int:
        return hash(tuple(self.to_dictionary.items()))

    def append(self, PwordOp: "PauliwordOp") -> "PauliwordOp":
        assert self.n_qubits == PwordOp.n_qubits, 'Pauliwords defined for different number of qubits'
        coeff = np.hstack((self.coeff_vec, PwordOp.coeff_vec))
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
        op_copy = PwordOp.copy()
        op_copy.coeff_vec *= -1
        return self + op_copy

    def multiply_by_constant(self, const: complex) -> "PauliwordOp":
        return self._derive(coeff_vec=self.coeff_vec * const)

    # ---- a3 / a4 -----------------------------------------------------------------------------------------
    def _multiply_by_operator(self, PwordOp: "PauliwordOp", zero_threshold: float = 1e-15) -> "PauliwordOp":
        """base.py:764-794: ``self`` is the inner (fast) index and the LEFT factor; fused product + cleanup on device."""
        assert self.n_qubits == PwordOp.n_qubits, 'PauliwordOps defined for different number of qubits'
        rows, coeff = kernels.mul_cleanup(self.packed, self.coeff_vec, PwordOp.packed, PwordOp.coeff_vec, True, zero_threshold)
        return PauliwordOp._from_packed(rows, self.n_qubits, coeff)

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
        if self.n_terms < other.n_terms:
            rows, coeff = kernels.mul_cleanup(other.packed, other.coeff_vec, self.packed, self.coeff_vec, False, zero_threshold)
        else:
            rows, coeff = kernels.mul_cleanup(self.packed, self.coeff_vec, other.packed, other.coeff_vec, True, zero_threshold)
        out = PauliwordOp._from_packed(rows, self.n_qubits, coeff)
        if is_state:
            # identities were mapped to Z: II == ZZ as states, so fold i^Y into the coefficients and merge again (base.py:854-857)
            return QuantumState(out.X_block.astype(int), out.coeff_vec * (1j ** out.Y_count)).cleanup()
        return out

    def expval(self, psi) -> complex:
        """base.py:796-819: <psi|H|psi>; many-term operators use one bra x op x ket product, otherwise term by term."""
        from .quantum_state import single_term_expval
        if self.n_terms > psi.n_terms and psi.n_terms > 10:
            return (psi.dagger * self * psi).real
        expvals = np.array([single_term_expval(P, psi) for P in self]) if self.n_terms > 1 else np.array(single_term_expval(self, psi))
        return np.sum(expvals * self.coeff_vec).real

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
        return kernels.commutes(self.packed, self.packed if PwordOp is self else PwordOp.packed)

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
        dev = kernels.DeviceOp.upload(self.packed)
        try:
            return kernels.noncontextual_dev(dev)
        finally:
            dev.free()

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
        return bool(commutator.n_terms == 0 or np.all(commutator.coeff_vec[0] == 0))

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
        if Pword.coeff_vec[0] != 1:
            warnings.warn(f'Pword coefficient {Pword.coeff_vec[0]: .8f} has been set to 1')
        if self.n_terms == 0:
            return self
        op = kernels.DeviceOp.upload(self.packed, self.coeff_vec)
        try:
            res, all_commute = kernels.rotate_single_dev(op, Pword.packed[0], angle, clifford_threshold=threshold)
            if all_commute:
                return self                                     # identity action: the SAME object (base.py:1131-1133)
            _warn_large_angle(angle, threshold)                 # only on the non-Clifford, non-commuting branch (base.py:1156-1157)
            rows, coeff = res.download()
            res.free()
        finally:
            op.free()
        if rows.shape[0] == 0 and kernels.rotation_args(angle, threshold)[2] < 0 and not np.any(self.commutes_termwise(Pword)):
            # non-Clifford and nothing left: the reference returns `commute_self + anticom_part` (base.py:1159-1161), and the sum of
            # two operators without terms is 0 * I (append, then cleanup(): base.py:631-632)
            return PauliwordOp(np.zeros((1, 2 * self.n_qubits), dtype=bool), [0])
        return PauliwordOp._from_packed(rows, self.n_qubits, coeff)

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
            if pauli_rotation.coeff_vec[0] != 1:
                warnings.warn(f'Pword coefficient {pauli_rotation.coeff_vec[0]: .8f} has been set to 1')
            if getattr(angle, 'imag', 0) != 0:
                warnings.warn('Complex component in angle: this will be ignored.')
            angle = float(np.real(angle))
            angles.append(angle)
            q_rows[r] = pauli_rotation.packed[0]
            cos_t[r], sin_t[r], ks[r] = kernels.rotation_args(angle)
        dev = kernels.DeviceOp.upload(self.packed, self.coeff_vec)
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
        try:
            step = 0
            while step < K:
                res, n_done, acted, clean = kernels.perform_rotations_dev(dev, q_rows[step:], cos_t[step:], sin_t[step:], ks[step:], clean)
                for r in np.flatnonzero(acted[:n_done]):
                    _warn_large_angle(angles[step + int(r)], 1e-18)
                if res is not None:
                    dev.free()
                    dev = res
                assert n_done > 0, 'symgpu_perform_rotations_dev made no progress'
                step += n_done
            rows, coeff = dev.download()
        finally:
            dev.free()
        if rows.shape[0] == 0:
            # cleanup() of an operator whose terms all cancelled has shape (0, 2n) (test_base.py:124-130)
            return PauliwordOp(np.zeros((0, 2 * self.n_qubits), dtype=bool), [])
        return PauliwordOp._from_packed(rows, self.n_qubits, coeff)

    def tensor(self, right_op: "PauliwordOp") -> "PauliwordOp":
        id_r = np.zeros([right_op.n_terms, self.n_qubits], dtype=bool)
        id_l = np.zeros([self.n_terms, right_op.n_qubits], dtype=bool)
        left = PauliwordOp(np.hstack([self.X_block, id_l, self.Z_block, id_l]), self.coeff_vec)
        right = PauliwordOp(np.hstack([id_r, right_op.X_block, id_r, right_op.Z_block]), right_op.coeff_vec)
        return left * right

    @cached_property
    def dagger(self) -> "PauliwordOp":
        return self._derive(coeff_vec=self.coeff_vec.conjugate())

    @cached_property
    def to_dictionary(self) -> Dict[str, complex]:
        op = self.cleanup()
        return {symplectic_to_string(v): c for v, c in zip(op.symp_matrix, op.coeff_vec)}

    # ---- a10 / f2: GF(2) generator routines (same device kernel as a8) ----------------------------------------
    @cached_property
    def generators(self) -> "PauliwordOp":
        """base.py:1436-1456: non-zero rows of ``_rref_binary(symp_matrix)``."""
        row_red = _rref_binary(self.symp_matrix)
        non_zero_rows = row_red[np.sum(row_red, axis=1).astype(bool)]
        gens = PauliwordOp(non_zero_rows, np.ones(non_zero_rows.shape[0], dtype=complex))
        assert check_independent(gens), 'generators are not independent'
        assert gens.n_terms <= 2 * self.n_qubits, 'cannot have an independent generating set of size greaterthan 2 time num qubits'
        return gens

    def generator_reconstruction(self, generators: "PauliwordOp", override_independence_check: bool = False):
        """base.py:523-560: ``cref_binary(vstack([G, M]))`` -> (R int[T, g], mask bool[T])."""
        if not override_independence_check:
            assert check_independent(generators), 'Supplied generators are algebraically dependent'
        dim = generators.n_terms
        reduced = cref_binary(np.vstack([generators.symp_matrix, self.symp_matrix]))
        mask = np.all(~reduced[dim:, dim:], axis=1)
        return reduced[dim:, :dim].astype(int), mask
