"""``IndependentOp`` — drop-in for the reference class (``symmer/operators/independent_op.py:9-383``): algebraically
independent +/-1 stabiliser sets, the symmetry-generator kernel with the GF(2) elimination on the MI355X
(``csrc/gf2.hip``), and — SURVEY.md §8f row f4 — the Clifford rotations onto single-qubit Paulis and the sector
assignment that the tapering workflow needs (host control flow over the device kernels).
Out of scope: clique selection for non-commuting generators (networkx) and ``QuantumState`` reference states.
"""
import warnings
from typing import Dict, List, Tuple, Union
import numpy as np
from .. import kernels, packing
from .base import PauliwordOp
from .utils import check_independent


_SINGLE_QUBIT_TARGETS = ('X', 'Y', 'Z')


def _weights(symp: np.ndarray) -> np.ndarray:
    """number of set symplectic bits per row (1 <=> a single-qubit X or Z)"""
    return np.count_nonzero(symp, axis=1)


class IndependentOp(PauliwordOp):
    """Algebraically independent stabilisers with eigenvalue assignments in {0, +1, -1} (kept as ``int``)."""

    def __init__(self, symp_matrix: np.ndarray, coeff_vec: Union[List[complex], np.ndarray] = None, target_sqp: str = 'Z'):
        rows = np.asarray(symp_matrix)
        PauliwordOp.__init__(self, rows, np.ones(rows.shape[0], dtype=complex) if coeff_vec is None else coeff_vec)
        # independent_op.py:33-43: eigenvalues first, then independence, then the rotation target
        self._check_stab()
        self.coeff_vec = self.coeff_vec.real.astype(int)
        self._check_independent()
        if target_sqp not in _SINGLE_QUBIT_TARGETS:
            raise ValueError('Target single-qubit Pauli not recognised - must be X or Z')
        self.target_sqp = target_sqp
        self.stabilizer_rotations, self.used_indices = None, None

    # ---- constructors ------------------------------------------------------------------------------------
    @classmethod
    def from_PauliwordOp(cls, PwordOp: PauliwordOp) -> "IndependentOp":
        return cls(PwordOp.symp_matrix, PwordOp.coeff_vec)

    @classmethod
    def from_list(cls, pauli_terms: List[str], coeff_vec: List[complex] = None) -> "IndependentOp":
        return cls.from_PauliwordOp(PauliwordOp.from_list(pauli_terms, coeff_vec))

    @classmethod
    def from_dictionary(cls, operator_dict: Dict[str, complex]) -> "IndependentOp":
        return cls.from_PauliwordOp(PauliwordOp.from_dictionary(operator_dict))

    @classmethod
    def symmetry_generators(cls, PwordOp: PauliwordOp, commuting_override: bool = False, largest_clique=False
                            ) -> "IndependentOp":
        """independent_op.py:90-144: kernel of ``H Omega`` over GF(2).  The reference column-reduces
        ``vstack([hstack([Z, X]), eye(2n)])`` (:124-125) and reads the identity part of the columns whose top part
        vanished (:126); the device builds the transposed matrix bit-packed, row-reduces it (blocked panel + Four-Russians
        sweep, ``csrc/gf2.hip``) and reads the same rows out, in the same order."""
        n = PwordOp.n_qubits
        if PwordOp.n_terms and n:
            packed_generators, _ = kernels.symmetry_kernel_handle(PwordOp._device(rows_only=True), n)
            found = packing.unpack_rows(packed_generators, n)
        else:
            found = np.eye(2 * n, dtype=bool)                     # nothing constrains the kernel
        generators = cls(found, np.ones(found.shape[0]))
        if generators.n_terms == 0:
            warnings.warn('The input PauliwordOp has no Z2 symmetries.')
        if generators.n_terms == 0 or commuting_override or bool(np.all(generators.adjacency_matrix)):
            return generators
        # the generators do not all commute: keep a commuting subset (independent_op.py:132-144 — networkx graph glue on
        # the device-computed adjacency matrix; exact for small sets or on request, greedy colouring otherwise)
        if largest_clique or generators.n_terms < 10:
            subset = generators.largest_clique(edge_relation='C')
        else:
            subset = generators.clique_cover(edge_relation='C', strategy='independent_set')[0]
            warnings.warn('Greedy method may identify non-optimal commuting symmetry terms; might be able to taper again.')
        return cls(subset.symp_matrix, np.ones(subset.n_terms, dtype=complex))

    # ---- validation --------------------------------------------------------------------------------------
    def _check_stab(self) -> None:
        allowed = {0, 1, -1}
        if any(c not in allowed for c in self.coeff_vec):
            raise ValueError(f'Stabilizer coefficients not +/-1: {self.coeff_vec}')

    def _check_independent(self) -> None:
        if check_independent(self):
            return
        raise ValueError('The supplied stabilizers are not independent')

    # ---- printing / closure of the PauliwordOp operations ------------------------------------------------------
    def __str__(self) -> str:
        from .utils import symplectic_to_string
        return ' \n'.join(f'{c} {symplectic_to_string(row)}' for row, c in zip(self.symp_matrix, self.coeff_vec))

    def __repr__(self) -> str:
        return self.__str__()

    def __add__(self, Pword: "IndependentOp") -> "IndependentOp":
        return self.from_PauliwordOp(super().__add__(Pword))

    def _rotate_by_single_Pword(self, Pword: PauliwordOp, angle: float = None) -> "IndependentOp":
        return self.from_PauliwordOp(super()._rotate_by_single_Pword(Pword, angle))

    def perform_rotations(self, rotations: List[Tuple[PauliwordOp, float]]) -> "IndependentOp":
        return self.from_PauliwordOp(super().perform_rotations(rotations))

    def __getitem__(self, key) -> "IndependentOp":
        picked = PauliwordOp.__getitem__(self, key)
        return IndependentOp(picked.symp_matrix, picked.coeff_vec)

    def __iter__(self):
        return iter([self[k] for k in range(self.n_terms)])

    # ---- f4: rotation onto single-qubit Paulis (independent_op.py:204-273) --------------------------------------
    def _recursive_rotations(self, basis: "IndependentOp") -> None:
        """independent_op.py:204-241, written as a loop: while a stabiliser of weight != 1 is left, take the lightest one,
        choose among its not-yet-used positions the one supported by the fewest other stabilisers, and rotate (pi/2, one
        device pass) by the Pauli that maps the stabiliser onto the conjugate single-qubit Pauli at that position."""
        n = self.n_qubits
        current = basis
        while True:
            weight = _weights(current.symp_matrix)
            single = (weight == 1) & (np.abs(current.coeff_vec) > 1e-15)          # the reference sees them after a cleanup
            qubits = np.where(current.symp_matrix[single])[1] % n
            self.used_indices += np.concatenate([qubits, qubits + n]).tolist()
            heavy = weight != 1
            if not np.any(heavy):
                return None
            rest = IndependentOp(current.symp_matrix[heavy], current.coeff_vec[heavy])
            lightest = rest.symp_matrix[np.argsort(_weights(rest.symp_matrix))[0]]
            candidates = np.setdiff1d(np.flatnonzero(lightest), np.array(self.used_indices))
            crowding = lightest * np.count_nonzero(rest.symp_matrix, axis=0)
            position = candidates[np.argmin(crowding[candidates])]
            conjugate = position + n if position < n else position - n            # X position <-> Z position of the same qubit
            generator = lightest.astype(int)
            generator[conjugate] ^= 1
            rotation = PauliwordOp(generator, [1])
            self.stabilizer_rotations.append((rotation, None))
            current = rest._rotate_by_single_Pword(rotation)

    def generate_stabilizer_rotations(self) -> None:
        """independent_op.py:243-273: the pi/2 rotations onto single-qubit Paulis, then one more per stabiliser whose
        single-qubit image is not of the requested kind (X, Y or Z)."""
        assert self.n_terms <= self.n_qubits, 'Too many terms in basis to reduce to single-qubit Paulis'
        assert np.all(self.adjacency_matrix), 'The basis is not commuting, hence the rotation is not possible'
        self.stabilizer_rotations, self.used_indices = [], []
        start = self.copy()
        self._recursive_rotations(start)
        n = self.n_qubits
        want_x, want_z = self.target_sqp in ('X', 'Y'), self.target_sqp in ('Y', 'Z')
        for image in start.perform_rotations(self.stabilizer_rotations):
            row = image.symp_matrix[0].astype(int)
            qubit = int(np.flatnonzero(row)[0]) % n
            fix = row.copy()
            fix[qubit] ^= int(want_x)
            fix[qubit + n] ^= int(want_z)
            if fix.any():
                self.stabilizer_rotations.append((PauliwordOp(fix, [1]), None))

    def update_sector(self, ref_state, threshold: float = 0.5) -> None:
        """independent_op.py:275-301.  A ``QuantumState`` (superpositions included) is measured stabiliser by stabiliser with
        ``single_term_expval`` — two operator x state products on the device each — and assigned +-1 where |<S>| > 0.5, else 0
        with the reference's warning; the state must be normalised (``AssertionError`` otherwise).  Like the reference, the
        cut-off is the fixed 0.5 of ``assign_value`` (independent_op.py:376): the ``threshold`` argument is accepted and not
        forwarded there either.  A computational-basis state given as a bit array takes the closed form
        <b|S|b> = (-1)^{|z & b|} for a diagonal stabiliser and 0 otherwise (the same values, without the products)."""
        from .quantum_state import QuantumState, single_term_expval
        if isinstance(ref_state, QuantumState):
            assert ref_state._is_normalized(), 'Reference state is not normalized.'
            assert ref_state.n_qubits == self.n_qubits, 'reference state defined over a different number of qubits'
            values = []
            for stabilizer in self:
                expval = single_term_expval(stabilizer, ref_state)
                values.append(int(np.sign(expval)) if abs(expval) > 0.5 else 0)
            self.coeff_vec = np.array(values, dtype=int)
        else:
            b = np.asarray(ref_state).reshape(-1)
            assert b.shape[0] == self.n_qubits and set(np.unique(b)).issubset({0, 1}), 'reference state must be a bit array over all qubits'
            diagonal = ~np.any(self.X_block, axis=1)
            signs = (-1) ** np.sum(np.bitwise_and(self.Z_block, b.astype(bool)), axis=1)
            self.coeff_vec = np.where(diagonal, signs, 0).astype(int)
        if np.any(self.coeff_vec == 0):
            from .utils import symplectic_to_string
            S_zero = [symplectic_to_string(r) for r in self.symp_matrix[self.coeff_vec == 0]]
            warnings.warn(f'The stabilizers {S_zero} were assigned zero values - bad reference state.')

    def rotate_onto_single_qubit_paulis(self) -> "IndependentOp":
        """independent_op.py:303-319: every stabiliser through the rotation chain, stacked again in the original order."""
        self.generate_stabilizer_rotations()
        if not self.stabilizer_rotations:
            return self
        images = [stabilizer.perform_rotations(self.stabilizer_rotations) for stabilizer in self]
        stacked = images[0]
        for image in images[1:]:
            stacked = stacked.append(image)
        return IndependentOp.from_PauliwordOp(stacked)
