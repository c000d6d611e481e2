"""``IndependentOp`` — drop-in for the reference class (``symmer/operators/independent_op.py:9-383``): algebraically
independent +/-1 stabiliser sets, the symmetry-generator kernel with the GF(2) elimination on the MI355X
(``csrc/gf2.hip``), and — SURVEY.md §8f row f4 — the Clifford rotations onto single-qubit Paulis and the sector
assignment that the tapering workflow needs (host control flow over the device kernels).
Out of scope: clique selection for non-commuting generators (networkx) and ``QuantumState`` reference states.
"""
import warnings
from functools import reduce
from typing import Dict, List, Tuple, Union
import numpy as np
from .. import kernels, packing
from .base import PauliwordOp
from .utils import check_independent


class IndependentOp(PauliwordOp):
    def __init__(self, symp_matrix: np.ndarray, coeff_vec: Union[List[complex], np.ndarray] = None, target_sqp: str = 'Z'):
        symp_matrix = np.asarray(symp_matrix)
        if coeff_vec is None:
            coeff_vec = np.ones(symp_matrix.shape[0], dtype=complex)
        super().__init__(symp_matrix, coeff_vec)
        self._check_stab()
        self.coeff_vec = self.coeff_vec.real.astype(int)
        self._check_independent()
        if target_sqp in ['X', 'Z', 'Y']:
            self.target_sqp = target_sqp
        else:
            raise ValueError('Target single-qubit Pauli not recognised - must be X or Z')
        self.stabilizer_rotations = None
        self.used_indices = None

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
        vanished (:126); the device builds the transposed matrix bit-packed, row-reduces it with the blocked
        pivot-broadcast sweep and reads the same rows out, in the same order."""
        if PwordOp.n_terms == 0 or PwordOp.n_qubits == 0:
            S_symp = np.eye(2 * PwordOp.n_qubits, dtype=bool)     # nothing constrains the kernel
        else:
            rows, _ = kernels.symmetry_kernel(PwordOp.packed, PwordOp.n_qubits)
            S_symp = packing.unpack_rows(rows, PwordOp.n_qubits)
        S = cls(S_symp, np.ones(S_symp.shape[0]))
        if S.n_terms == 0:
            warnings.warn('The input PauliwordOp has no Z2 symmetries.')
            return S
        if commuting_override or np.all(S.adjacency_matrix):
            return S
        # non-commuting generators: take a commuting subset (independent_op.py:132-144; networkx graph glue on the
        # device-computed adjacency matrix)
        if S.n_terms < 10 or largest_clique:
            S_commuting = S.largest_clique(edge_relation='C')
        else:
            S_commuting = S.clique_cover(edge_relation='C', strategy='independent_set')[0]
            warnings.warn('Greedy method may identify non-optimal commuting symmetry terms; might be able to taper again.')
        return cls(S_commuting.symp_matrix, np.ones(S_commuting.n_terms, dtype=complex))

    def _check_stab(self) -> None:
        if not set(self.coeff_vec).issubset({0, +1, -1}):
            raise ValueError(f'Stabilizer coefficients not +/-1: {self.coeff_vec}')

    def _check_independent(self) -> None:
        if not check_independent(self):
            raise ValueError('The supplied stabilizers are not independent')

    def __str__(self) -> str:
        from .utils import symplectic_to_string
        out_string = ''
        for pauli_vec, coeff in zip(self.symp_matrix, self.coeff_vec):
            out_string += f'{coeff} {symplectic_to_string(pauli_vec)} \n'
        return out_string[:-2]

    def __repr__(self) -> str:
        return str(self)

    def __add__(self, Pword: "IndependentOp") -> "IndependentOp":
        return self.from_PauliwordOp(super().__add__(Pword))

    def _rotate_by_single_Pword(self, Pword: PauliwordOp, angle: float = None) -> "IndependentOp":
        return self.from_PauliwordOp(super()._rotate_by_single_Pword(Pword, angle))

    def perform_rotations(self, rotations: List[Tuple[PauliwordOp, float]]) -> "IndependentOp":
        return self.from_PauliwordOp(super().perform_rotations(rotations))

    # ---- f4: rotation onto single-qubit Paulis (independent_op.py:204-273) --------------------------------------
    def _recursive_rotations(self, basis: "IndependentOp") -> None:
        """independent_op.py:204-241: repeatedly pick the lowest-weight non-single-qubit stabiliser, its least-supported
        qubit, and the pi/2 rotation that maps it onto that qubit; every rotation is one device pass."""
        non_sqp = np.where(np.sum(basis.symp_matrix, axis=1) != 1)
        basis_non_sqp = IndependentOp(basis.symp_matrix[non_sqp], basis.coeff_vec[non_sqp])
        # the reference takes them from (basis - basis_non_sqp), i.e. after a cleanup: zero-coefficient terms drop out
        sqp_rows = basis.symp_matrix[(np.sum(basis.symp_matrix, axis=1) == 1) & (np.abs(basis.coeff_vec) > 1e-15)]
        sqp_indices = np.where(sqp_rows)[1] % self.n_qubits
        self.used_indices += np.append(sqp_indices, sqp_indices + self.n_qubits).tolist()
        if basis_non_sqp.n_terms == 0:
            return None
        row_sum = np.sum(basis_non_sqp.symp_matrix, axis=1)
        pivot_row = basis_non_sqp.symp_matrix[np.argsort(row_sum)][0]
        non_I = np.setdiff1d(np.where(pivot_row)[0], np.array(self.used_indices))
        col_sum = np.sum(basis_non_sqp.symp_matrix, axis=0)
        support = pivot_row * col_sum
        pivot_point = non_I[np.argmin(support[non_I])]
        target = np.zeros(2 * self.n_qubits, dtype=int)
        target[pivot_point + self.n_qubits * (-1) ** (pivot_point // self.n_qubits)] = 1
        pivot_rotation = PauliwordOp(np.bitwise_xor(target, pivot_row.astype(int)), [1])
        self.stabilizer_rotations.append((pivot_rotation, None))
        rotated_basis = basis_non_sqp._rotate_by_single_Pword(pivot_rotation)
        return self._recursive_rotations(rotated_basis)

    def generate_stabilizer_rotations(self) -> None:
        """independent_op.py:243-273."""
        assert self.n_terms <= self.n_qubits, 'Too many terms in basis to reduce to single-qubit Paulis'
        assert np.all(self.adjacency_matrix), 'The basis is not commuting, hence the rotation is not possible'
        self.stabilizer_rotations = []
        self.used_indices = []
        basis = self.copy()
        self._recursive_rotations(basis)
        rotated_basis = basis.perform_rotations(self.stabilizer_rotations)
        for P in rotated_basis:
            sqp_index = np.where(P.symp_matrix[0])[0][0] % self.n_qubits
            target = np.zeros(2 * self.n_qubits, dtype=int)
            if self.target_sqp in ['X', 'Y']:
                target[sqp_index] = 1
            if self.target_sqp in ['Y', 'Z']:
                target[sqp_index + self.n_qubits] = 1
            R_symp = np.bitwise_xor(target, P.symp_matrix[0].astype(int))
            if np.any(R_symp):
                self.stabilizer_rotations.append((PauliwordOp(R_symp, [1]), None))

    def update_sector(self, ref_state: Union[List[int], np.ndarray], threshold: float = 0.5) -> None:
        """independent_op.py:275-301 for a computational-basis reference state given as a bit array: the expectation
        value of a stabiliser is (-1)^{|z & b|} if it is diagonal (no X/Y) and 0 otherwise (then the assignment is 0,
        with the reference's warning).  ``QuantumState`` superpositions are outside the accelerated path."""
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
        """independent_op.py:303-319."""
        self.generate_stabilizer_rotations()
        if self.stabilizer_rotations != []:
            return IndependentOp.from_PauliwordOp(
                reduce(lambda x, y: x.append(y), [s.perform_rotations(self.stabilizer_rotations) for s in self]))
        return self

    def __getitem__(self, key) -> "IndependentOp":
        P = PauliwordOp.__getitem__(self, key)
        return IndependentOp(P.symp_matrix, P.coeff_vec)

    def __iter__(self):
        return iter([self[i] for i in range(self.n_terms)])
