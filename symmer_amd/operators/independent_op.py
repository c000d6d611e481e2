"""``IndependentOp`` — drop-in for the reference class on the hot path
(``symmer/operators/independent_op.py:9-159``): algebraically independent +/-1 stabiliser sets and the
symmetry-generator kernel, with the GF(2) elimination on the MI355X (``csrc/gf2.hip``).
Out of scope: stabiliser rotations, sector assignment, clique selection (networkx) — SURVEY.md §2 #4.
"""
import warnings
from typing import Dict, List, Union
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
        # The reference now picks a commuting subset with networkx clique routines (independent_op.py:132-144),
        # which are graph glue outside the accelerated path.
        raise NotImplementedError('non-commuting symmetry generators: clique selection is outside the hot path; '
                                  'pass commuting_override=True to obtain the full generating set')

    def _check_stab(self) -> None:
        if not set(self.coeff_vec).issubset({0, +1, -1}):
            raise ValueError(f'Stabilizer coefficients not +/-1: {self.coeff_vec}')

    def _check_independent(self) -> None:
        if not check_independent(self):
            raise ValueError('The supplied stabilizers are not independent')

    def __str__(self) -> str:
        if self.n_terms == 0:
            return ''
        return super().__str__()

    def __getitem__(self, key) -> "IndependentOp":
        P = super().__getitem__(key)
        return IndependentOp(P.symp_matrix, P.coeff_vec, target_sqp=self.target_sqp)
