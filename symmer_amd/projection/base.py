"""``S3Projection`` — stabiliser-subspace projection (reference ``symmer/projection/base.py:7-124``), SURVEY.md §8f
row f4: every data-parallel step (commutation with the rotated stabilisers, the Clifford rotation chain, the final
duplicate cleanup) runs on the MI355X through ``PauliwordOp``; the index bookkeeping is host NumPy as in the reference.
"""
from typing import List, Union
import numpy as np
from ..operators import PauliwordOp, IndependentOp


class S3Projection:
    rotated_flag = False

    def __init__(self, stabilizers: IndependentOp) -> None:
        self.stabilizers = stabilizers

    def _perform_projection(self, operator: PauliwordOp) -> PauliwordOp:
        """base.py:44-84: drop the terms that anticommute with a rotated stabiliser, fix the eigenvalues of the rest,
        delete the stabilised qubit columns and merge duplicates."""
        assert operator.n_qubits == self.stabilizers.n_qubits, 'The input operator does not have the same number of qubits as the stabilizers'
        assert self.rotated_flag, 'The operator has not been rotated - intended for use with perform_projection method'
        self.rotated_flag = False
        commutes_with_all = np.all(operator.commutes_termwise(self.rotated_stabilizers), axis=1)
        op_kept = operator.symp_matrix[commutes_with_all]
        cf_kept = operator.coeff_vec[commutes_with_all]
        stab_symp_indices = np.where(self.rotated_stabilizers.symp_matrix)[1]
        eigval_assignment = op_kept[:, stab_symp_indices] * self.rotated_stabilizers.coeff_vec
        eigval_assignment[eigval_assignment == 0] = 1
        coeff_sign_flip = cf_kept * (np.prod(eigval_assignment, axis=1)).T
        unfixed = np.hstack([self.free_qubit_indices, self.free_qubit_indices + operator.n_qubits])
        projected = op_kept[:, unfixed]
        if projected.shape[1]:
            return PauliwordOp(projected, coeff_sign_flip).cleanup()
        return PauliwordOp(np.zeros((1, 0), dtype=bool), [np.sum(coeff_sign_flip)])

    def perform_projection(self, operator: PauliwordOp, ref_state: Union[List[int], np.ndarray] = None,
                           sector: Union[List[int], np.ndarray] = None) -> PauliwordOp:
        """base.py:86-124."""
        if sector is None and ref_state is not None:
            self.stabilizers.update_sector(ref_state)
        elif sector is not None:
            self.stabilizers.coeff_vec = np.array(sector, dtype=int)
        self.rotated_stabilizers = self.stabilizers.rotate_onto_single_qubit_paulis()
        self.stab_qubit_indices = np.where(self.rotated_stabilizers.symp_matrix)[1] % operator.n_qubits
        self.free_qubit_indices = np.setdiff1d(np.arange(operator.n_qubits), self.stab_qubit_indices)
        if len(self.stabilizers.stabilizer_rotations) > 0:
            op_rotated = operator.perform_rotations(self.stabilizers.stabilizer_rotations)
        else:
            op_rotated = operator
        self.rotated_flag = True
        return self._perform_projection(operator=op_rotated)
