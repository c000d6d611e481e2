"""``S3Projection`` — stabiliser-subspace projection (reference ``symmer/projection/base.py:7-124``), SURVEY.md §8f
row f4: every data-parallel step (commutation with the rotated stabilisers, the Clifford rotation chain, the final
duplicate cleanup) runs on the MI355X through ``PauliwordOp``; the index bookkeeping is host NumPy as in the reference.
"""
from typing import List, Union
import numpy as np
from ..operators import PauliwordOp, IndependentOp


class S3Projection:
    rotated_flag = False          # set by perform_projection, consumed by _perform_projection

    def __init__(self, stabilizers: IndependentOp) -> None:
        self.stabilizers = stabilizers

    def _perform_projection(self, operator: PauliwordOp) -> PauliwordOp:
        """base.py:44-84 on an operator that has ALREADY been taken through the stabiliser rotations: terms anticommuting
        with a single-qubit stabiliser vanish in the subspace; in the others every stabilised position that is occupied
        contributes the stabiliser's eigenvalue; the stabilised qubits are then deleted and equal terms merged."""
        assert operator.n_qubits == self.stabilizers.n_qubits, 'The input operator does not have the same number of qubits as the stabilizers'
        assert self.rotated_flag, 'The operator has not been rotated - intended for use with perform_projection method'
        self.rotated_flag = False
        fixed = self.rotated_stabilizers
        survives = np.all(operator.commutes_termwise(fixed), axis=1)                  # device commutation kernel
        terms, weights = operator.symp_matrix[survives], operator.coeff_vec[survives]
        # symplectic column of each single-qubit stabiliser and its eigenvalue; an occupied column contributes the
        # eigenvalue (the reference's product treats an eigenvalue 0 like an unoccupied column: factor 1)
        columns = np.nonzero(fixed.symp_matrix)[1]
        eigenvalues = np.asarray(fixed.coeff_vec)
        factors = np.where(terms[:, columns] & (eigenvalues != 0), eigenvalues, 1)
        weights = weights * np.prod(factors, axis=1)
        keep_columns = np.concatenate([self.free_qubit_indices, self.free_qubit_indices + operator.n_qubits])
        if keep_columns.size == 0:
            return PauliwordOp(np.zeros((1, 0), dtype=bool), [np.sum(weights)])        # every qubit stabilised: a scalar
        return PauliwordOp(terms[:, keep_columns], weights).cleanup()                  # device cleanup merges equal terms

    def perform_projection(self, operator: PauliwordOp, ref_state: Union[List[int], np.ndarray] = None,
                           sector: Union[List[int], np.ndarray] = None) -> PauliwordOp:
        """base.py:86-124: sector from ``sector`` (wins) or from the reference state, stabilisers rotated onto single-qubit
        Paulis, the operator taken through the same rotation chain (device resident), then ``_perform_projection``."""
        if sector is not None:
            self.stabilizers.coeff_vec = np.array(sector, dtype=int)
        elif ref_state is not None:
            self.stabilizers.update_sector(ref_state)
        self.rotated_stabilizers = self.stabilizers.rotate_onto_single_qubit_paulis()
        n = operator.n_qubits
        self.stab_qubit_indices = np.nonzero(self.rotated_stabilizers.symp_matrix)[1] % n
        self.free_qubit_indices = np.setdiff1d(np.arange(n), self.stab_qubit_indices)
        chain = self.stabilizers.stabilizer_rotations
        rotated = operator.perform_rotations(chain) if len(chain) > 0 else operator
        self.rotated_flag = True
        return self._perform_projection(operator=rotated)
