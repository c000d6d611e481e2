"""``S3Projection`` — stabiliser-subspace projection (reference ``symmer/projection/base.py:7-124``), SURVEY.md §8f
row f4: every data-parallel step (commutation with the rotated stabilisers, the Clifford rotation chain, the final
duplicate cleanup) runs on the MI355X through ``PauliwordOp``; the index bookkeeping is host NumPy as in the reference.
"""
from typing import List, Union
import numpy as np
from ..operators import PauliwordOp, IndependentOp
from .. import kernels, packing


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
        n = operator.n_qubits
        keep = np.asarray(self.free_qubit_indices, dtype=np.int64)
        if keep.size == 0:
            # every qubit stabilised: a scalar (projection/base.py:83-84) — the surviving weights times their signs, from the packed rows
            survives = np.all(operator.commutes_termwise(fixed), axis=1)
            neg = np.zeros(operator.packed.shape[1], dtype='<u8')
            for row, ev in zip(fixed.packed, kernels.sector_signs(fixed.coeff_vec)):      # (ValueError for anything but -1, 0, +1, before use)
                if ev == -1:
                    neg |= row
            odd = (packing.popcount_rows(operator.packed[survives] & neg) & 1).astype(bool)
            weights = np.where(odd, -operator._c()[survives], operator._c()[survives])
            return PauliwordOp(np.zeros((1, 0), dtype=bool), [np.sum(weights)])
        # the whole step on the device, on packed rows (csrc/project.hip): anticommutation with the fixed stabilisers, eigenvalue signs, deletion
        # of the stabilised qubits and the final merge of equal terms — no one-byte-per-bit matrix, no operator round trip through the host
        # (the operator arrives resident from perform_rotations and the projected operator stays resident: no round trip)
        res, n_survived = kernels.project_dev(operator._device(), fixed.packed, np.asarray(fixed.coeff_vec), keep, n)
        if n_survived == 0:
            # nothing commutes with the stabilisers: the reference's cleanup() of an operator without terms is 0 * I (base.py:631-632)
            return PauliwordOp(np.zeros((1, 2 * keep.size), dtype=bool), [0])
        return PauliwordOp._from_device(res, int(keep.size))

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
