"""``QubitTapering`` (reference ``symmer/projection/qubit_tapering.py:8-106``), SURVEY.md §8f row f4: symmetry
generators by GF(2) elimination on the device, Clifford rotation chain with the operator device-resident, projection.
Reference states are computational-basis bit arrays (``QuantumState`` is outside the accelerated path)."""
import warnings
from functools import cached_property
from typing import List, Union
import numpy as np
from ..operators import PauliwordOp, IndependentOp
from .base import S3Projection


class QubitTapering(S3Projection):
    name = 'qubit_tapering'

    def __init__(self, operator: PauliwordOp, target_sqp: str = 'Z') -> None:
        self.operator = operator
        self.target_sqp = target_sqp
        self.n_taper = self.symmetry_generators.n_terms
        super().__init__(self.symmetry_generators)

    @cached_property
    def symmetry_generators(self) -> IndependentOp:
        stabilizers = IndependentOp.symmetry_generators(self.operator)
        stabilizers.target_sqp = self.target_sqp
        return stabilizers

    def taper_it(self, ref_state: Union[List[int], np.ndarray] = None, sector: Union[List[int], np.ndarray] = None,
                 aux_operator: PauliwordOp = None) -> PauliwordOp:
        """qubit_tapering.py:54-106."""
        if self.symmetry_generators != self.stabilizers:
            warnings.warn('the defined symmetry generators have been updated from parent class stabilizers')
            S3Projection.__init__(self, self.symmetry_generators)
        operator_to_taper = aux_operator.copy() if aux_operator is not None else self.operator.copy()
        # (the reference additionally projects a QuantumState reference state, qubit_tapering.py:101-104: out of scope)
        return self.perform_projection(operator=operator_to_taper, ref_state=ref_state, sector=sector)
