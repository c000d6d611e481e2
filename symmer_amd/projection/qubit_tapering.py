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
    """One qubit is removed per independent Z2 symmetry of ``operator``."""
    name = 'qubit_tapering'

    def __init__(self, operator: PauliwordOp, target_sqp: str = 'Z') -> None:
        self.operator, self.target_sqp = operator, target_sqp
        generators = self.symmetry_generators                      # cached: one GF(2) elimination per instance
        self.n_taper = generators.n_terms
        S3Projection.__init__(self, generators)

    @cached_property
    def symmetry_generators(self) -> IndependentOp:
        found = IndependentOp.symmetry_generators(self.operator)
        found.target_sqp = self.target_sqp
        return found

    def taper_it(self, ref_state: Union[List[int], np.ndarray] = None, sector: Union[List[int], np.ndarray] = None,
                 aux_operator: PauliwordOp = None) -> PauliwordOp:
        """qubit_tapering.py:54-106: project ``aux_operator`` (default: the operator itself) into the symmetry sector given
        by ``sector`` or identified from ``ref_state``."""
        if self.stabilizers != self.symmetry_generators:
            warnings.warn('the defined symmetry generators have been updated from parent class stabilizers')
            S3Projection.__init__(self, self.symmetry_generators)
        source = self.operator if aux_operator is None else aux_operator
        # (the reference additionally projects a QuantumState reference state, qubit_tapering.py:101-104: out of scope)
        return self.perform_projection(operator=source.copy(), ref_state=ref_state, sector=sector)
