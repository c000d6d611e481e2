"""symmer_amd — MI355X-native implementation of Symmer's symplectic Pauli-operator hot path.

Drop-in classes for that path (same names as ``symmer.operators``): ``PauliwordOp``, ``IndependentOp``; free
functions in ``symmer_amd.operators``.  All data-parallel work runs in hand-written HIP kernels for gfx950
(``symmer_amd/csrc``) behind the C ABI of ``include/symgpu.h``; there is no CPU fallback.
"""
from ._lib import SymgpuError
from .operators import PauliwordOp, IndependentOp, QuantumState

__all__ = ['PauliwordOp', 'IndependentOp', 'QuantumState', 'SymgpuError']
