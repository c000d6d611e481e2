from .utils import (symplectic_cleanup, matmul_GF2, mul_symplectic, _rref_binary, rref_binary, _cref_binary,
                    cref_binary, check_independent, symplectic_to_string, string_to_symplectic,
                    random_symplectic_matrix, check_adjmat_noncontextual, check_jordan_independent, numba_binary_matmal_GF2,
                    numba_dot_matmal_GF2)
from .base import PauliwordOp
from .independent_op import IndependentOp
from .quantum_state import QuantumState, single_term_expval
