"""``QuantumState`` — SURVEY.md §8f row f3 (reference ``symmer/operators/base.py:1564-2272``, the algebraic core only).

A sparse statevector is carried as a ``PauliwordOp`` (``|0> -> Z``, ``|1> -> X``, base.py:1564-1580), so applying an operator
to a state is the SAME device path as operator x operator: the fused all-pairs product + cleanup kernel, followed by the
``i^{Y}`` post-factor (base.py:854-857).  Expectation values reuse it (base.py:796-819, 2438-2471).
Not provided (dense/sparse matrices, sampling, partial traces, plotting): outside the accelerated path.
"""
from copy import deepcopy
from functools import cached_property
from numbers import Number
from typing import Dict, List, Union
import numpy as np


class QuantumState:
    sigfig = 3

    def __init__(self, state_matrix, coeff_vector=None, vec_type: str = 'ket') -> None:
        from .base import PauliwordOp
        if isinstance(state_matrix, list):
            state_matrix = np.array(state_matrix)
        if isinstance(coeff_vector, list):
            coeff_vector = np.array(coeff_vector)
        state_matrix = np.asarray(state_matrix)
        if len(state_matrix.shape) == 1:
            state_matrix = state_matrix.reshape([1, -1])
        state_matrix = state_matrix.astype(int)
        assert set(np.unique(state_matrix)).issubset({0, 1})
        self.n_terms, self.n_qubits = state_matrix.shape
        self.state_matrix = state_matrix
        if coeff_vector is None:
            coeff_vector = np.ones(self.n_terms) / np.sqrt(self.n_terms)
        self.vec_type = vec_type
        self.state_op = PauliwordOp(np.hstack([state_matrix, 1 - state_matrix]), coeff_vector)

    def copy(self) -> "QuantumState":
        return deepcopy(self)

    @classmethod
    def random(cls, num_qubits: int, num_terms: int, vec_type: str = 'ket') -> "QuantumState":
        random_state = np.random.randint(0, 2, (num_terms, num_qubits))
        coeff_vec = np.random.rand(num_terms) + np.random.rand(num_terms) * 1j
        return QuantumState(random_state, coeff_vec, vec_type=vec_type).cleanup().normalize

    @classmethod
    def zero(cls, n_qubits: int, vec_type: str = 'ket') -> "QuantumState":
        return QuantumState(np.zeros(n_qubits).reshape(1, -1), coeff_vector=np.array([1]), vec_type=vec_type)

    @classmethod
    def from_dictionary(cls, state_dict: Dict[str, complex]) -> "QuantumState":
        bits, coeffs = zip(*state_dict.items())
        coeffs = np.array([complex(*c) if isinstance(c, (tuple, list)) else c for c in coeffs])
        return cls(np.array([[int(i) for i in b] for b in bits]), coeffs)

    @cached_property
    def to_dictionary(self) -> Dict[str, complex]:
        s = self.cleanup()
        return dict(zip([''.join(str(i) for i in row) for row in s.state_matrix], s.state_op.coeff_vec))

    def __str__(self) -> str:
        out = ''
        for basis_vec, coeff in zip(self.state_matrix, self.state_op.coeff_vec):
            b = ''.join(str(i) for i in basis_vec)
            out += f'{coeff: .{self.sigfig}f} |{b}> +\n' if self.vec_type == 'ket' else f'{coeff: .{self.sigfig}f} <{b}| +\n'
        return out[:-3]

    __repr__ = __str__

    def __eq__(self, Qstate: "QuantumState") -> bool:
        return self.state_op == Qstate.state_op

    def __hash__(self):
        return hash(tuple(self.to_dictionary.items()))

    def __add__(self, Qstate: "QuantumState") -> "QuantumState":
        new_state = self.state_op + Qstate.state_op
        return QuantumState(new_state.X_block, new_state.coeff_vec)

    def __radd__(self, add_obj):
        if isinstance(add_obj, Number) and add_obj == 0:
            return self
        return self + add_obj

    def __sub__(self, Qstate: "QuantumState") -> "QuantumState":
        new_state_op = self.state_op - Qstate.state_op
        return QuantumState(new_state_op.X_block, new_state_op.coeff_vec)

    def __mul__(self, mul_obj):
        """bra * ket -> inner product; bra * PauliwordOp -> bra (base.py:1781-1830)."""
        from .base import PauliwordOp
        if isinstance(mul_obj, Number):
            return QuantumState(self.state_matrix, self.state_op.coeff_vec * mul_obj)
        assert self.n_qubits == mul_obj.n_qubits, 'Multiplication object defined for different number of qubits'
        assert self.vec_type == 'bra', 'Cannot multiply a ket from the right'
        if isinstance(mul_obj, QuantumState):
            assert mul_obj.vec_type == 'ket', 'Cannot multiply a bra with another bra'
            # base.py:1808-1815: the state with fewer terms is the left one; both are cleaned (to_dictionary, :2104), then the coefficients
            # of the basis strings they share are multiplied and added in the left state's order — here a hash join of the packed
            # basis rows on the device (csrc/project.hip), the products added in that same order
            left, right = (self, mul_obj) if self.state_op.n_terms < mul_obj.n_terms else (mul_obj, self)
            from .. import kernels
            cleaned = [kernels.cleanup_dev(s.state_op._device()) for s in (left, right)]
            try:
                return kernels.state_inner_dev(cleaned[0], cleaned[1])
            finally:
                for h in cleaned:
                    h.free()
        if isinstance(mul_obj, PauliwordOp):
            new_state_op = self.state_op * mul_obj
            coeff = new_state_op.coeff_vec * ((-1j) ** new_state_op.Y_count)
            return QuantumState(new_state_op.X_block, coeff, vec_type='bra').cleanup()
        raise ValueError('Trying to multiply QuantumState by unrecognised object - must be another Quantum state or PauliwordOp')

    def __getitem__(self, key) -> "QuantumState":
        if isinstance(key, (int, np.integer)):
            key = int(key)
            if key < 0:
                key += self.n_terms
            assert key < self.n_terms, 'Index out of range'
            mask = [key]
        elif isinstance(key, slice):
            mask = np.arange(0 if key.start is None else key.start, self.n_terms if key.stop is None else key.stop, key.step)
        else:
            mask = np.asarray(key)
        return QuantumState(self.state_matrix[mask], self.state_op.coeff_vec[mask])

    def __iter__(self):
        return iter([self[i] for i in range(self.n_terms)])

    def cleanup(self, zero_threshold=1e-15) -> "QuantumState":
        clean = self.state_op.cleanup(zero_threshold=zero_threshold)
        return QuantumState(clean.X_block, clean.coeff_vec, vec_type=self.vec_type)

    def sort(self, by='decreasing', key='magnitude') -> "QuantumState":
        if key == 'magnitude':
            order = np.argsort(-abs(self.state_op.coeff_vec))
        elif key == 'support':
            order = np.argsort(-np.sum(self.state_matrix, axis=1))
        else:
            raise ValueError('Only permitted sort key values are magnitude or support')
        if by == 'increasing':
            order = order[::-1]
        elif by != 'decreasing':
            raise ValueError('Only permitted sort by values are increasing or decreasing')
        return QuantumState(self.state_matrix[order], self.state_op.coeff_vec[order])

    @cached_property
    def normalize(self) -> "QuantumState":
        return QuantumState(self.state_matrix, self.state_op.coeff_vec / np.linalg.norm(self.state_op.coeff_vec), vec_type=self.vec_type)

    @cached_property
    def dagger(self) -> "QuantumState":
        return QuantumState(self.state_matrix, self.state_op.coeff_vec.conjugate(), vec_type='bra' if self.vec_type == 'ket' else 'ket')

    def _is_normalized(self) -> bool:
        return bool(np.isclose(np.linalg.norm(self.state_op.cleanup().coeff_vec), 1))


def single_term_expval(P_op, psi: QuantumState) -> float:
    """base.py:2438-2471: <psi|P|psi> for ONE Pauli term (its coefficient is ignored) as the difference of the squared
    norms of the projections onto the +1 / -1 eigenspaces, each obtained with one operator x state product on the device."""
    from .base import PauliwordOp
    assert P_op.n_terms == 1, 'Supplied multiple Pauli terms.'
    proj = np.vstack([np.zeros(P_op.n_qubits * 2, dtype=bool), P_op.symp_matrix])
    norm_ev = lambda ev: np.linalg.norm((PauliwordOp(proj, [.5, .5 * ev]) * psi).state_op.coeff_vec)
    return (norm_ev(+1) ** 2 - norm_ev(-1) ** 2).real
