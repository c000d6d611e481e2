"""``CircuitSymmerlator`` — SURVEY.md §8f row f1 (reference ``symmer/evolution/circuit_symmerlator.py:8-166``).

A circuit is stored as a sequence of single-Pauli rotations (Clifford gates = multiples of pi/2, up to a global phase
that cancels in expectation values); applying it to an observable is ONE device-resident ``perform_rotations`` chain on
the MI355X (Heisenberg picture, sequence reversed), and ``evaluate`` sums the coefficients of the surviving I/Z strings
(the <0|...|0> expectation value).  QASM import (qiskit) is outside the accelerated path.
"""
from typing import List
import numpy as np
from ..operators import PauliwordOp
from .. import packing

_Q = np.pi / 2

# gate -> list of (Pauli letters on the gate's qubits, multiple of pi/2); same decompositions as the reference (:58-116)
_CLIFFORD = {
    'x': [('X', 2)], 'y': [('Y', 2)], 'z': [('Z', 2)],
    'h': [('Z', 2), ('Y', 1)],
    's': [('Z', 1)], 'sdg': [('Z', 3)],
    'sx': [('X', 1)], 'sy': [('Y', 1)], 'sz': [('Z', 1)],
    'cx': [('ZX', 1), ('ZI', 3), ('IX', 3)],
    'cy': [('ZY', 1), ('ZI', 3), ('IY', 3)],
    'cz': [('ZZ', 1), ('ZI', 3), ('IZ', 3)],
}


class CircuitSymmerlator:
    def __init__(self, n_qubits: int) -> None:
        self.n_qubits = n_qubits
        self.sequence = []
        self.gate_map = {'x': self.X, 'y': self.Y, 'z': self.Z, 'rx': self.RX, 'ry': self.RY, 'rz': self.RZ, 'sx': self.sqrtX,
                         'sy': self.sqrtY, 'sz': self.sqrtZ, 'cx': self.CX, 'cy': self.CY, 'cz': self.CZ, 'h': self.H, 's': self.S,
                         'sdg': self.Sdag, '': self.R, 't': self.T, 'ccx': self.Toffoli, 'swap': self.SWAP}

    def get_rotation_string(self, pauli: str, indices: List[int]) -> PauliwordOp:
        pauli = list(pauli)
        assert len(pauli) == len(indices), 'Number of Paulis and indices do not match'
        assert set(pauli).issubset({'I', 'X', 'Y', 'Z'}), 'Pauli operators are either I, X, Y or Z.'
        # the packed row directly (a few set bits) instead of an n-character string parsed by from_list: a depth-2,000 circuit on
        # 1,000 qubits spent 0.48 s building its 4,000 generators that way — the device run takes 0.02 s
        wq = packing.words_per_block(self.n_qubits)
        row = np.zeros((1, 2 * wq), dtype='<u8')
        for i, P in zip(indices, pauli):
            # the reference fills a list of characters (R[i] = P, circuit_symmerlator.py:38-40): negative indices count from the
            # end and a repeated index keeps the LAST letter — so both bits of the qubit are cleared before they are set
            assert -self.n_qubits <= i < self.n_qubits, 'qubit index out of range'
            i %= self.n_qubits
            bit = np.uint64(1) << np.uint64(i & 63)
            row[0, i >> 6] &= ~bit
            row[0, wq + (i >> 6)] &= ~bit
            if P in 'XY':
                row[0, i >> 6] |= bit
            if P in 'ZY':
                row[0, wq + (i >> 6)] |= bit
        return PauliwordOp._from_packed(row, self.n_qubits, [1])

    def pi_2_multiple(self, multiple: int) -> float:
        return _Q * multiple

    def _clifford(self, name: str, indices: List[int]) -> None:
        for letters, mult in _CLIFFORD[name]:
            self.sequence.append((self.get_rotation_string(letters, indices), self.pi_2_multiple(mult)))

    # Clifford gates
    def X(self, index: int) -> None: self._clifford('x', [index])
    def Y(self, index: int) -> None: self._clifford('y', [index])
    def Z(self, index: int) -> None: self._clifford('z', [index])
    def H(self, index: int) -> None: self._clifford('h', [index])
    def S(self, index: int) -> None: self._clifford('s', [index])
    def Sdag(self, index: int) -> None: self._clifford('sdg', [index])
    def sqrtX(self, index: int) -> None: self._clifford('sx', [index])
    def sqrtY(self, index: int) -> None: self._clifford('sy', [index])
    def sqrtZ(self, index: int) -> None: self._clifford('sz', [index])
    def CX(self, control: int, target: int) -> None: self._clifford('cx', [control, target])
    def CY(self, control: int, target: int) -> None: self._clifford('cy', [control, target])
    def CZ(self, control: int, target: int) -> None: self._clifford('cz', [control, target])

    def SWAP(self, qubit_1: int, qubit_2: int) -> None:
        self.CX(qubit_1, qubit_2); self.CX(qubit_2, qubit_1); self.CX(qubit_1, qubit_2)

    # non-Clifford gates (term count may double per gate)
    def R(self, pauli: str, indices: List[int], angle: float) -> None:
        self.sequence.append((self.get_rotation_string(pauli, indices), -angle))

    def RX(self, index: int, angle: float) -> None: self.R('X', [index], angle)
    def RY(self, index: int, angle: float) -> None: self.R('Y', [index], angle)
    def RZ(self, index: int, angle: float) -> None: self.R('Z', [index], angle)

    def T(self, index: int, angle: float) -> None:
        raise NotImplementedError()

    def Toffoli(self, control_1: int, control_2: int, target: int) -> None:
        raise NotImplementedError()

    # execution
    def apply_sequence(self, operator: PauliwordOp) -> PauliwordOp:
        assert operator.n_qubits == self.n_qubits, 'The operator is defined over a different number of qubits'
        return operator.perform_rotations(self.sequence[::-1])

    def evaluate(self, operator: PauliwordOp) -> complex:
        """<0|U^+ O U|0>: only I/Z strings (no X bit) contribute.  ``perform_rotations`` always returns a cleaned operator
        (also for an empty sequence), so the reference's extra ``cleanup()`` (circuit_symmerlator.py:163) is the identity here."""
        rotated = self.apply_sequence(operator)
        diagonal = ~np.any(rotated.X_block, axis=1)
        return np.sum(rotated.coeff_vec[diagonal]) if rotated.n_terms else 0
