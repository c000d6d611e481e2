"""Free functions of the symplectic hot path with the reference's names and argument meaning
(``symmer/operators/utils.py``), running on the MI355X through ``libsymgpu.so``.

Bool-matrix in / bool-matrix out like the reference; packing to the C-ABI's 64-bit words happens here.
Everything that is NOT on the hot path (string converters) is plain host code.
"""
from typing import Tuple
import numpy as np
from .. import kernels, packing


# ---- string <-> symplectic glue (host; reference utils.py:80-163) --------------------------------
_CHAR = np.array(['I', 'X', 'Z', 'Y'])


def symplectic_to_string(symp_vec) -> str:
    symp_vec = np.asarray(symp_vec, dtype=bool)
    n = symp_vec.shape[0] // 2
    code = symp_vec[:n].astype(np.int8) + 2 * symp_vec[n:].astype(np.int8)
    return ''.join(_CHAR[code])


def string_to_symplectic(pauli_str: str, n_qubits: int) -> np.ndarray:
    assert len(pauli_str) == n_qubits, 'Number of qubits is incompatible with pauli string'
    assert set(pauli_str).issubset({'I', 'X', 'Y', 'Z'}), 'pauliword must only contain X,Y,Z,I terms'
    chars = np.frombuffer(pauli_str.encode('ascii'), dtype=np.uint8)
    x = (chars == ord('X')) | (chars == ord('Y'))
    z = (chars == ord('Z')) | (chars == ord('Y'))
    return np.hstack([x, z]).astype(int)


def random_symplectic_matrix(n_qubits, n_terms, diagonal=False, density=0.3):
    """Reference utils.py:281-290 (host RNG; input generation, not part of the kernels)."""
    if diagonal:
        z = np.random.choice([True, False], size=[n_terms, n_qubits], p=[density / 2, 1 - density / 2])
        return np.hstack([np.zeros_like(z), z])
    return np.random.choice([True, False], size=[n_terms, 2 * n_qubits], p=[density, 1 - density])


# ---- a5 -------------------------------------------------------------------------------------------
def symplectic_cleanup(symp_matrix: np.ndarray, coeff_vec: np.ndarray, zero_threshold: float = None
                       ) -> Tuple[np.ndarray, np.ndarray]:
    """Reference utils.py:230-279: merge duplicate rows summing coefficients, drop ``abs(c) <= zero_threshold``
    (strict ``>`` keeps; ``None`` keeps all), first-occurrence order."""
    symp_matrix = np.asarray(symp_matrix, dtype=bool)
    n = symp_matrix.shape[1] // 2
    rows, coeff = kernels.cleanup(packing.pack_rows(symp_matrix), coeff_vec, zero_threshold)
    return packing.unpack_rows(rows, n), coeff


# ---- a6 -------------------------------------------------------------------------------------------
def matmul_GF2(A: np.ndarray, B: np.ndarray) -> np.ndarray:
    """Reference utils.py:9-26: ``(A @ B) % 2`` on boolean matrices, as a bit-packed AND/XOR-popcount kernel.
    A is [N, K], B is [K, M]; rows of A and columns of B are packed into the X halves of pseudo-symplectic rows
    (Z halves zero / swapped) so that the commutation kernel's symplectic form equals the plain GF(2) dot."""
    A = np.asarray(A, dtype=bool); B = np.asarray(B, dtype=bool)
    assert A.shape[1] == B.shape[0]
    K = A.shape[1]
    wq = packing.words_per_block(K)
    a = np.zeros((A.shape[0], 2 * wq), dtype='<u8'); b = np.zeros((B.shape[1], 2 * wq), dtype='<u8')
    a[:, :wq] = packing.pack_bits(A, wq)              # x_i = row of A, z_i = 0
    b[:, wq:] = packing.pack_bits(B.T, wq)            # z'_j = column of B, x'_j = 0   ->  <i,j> = x_i . z'_j
    return ~kernels.commutes(a, b)


def numba_binary_matmal_GF2(A: np.ndarray, B: np.ndarray) -> np.ndarray:
    """Reference utils.py:28-61 (AND/XOR loops under numba): the same boolean ``(A @ B) % 2`` on the device kernel."""
    return matmul_GF2(A, B)


def numba_dot_matmal_GF2(A: np.ndarray, B: np.ndarray) -> np.ndarray:
    """Reference utils.py:63-78 (``np.dot`` in float64, ``% 2``): the same boolean ``(A @ B) % 2`` on the device kernel."""
    return matmul_GF2(A, B)


def mul_symplectic(symp_vec1, coeff1, symp_vec2, coeff2):
    """Reference utils.py:429-470 (scalar twin of the all-pairs product): one pair through the device kernel."""
    s1 = np.asarray(symp_vec1, dtype=bool).reshape(1, -1); s2 = np.asarray(symp_vec2, dtype=bool).reshape(1, -1)
    rows, coeff = kernels.mul_allpairs(packing.pack_rows(s1), [coeff1], packing.pack_rows(s2), [coeff2], True)
    return packing.unpack_rows(rows, s1.shape[1] // 2)[0], coeff[0]


# ---- a8 -------------------------------------------------------------------------------------------
def _rref_binary(matrix: np.ndarray) -> np.ndarray:
    """Reference utils.py:292-315: GF(2) reduced row-echelon form WITHOUT row swaps."""
    matrix = np.asarray(matrix, dtype=bool)
    if matrix.shape[0] == 0 or matrix.shape[1] == 0:
        return matrix.copy()
    red, _ = kernels.rref(packing.pack_bits(matrix))
    return packing.unpack_bits(red, matrix.shape[1])


def rref_binary(matrix: np.ndarray) -> np.ndarray:
    """Reference utils.py:317-335: rows with a pivot ordered by pivot column, zero rows last."""
    matrix = np.asarray(matrix, dtype=bool)
    if matrix.shape[0] == 0 or matrix.shape[1] == 0:
        return matrix.copy()
    red, _, piv = kernels.rref(packing.pack_bits(matrix), want_pivots=True)
    has = np.flatnonzero(piv >= 0)
    order = np.concatenate([has[np.argsort(piv[has], kind='stable')], np.flatnonzero(piv < 0)])
    return packing.unpack_bits(red, matrix.shape[1])[order]


def _cref_binary(matrix: np.ndarray) -> np.ndarray:
    """Reference utils.py:337-347."""
    return _rref_binary(np.asarray(matrix, dtype=bool).T).T


def cref_binary(matrix: np.ndarray) -> np.ndarray:
    """Reference utils.py:349-359."""
    return rref_binary(np.asarray(matrix, dtype=bool).T).T


def check_independent(operators) -> bool:
    """Reference utils.py:504-519."""
    if operators.n_terms > 2 * operators.n_qubits:
        return False
    if operators.n_terms == 0:
        return True
    if operators.n_qubits == 0:
        return False                                               # a term without qubits is a zero row
    # no zero row in _rref_binary(symp_matrix) <=> full row rank; the rank comes from the packed rows on the device (csrc/genrec.hip)
    return kernels.gf2_rank_dev(operators._device(rows_only=True)) == operators.n_terms


# ---- f4: noncontextuality test on the device-computed adjacency matrix (reference utils.py:567-589) --------------
def check_adjmat_noncontextual(adjmat) -> bool:
    """Terms that do not commute universally must split into cliques: the unique rows of the masked adjacency matrix
    have to be disjoint (https://doi.org/10.1103/PhysRevLett.123.200501)."""
    adjmat = np.asarray(adjmat, dtype=bool)
    mask_non_universal = np.where(~np.all(adjmat, axis=1))[0]
    unique_character = np.unique(adjmat[mask_non_universal, :][:, mask_non_universal], axis=0)
    return bool(np.all(np.count_nonzero(unique_character, axis=0) == 1))


def check_jordan_independent(operators) -> bool:
    """Reference utils.py:521-566: independence under the Jordan product PQ = {P,Q}/2.  At most 3n terms; the globally
    commuting terms must be independent as ordinary generators; then the rows [X-only | Z-only | Y] (Y treated as its own
    symbol) must have no dependency — all three tests on the device GF(2) / commutation kernels."""
    if operators.n_terms > 3 * operators.n_qubits:
        return False
    commute = operators.commutes_termwise(operators)
    universal = operators[np.all(commute, axis=1)]
    if not check_independent(universal):
        return False
    y = operators.X_block & operators.Z_block
    three_symbol = np.hstack([operators.X_block ^ y, operators.Z_block ^ y, y])
    reduced = _rref_binary(three_symbol)
    return bool(~np.any(np.all(~reduced, axis=1)))
