"""Packing convention of the C-ABI (``include/symgpu.h``).

A symplectic row ``[x_0..x_{n-1} | z_0..z_{n-1}]`` (reference layout, symmer/operators/base.py:42-74)
becomes ``2*Wq`` little-endian uint64 words, X words first then Z words, ``Wq = max(1, ceil(n/64))``;
bit ``j`` of word ``w`` is qubit ``64*w + j``; padding bits are zero.
"""
import numpy as np


def words_per_block(n_qubits):
    return max(1, (int(n_qubits) + 63) // 64)


def pack_bits(mat, n_words=None):
    """bool[R, C] -> uint64[R, ceil(C/64)] (GF(2) matrix rows, same bit rule)."""
    mat = np.asarray(mat, dtype=bool)
    R, C = mat.shape
    wc = max(1, (C + 63) // 64) if n_words is None else n_words
    if C == wc * 64:
        by = np.packbits(mat, axis=1, bitorder='little')
    else:
        bits = np.zeros((R, wc * 64), dtype=bool)
        bits[:, :C] = mat
        by = np.packbits(bits, axis=1, bitorder='little')
    return np.ascontiguousarray(by).view('<u8').reshape(R, wc)


def unpack_bits(words, n_cols):
    words = np.ascontiguousarray(words, dtype='<u8')
    R = words.shape[0]
    if R == 0:
        return np.zeros((0, n_cols), dtype=bool)
    by = words.view(np.uint8).reshape(R, -1)
    return np.unpackbits(by, axis=1, bitorder='little', count=None)[:, :n_cols].astype(bool)


def pack_rows(symp_matrix):
    """bool[T, 2n] -> uint64[T, 2*Wq]."""
    symp_matrix = np.asarray(symp_matrix, dtype=bool)
    T, two_n = symp_matrix.shape
    n = two_n // 2
    wq = words_per_block(n)
    out = np.empty((T, 2 * wq), dtype='<u8')
    out[:, :wq] = pack_bits(symp_matrix[:, :n], wq)
    out[:, wq:] = pack_bits(symp_matrix[:, n:], wq)
    return out


def unpack_rows(packed, n_qubits):
    """uint64[T, 2*Wq] -> bool[T, 2n]."""
    packed = np.ascontiguousarray(packed, dtype='<u8')
    T = packed.shape[0]
    wq = packed.shape[1] // 2
    out = np.empty((T, 2 * n_qubits), dtype=bool)
    out[:, :n_qubits] = unpack_bits(packed[:, :wq], n_qubits)
    out[:, n_qubits:] = unpack_bits(packed[:, wq:], n_qubits)
    return out


def popcount_rows(packed):
    """uint64[T, W] -> int64[T]: set bits per row."""
    packed = np.ascontiguousarray(packed, dtype='<u8')
    if packed.shape[0] == 0:
        return np.zeros(0, dtype=np.int64)
    if hasattr(np, 'bitwise_count'):                                      # NumPy >= 2.0
        return np.bitwise_count(packed).sum(axis=1, dtype=np.int64)
    return _POP8[packed.view(np.uint8)].reshape(packed.shape[0], -1).sum(axis=1, dtype=np.int64)   # NumPy 1.x: byte table


_POP8 = np.array([bin(v).count('1') for v in range(256)], dtype=np.uint8)
