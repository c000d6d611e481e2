"""NumPy oracle: a CPU restatement of the reference's symplectic hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``symmer_amd/`` may import this module; it is imported by
``tests/``, by ``__graft_entry__.smoke()`` and by ``bench.py``'s ``cpu_baseline`` leg as the
*checker* / the *reported CPU baseline*, never as the product path.

Parity status: PINNED.  Every function below is checked (a) against the reference itself, imported
in the build container through ``oracle/tools/ref_shim.py`` (``oracle/tools/check_oracle_vs_ref.py``,
``oracle/tools/gen_golden.py``), and (b) against the committed golden fixtures
``tests/golden/*.npz`` in ``tests/test_oracle_golden.py``.  One third-party piece is not under
/root/reference: qiskit 1.2.4's Rust ``unordered_unique`` (``poetry.lock:2945``), call site
``symmer/operators/utils.py:271``.  Its published algorithm (iterate rows, hash-map row -> id, record
the index on first sight, ``inverse[i] = id``) is restated in ``first_occurrence_unique``; the
resulting *row order after cleanup* is therefore pinned to that published algorithm, not to a run of
the Rust code (no reference test asserts post-cleanup order).

Data layout is the reference's: ``symp`` is a C-order ``bool[T, 2n]`` with the X block in columns
``0..n-1`` and the Z block in ``n..2n-1``; coefficients are ``complex128[T]``.
All ``file:line`` citations are relative to ``/root/reference/``.
"""
import numpy as np

__all__ = [
    'first_occurrence_unique', 'symplectic_cleanup', 'cleanup_op', 'y_count', 'multiply_by_operator',
    'mul', 'matmul_gf2', 'commutes_termwise', 'rref_noswap', 'rref_ordered', 'cref_noswap',
    'cref_ordered', 'rotate_by_single_pword', 'perform_rotations', 'symmetry_generators_symp',
    'check_independent', 'generator_reconstruction', 'generators', 'pack_rows', 'unpack_rows',
    'lex_order',
]


# --------------------------------------------------------------------------------------------
# packing convention of the C-ABI, restated independently of symmer_amd.packing
# --------------------------------------------------------------------------------------------
def pack_rows(symp):
    """bool[T, 2n] -> uint64[T, 2*Wq]; bit j of word w <-> qubit 64*w + j; X words then Z words."""
    symp = np.asarray(symp, dtype=bool)
    T, two_n = symp.shape
    n = two_n // 2
    wq = max(1, (n + 63) // 64)
    out = np.zeros((T, 2 * wq), dtype=np.uint64)
    for blk in range(2):
        bits = np.zeros((T, wq * 64), dtype=np.uint8)
        bits[:, :n] = symp[:, blk * n:(blk + 1) * n]
        by = np.packbits(bits, axis=1, bitorder='little')
        out[:, blk * wq:(blk + 1) * wq] = np.ascontiguousarray(by).view('<u8')
    return out


def unpack_rows(packed, n):
    packed = np.ascontiguousarray(packed, dtype=np.uint64)
    T = packed.shape[0]
    wq = packed.shape[1] // 2
    out = np.zeros((T, 2 * n), dtype=bool)
    for blk in range(2):
        by = np.ascontiguousarray(packed[:, blk * wq:(blk + 1) * wq]).view(np.uint8)
        bits = np.unpackbits(by, axis=1, bitorder='little')
        out[:, blk * n:(blk + 1) * n] = bits[:, :n].astype(bool)
    return out


# --------------------------------------------------------------------------------------------
# a5: duplicate-term cleanup  (symmer/operators/utils.py:230-279, base.py:617-638)
# --------------------------------------------------------------------------------------------
def first_occurrence_unique(rows):
    """(indices of first occurrence in input order, inverse map) of the rows of a 2-D array.

    Restates qiskit 1.2.4 ``unordered_unique`` as called at utils.py:271.
    """
    rows = np.ascontiguousarray(rows)
    if rows.shape[0] == 0:
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
    if rows.shape[1] == 0:
        return np.zeros(1, dtype=np.int64), np.zeros(rows.shape[0], dtype=np.int64)
    keys = rows.view(np.dtype((np.void, rows.dtype.itemsize * rows.shape[1]))).ravel()
    _, first, inverse = np.unique(keys, return_index=True, return_inverse=True)
    by_first = np.argsort(first, kind='stable')
    new_id = np.empty_like(by_first)
    new_id[by_first] = np.arange(by_first.size)
    return first[by_first], new_id[inverse.ravel()]


def symplectic_cleanup(symp, coeff, zero_threshold=None):
    """utils.py:230-279 — merge duplicate rows (sequential sum in input order, ``np.add.at``), keep
    ``abs(c) > zero_threshold`` (strict), output in first-occurrence order."""
    symp = np.asarray(symp, dtype=bool)
    coeff = np.asarray(coeff, dtype=complex)
    first, inverse = first_occurrence_unique(symp.astype(np.uint8))
    out_rows = symp[first]
    out_coeff = np.zeros(first.shape[0], dtype=complex)
    np.add.at(out_coeff, inverse, coeff)
    if zero_threshold is not None:
        keep = np.abs(out_coeff) > zero_threshold
        out_rows, out_coeff = out_rows[keep], out_coeff[keep]
    return out_rows, out_coeff


def cleanup_op(symp, coeff, zero_threshold=1e-15):
    """base.py:617-638 including the two edge cases (0 terms -> one identity row with coeff 0;
    0 qubits -> scalar term; the reference raises there, SURVEY §8a' — we return the scalar)."""
    symp = np.asarray(symp, dtype=bool)
    coeff = np.asarray(coeff, dtype=complex)
    if symp.shape[1] == 0:
        return np.zeros((1, 0), dtype=bool), np.array([np.sum(coeff)], dtype=complex)
    if symp.shape[0] == 0:
        return np.zeros((1, symp.shape[1]), dtype=bool), np.zeros(1, dtype=complex)
    return symplectic_cleanup(symp, coeff, zero_threshold)


# --------------------------------------------------------------------------------------------
# a2/a3/a4: Y count and the all-pairs product (base.py:604-615, 764-794, 821-859)
# --------------------------------------------------------------------------------------------
def y_count(symp):
    symp = np.asarray(symp, dtype=bool)
    n = symp.shape[1] // 2
    return np.sum(symp[:, :n] & symp[:, n:], axis=1)


def product_rows_and_coeffs(symp_l, coeff_l, symp_r, coeff_r):
    """Uncleaned product ``L._multiply_by_operator(R)`` of base.py:783-792.

    Row ``q*N + p`` is ``L[p] xor R[q]`` with coefficient
    ``l_p * r_q * (-1)^{|x_p & z_q|} * i^{(3(Y_p+Y_q)+Y_out) mod 4}`` (L acts first = left factor).
    """
    symp_l = np.asarray(symp_l, dtype=bool); symp_r = np.asarray(symp_r, dtype=bool)
    coeff_l = np.asarray(coeff_l, dtype=complex); coeff_r = np.asarray(coeff_r, dtype=complex)
    N, two_n = symp_l.shape
    M = symp_r.shape[0]
    n = two_n // 2
    r3 = symp_r.reshape(M, 1, two_n)
    prod = symp_l[None, :, :] ^ r3                                        # [M, N, 2n]
    y_in = y_count(symp_l)[None, :] + y_count(symp_r)[:, None]            # [M, N]
    y_out = np.sum(prod[:, :, :n] & prod[:, :, n:], axis=2)
    flips = np.sum(symp_l[None, :, :n] & r3[:, :, n:], axis=2) % 2
    phase = ((-1) ** flips) * (1j) ** ((3 * y_in + y_out) % 4)
    coeff = (phase * np.outer(coeff_l, coeff_r).T).reshape(-1)
    return prod.reshape(M * N, two_n), coeff


def multiply_by_operator(symp_l, coeff_l, symp_r, coeff_r, zero_threshold=1e-15):
    """base.py:764-794: product followed by ``symplectic_cleanup``."""
    assert np.asarray(symp_l).shape[1] == np.asarray(symp_r).shape[1]
    rows, coeff = product_rows_and_coeffs(symp_l, coeff_l, symp_r, coeff_r)
    return symplectic_cleanup(rows, coeff, zero_threshold)


def mul(symp_a, coeff_a, symp_b, coeff_b, zero_threshold=1e-15):
    """``A * B`` with the operand swap of base.py:847-852 (dagger = conjugate coefficients,
    base.py:1366-1376): the operand with fewer terms is the outer index."""
    symp_a = np.asarray(symp_a, dtype=bool); symp_b = np.asarray(symp_b, dtype=bool)
    coeff_a = np.asarray(coeff_a, dtype=complex); coeff_b = np.asarray(coeff_b, dtype=complex)
    if symp_a.shape[0] < symp_b.shape[0]:
        rows, coeff = multiply_by_operator(symp_b, coeff_b.conjugate(), symp_a, coeff_a.conjugate(),
                                           zero_threshold)
        return rows, coeff.conjugate()
    return multiply_by_operator(symp_a, coeff_a, symp_b, coeff_b, zero_threshold)


# --------------------------------------------------------------------------------------------
# a6: commutation (base.py:938-971, utils.py:9-26, 63-78)
# --------------------------------------------------------------------------------------------
def matmul_gf2(a, b):
    """utils.py:63-78: float64 dot, mod 2, to bool (exact: sums < 2^53)."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return np.asarray(np.dot(a, b) % 2, dtype=np.bool_)


def commutes_termwise(symp_a, symp_b):
    """base.py:971: ``~matmul_GF2(A, hstack(Z_B, X_B).T)``; True = commute."""
    symp_a = np.asarray(symp_a, dtype=bool); symp_b = np.asarray(symp_b, dtype=bool)
    assert symp_a.shape[1] == symp_b.shape[1]
    n = symp_a.shape[1] // 2
    omega_b = np.hstack((symp_b[:, n:], symp_b[:, :n])).T
    if symp_a.shape[0] == 0 or symp_b.shape[0] == 0:
        return np.ones((symp_a.shape[0], symp_b.shape[0]), dtype=bool)
    return ~matmul_gf2(symp_a, omega_b)


# --------------------------------------------------------------------------------------------
# a8: GF(2) row reduction (utils.py:292-359)
# --------------------------------------------------------------------------------------------
def rref_noswap(matrix, count_xors=False):
    """utils.py:292-315 (``_rref_binary``): for each row in order, if non-zero take its leftmost set
    column as pivot and XOR the row into every OTHER row that has that column set.  No row swaps.
    With ``count_xors`` also returns sum_i |update_set_i| (the unit of the row-XOR metric)."""
    m = np.array(matrix, dtype=bool, copy=True)
    n_xor = 0
    for i in range(m.shape[0]):
        nz = np.flatnonzero(m[i])
        if nz.size == 0:
            continue
        pivot = nz[0]
        targets = np.flatnonzero(m[:, pivot])
        targets = targets[targets != i]
        n_xor += targets.size
        if targets.size:
            m[targets] ^= m[i]
    return (m, n_xor) if count_xors else m


def rref_ordered(matrix):
    """utils.py:317-335 (``rref_binary``): rows with a pivot ordered by pivot column, zero rows last.
    (Divergence: the reference raises on an all-zero matrix because ``zip(*[])`` cannot unpack;
    we return the zero matrix.)"""
    red = rref_noswap(matrix)
    lead = [(int(np.flatnonzero(r)[0]), i) for i, r in enumerate(red) if r.any()]
    lead.sort(key=lambda t: t[0])          # stable: ties impossible (pivot columns are distinct)
    order = [i for _, i in lead]
    used = set(order)
    order += [i for i in range(red.shape[0]) if i not in used]
    return red[order]


def cref_noswap(matrix):
    """utils.py:337-347 (``_cref_binary``)."""
    return rref_noswap(np.asarray(matrix, dtype=bool).T).T


def cref_ordered(matrix):
    """utils.py:349-359 (``cref_binary``)."""
    return rref_ordered(np.asarray(matrix, dtype=bool).T).T


def check_independent(symp):
    """utils.py:504-519."""
    symp = np.asarray(symp, dtype=bool)
    if symp.shape[0] > symp.shape[1]:
        return False
    red = rref_noswap(symp)
    return not np.any(np.all(~red, axis=1))


def check_jordan_independent(symp):
    """utils.py:521-566: at most 3n terms; the globally commuting terms are independent; no dependency among the rows
    [X-only | Z-only | Y] (Y as a third symbol)."""
    symp = np.asarray(symp, dtype=bool)
    t, n = symp.shape[0], symp.shape[1] // 2
    if t > 3 * n:
        return False
    universal = np.all(commutes_termwise(symp, symp), axis=1)
    if not check_independent(symp[universal]):
        return False
    x, z = symp[:, :n], symp[:, n:]
    y = x & z
    red = rref_noswap(np.hstack([x ^ y, z ^ y, y]))
    return not np.any(np.all(~red, axis=1))


def reindex(symp, old_indices, new_indices):
    """base.py:493-521: column ``old`` of the result is column ``new`` of the input, in both blocks."""
    symp = np.asarray(symp, dtype=bool)
    n = symp.shape[1] // 2
    x, z = symp[:, :n].copy(), symp[:, n:].copy()
    x[:, list(old_indices)] = symp[:, :n][:, list(new_indices)]
    z[:, list(old_indices)] = symp[:, n:][:, list(new_indices)]
    return np.hstack([x, z])


def generators(symp):
    """base.py:1436-1456: non-zero rows of ``_rref_binary(symp)``."""
    red = rref_noswap(symp)
    return red[np.any(red, axis=1)]


def generator_reconstruction(symp_op, symp_gen):
    """base.py:523-560: ``cref_binary(vstack([G, M]))`` -> (R int[T, g], mask bool[T])."""
    symp_op = np.asarray(symp_op, dtype=bool); symp_gen = np.asarray(symp_gen, dtype=bool)
    g = symp_gen.shape[0]
    red = cref_ordered(np.vstack([symp_gen, symp_op]))
    mask = np.all(~red[g:, g:], axis=1)
    return red[g:, :g].astype(int), mask


# --------------------------------------------------------------------------------------------
# a9: symmetry generators (independent_op.py:90-144, build :124, reduce :125, read-out :126)
# --------------------------------------------------------------------------------------------
def symmetry_generators_symp(symp_h):
    symp_h = np.asarray(symp_h, dtype=bool)
    M, two_n = symp_h.shape
    n = two_n // 2
    stack = np.vstack([np.hstack([symp_h[:, n:], symp_h[:, :n]]), np.eye(two_n, dtype=bool)])
    red = cref_noswap(stack)
    vanish = np.all(~red[:M], axis=0)
    return red[M:, vanish].T


# --------------------------------------------------------------------------------------------
# a7: single-Pauli rotation (base.py:1090-1186), replayed step by step as the reference does
# --------------------------------------------------------------------------------------------
def rotate_by_single_pword(symp, coeff, q_row, angle=None, threshold=1e-18):
    """Returns (rows, coeff) of ``P._rotate_by_single_Pword(Q, angle)`` (Q coefficient taken as 1).
    If every term commutes the input is returned unchanged (base.py:1131-1133)."""
    symp = np.asarray(symp, dtype=bool); coeff = np.asarray(coeff, dtype=complex)
    q_row = np.asarray(q_row, dtype=bool).reshape(1, -1)
    if angle is None:
        angle = np.pi / 2
    angle = float(np.real(angle))
    com = commutes_termwise(symp, q_row).ravel()
    if np.all(com):
        return symp, coeff
    com_rows, com_c = symp[com], coeff[com]
    ac_rows, ac_c = symp[~com], coeff[~com]
    one = np.ones(1, dtype=complex)
    multiple = angle * 2 / np.pi
    k = round(multiple)
    if abs(k - multiple) <= threshold:
        if k % 2 == 0:
            rot_rows, rot_c = ac_rows, ac_c
        else:
            rot_rows, rot_c = mul(ac_rows, ac_c, q_row, one)
            rot_c = rot_c * (-1j)
        if k in (2, 3):                       # NB not reduced mod 4 (base.py:1148)
            rot_c = rot_c * (-1)
        return np.vstack([rot_rows, com_rows]), np.hstack([rot_c, com_c])
    pq_rows, pq_c = mul(ac_rows, ac_c, q_row, one)
    part_rows, part_c = cleanup_op(np.vstack([ac_rows, pq_rows]),
                                   np.hstack([ac_c * np.cos(angle), pq_c * (-1j * np.sin(angle))]))
    return cleanup_op(np.vstack([com_rows, part_rows]), np.hstack([com_c, part_c]))


def perform_rotations(symp, coeff, rotations):
    """base.py:1163-1186: each rotation followed by ``cleanup()``; empty list -> ``cleanup()``."""
    symp = np.asarray(symp, dtype=bool); coeff = np.asarray(coeff, dtype=complex)
    if len(rotations) == 0:
        return cleanup_op(symp, coeff)
    for q_row, angle in rotations:
        symp, coeff = rotate_by_single_pword(symp, coeff, q_row, angle)
        symp, coeff = cleanup_op(symp, coeff)
    return symp, coeff


def lex_order(symp):
    """``sort('lex')`` of base.py:469-470: ``np.lexsort(symp.T)`` (last column is the primary key)."""
    symp = np.asarray(symp, dtype=bool)
    if symp.shape[0] == 0:
        return np.zeros(0, dtype=np.int64)
    return np.lexsort(symp.T)


def check_adjmat_noncontextual(adjmat):
    """utils.py:567-589: terms that do not commute with every term must split into cliques — the unique rows of the adjacency matrix
    restricted to those terms have to be disjoint."""
    adjmat = np.asarray(adjmat, dtype=bool)
    non_universal = np.where(~np.all(adjmat, axis=1))[0]
    unique_rows = np.unique(adjmat[non_universal, :][:, non_universal], axis=0)
    return bool(np.all(np.count_nonzero(unique_rows, axis=0) == 1))


def is_noncontextual(symp):
    """base.py:1074-1088: fewer than four terms are always noncontextual, otherwise the test on the adjacency matrix."""
    symp = np.asarray(symp, dtype=bool)
    if symp.shape[0] < 4:
        return True
    return check_adjmat_noncontextual(commutes_termwise(symp, symp))


def sort_order(symp, coeff, by='magnitude', key='decreasing'):
    """Term order of ``PauliwordOp.sort`` (base.py:455-492): the same NumPy sorts on the same score vectors."""
    symp = np.asarray(symp, dtype=bool)
    n = symp.shape[1] // 2
    X, Z = symp[:, :n].astype(int), symp[:, n:].astype(int)
    if by == 'magnitude':
        order = np.argsort(-abs(np.asarray(coeff)))
    elif by == 'lex':
        order = np.lexsort(symp.T)
    elif by == 'weight':
        order = np.argsort(-np.sum(symp.astype(int), axis=1))
    elif by == 'support':
        occ = np.ascontiguousarray(np.logical_or(symp[:, :n], symp[:, n:]))
        order = np.argsort(occ.view(np.dtype((np.void, occ.dtype.itemsize * occ.shape[1]))).ravel())[::-1]
    elif by == 'Z':
        order = np.argsort(np.sum((n + 1) * X + Z, axis=1))
    elif by == 'X':
        order = np.argsort(np.sum(X + (n + 1) * Z, axis=1))
    elif by == 'Y':
        order = np.argsort(np.sum(abs(X - Z), axis=1))
    else:
        raise ValueError('Only permitted sort by values are magnitude, weight, X, Y or Z')
    if key == 'increasing':
        order = order[::-1]
    elif key != 'decreasing':
        raise ValueError('Only permitted sort by values are increasing or decreasing')
    return order
