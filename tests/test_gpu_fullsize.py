"""GPU: BASELINE.json's full sizes, checked through size-independent properties (round trips, linearity checksums,
cross-kernel consistency) plus exact oracle comparison on sub-blocks the C oracle finishes in seconds."""
import ctypes, os
import numpy as np
import pytest
from symmer_amd import PauliwordOp, IndependentOp, kernels, packing, _lib
from symmer_amd.kernels import DeviceOp
from oracle import oracle_c as oc

pytestmark = pytest.mark.gpu


def dyadic(rng, t):
    return (rng.integers(-8, 9, t) + 1j * rng.integers(-8, 9, t)) / 16.0


def sort_rows(rows, coeff):
    order = np.lexsort(rows.T[::-1])
    return rows[order], coeff[order]


def test_northstar_product_slab_checksums():
    """1,000 qubits, 10^5 x 10^5 terms: one 256-row output slab (2.56e7 pairs, 6.5 GB) — XOR-fold linearity,
    coefficient-sum vs the oracle on the slab's coefficients, exact rows/coefficients on a 3-row sub-slab."""
    n, Ni, No = 1000, 100000, 100000
    A = DeviceOp.random(Ni, n, 0.3, seed=1234); B = DeviceOp.random(No, n, 0.3, seed=99991)
    a_rows, a_c = A.download(); b_rows, b_c = B.download()
    out = DeviceOp.alloc(256 * Ni, 16, with_coeff=True)
    lib = _lib.lib()
    for (o0, o1) in ((0, 256), (99872, 100000), (5000, 5003)):
        _lib.check(lib.symgpu_mul_allpairs_dev(A.handle, B.handle, o0, o1, 1, out.handle))
        assert out.n_terms == (o1 - o0) * Ni
        x, csum = out.checksum()
        expect = np.zeros(32, dtype='<u8')
        if (o1 - o0) % 2:
            expect ^= np.bitwise_xor.reduce(a_rows, axis=0)
        if Ni % 2:
            expect ^= np.bitwise_xor.reduce(b_rows[o0:o1], axis=0)
        assert np.array_equal(x, expect)
        if o1 - o0 == 3:
            rows, coeff = out.download()
            er, ec = oc.mul_allpairs(a_rows, a_c, b_rows[o0:o1], b_c[o0:o1], True)
            assert np.array_equal(rows, er) and np.array_equal(coeff, ec)
            assert abs(csum - ec.sum()) <= 1e-9 * np.abs(ec).sum()
    for h in (A, B, out):
        h.free()


def test_cfg3_square_with_cleanup_consistency():
    """1,000 qubits, 10^4 terms squared (10^8 pairs) + cleanup, dyadic coefficients: the number of surviving terms must be
    1 + #commuting pairs with non-zero doubled coefficient (anticommuting pairs cancel EXACTLY), the coefficient sum is
    conserved exactly, and every surviving row is a product of two input rows."""
    rng = np.random.default_rng(1237)
    n, N = 1000, 10000
    symp = rng.random((N, 2 * n)) < 0.3
    c = dyadic(rng, N); c[c == 0] = 0.5
    A = PauliwordOp(symp, c)
    R = A * A
    adj = A.adjacency_matrix
    n_comm_pairs = (int(adj.sum()) - N) // 2
    assert R.n_terms == 1 + n_comm_pairs
    assert not R.symp_matrix[0].any()                                    # identity first (first occurrence: pair (0,0))
    # conservation: sum of output coefficients == sum over all pairs, computed slab-wise from the uncleaned product
    dA = DeviceOp.upload(A.packed, A.coeff_vec)
    out = DeviceOp.alloc(1000 * N, 16, with_coeff=True)
    total = 0j
    for o0 in range(0, N, 1000):
        _lib.check(_lib.lib().symgpu_mul_allpairs_dev(dA.handle, dA.handle, o0, o0 + 1000, 1, out.handle))
        total += out.checksum()[1]
    assert total == R.coeff_vec.sum()                                    # dyadic: exact
    assert R.coeff_vec[0] == np.sum(c * c)                               # P_i * P_i = I with coefficient c_i^2
    dA.free(); out.free()


def test_cfg3_round3_cleanup_equals_round2_flow_full_size(monkeypatch):
    """BASELINE cfg3 at full size (10^4 terms squared, 10^8 pairs -> 2.5e7 terms): the round-3 cleanup (terms that merge with nothing
    decided in index order before the sort, fused output stage) against the round-2 flow (every term filed from the sorted order,
    batched list + stream stage; SYMGPU_CLEANUP_LAZY=0, SYMGPU_EMIT_FUSED=0): the same rows in the same order with the same
    coefficients, compared chunk by chunk over the whole result — and the same again for the general pair path
    (SYMGPU_CLEANUP_NOSQUARE=1: 10^8 keys) on a 4,000-term operator."""
    lib = _lib.lib()

    def square(op):
        h = ctypes.c_void_p()
        _lib.check(lib.symgpu_mul_cleanup_dev(op.handle, op.handle, 1, 1e-15, 1, ctypes.byref(h)))
        return DeviceOp(h)

    def same(x, y):
        assert x.n_terms == y.n_terms and x.n_terms > 0
        wq = x.info()[1]
        step = 1 << 20
        a = DeviceOp.alloc(step, wq, with_coeff=True); b = DeviceOp.alloc(step, wq, with_coeff=True)
        for o in range(0, x.n_terms, step):
            c = min(step, x.n_terms - o)
            _lib.check(lib.symgpu_op_copy_rows(a.handle, 0, x.handle, o, c)); a.set_rows(c)
            _lib.check(lib.symgpu_op_copy_rows(b.handle, 0, y.handle, o, c)); b.set_rows(c)
            ra, ca = a.download(); rb, cb = b.download()
            assert np.array_equal(ra, rb) and np.array_equal(ca, cb), o
        a.free(); b.free()

    for N, nosquare in ((10000, False), (4000, True)):
        A = DeviceOp.random(N, 1000, 0.3, seed=4242 + N)
        if nosquare: monkeypatch.setenv('SYMGPU_CLEANUP_NOSQUARE', '1')
        new = square(A)
        monkeypatch.setenv('SYMGPU_CLEANUP_LAZY', '0'); monkeypatch.setenv('SYMGPU_EMIT_FUSED', '0')
        old = square(A)
        monkeypatch.delenv('SYMGPU_CLEANUP_LAZY'); monkeypatch.delenv('SYMGPU_EMIT_FUSED')
        if nosquare: monkeypatch.delenv('SYMGPU_CLEANUP_NOSQUARE')
        same(new, old)
        new.free(); old.free(); A.free()


def test_cfg2_rotation_roundtrip():
    """1,000 qubits, 10^5 terms: R(-t) R(t) P == P (rows as a set, coefficients 1e-12); Clifford pi/2 then 3pi/2 == P exactly."""
    rng = np.random.default_rng(1236)
    n, N = 1000, 100000
    P = PauliwordOp(rng.random((N, 2 * n)) < 0.3, dyadic(rng, N)).cleanup()
    Q = PauliwordOp((rng.random(2 * n) < 0.3).reshape(1, -1), [1])
    fwd = P._rotate_by_single_Pword(Q, 0.3)
    assert P.n_terms < fwd.n_terms <= 2 * P.n_terms
    n_anti = int((~P.commutes_termwise(Q)).sum())
    assert fwd.n_terms == P.n_terms + n_anti                              # no accidental merges at n=1000
    back = fwd._rotate_by_single_Pword(Q, -0.3)
    r1, c1 = sort_rows(back.packed, back.coeff_vec); r0, c0 = sort_rows(P.packed, P.coeff_vec)
    assert np.array_equal(r1, r0) and np.allclose(c1, c0, rtol=0, atol=1e-12)
    cl = P._rotate_by_single_Pword(Q, np.pi / 2)._rotate_by_single_Pword(Q, 3 * np.pi / 2)
    r1, c1 = sort_rows(cl.packed, cl.coeff_vec)
    assert np.array_equal(r1, r0) and np.array_equal(c1, c0)
    chain = P.perform_rotations([(Q, 0.3), (Q, -0.3)])
    r1, c1 = sort_rows(chain.packed, chain.coeff_vec)
    assert np.array_equal(r1, r0) and np.allclose(c1, c0, rtol=0, atol=1e-12)


def test_cfg4_symmetry_generators_full_size():
    """2,000 qubits x 50,000 terms with 32 planted symmetries, scrambled by 16 Clifford rotations: exactly 32 generators,
    each commuting with every term, mutually independent; [4000 x 54000] elimination, ~8e6 reference row-XORs."""
    rng = np.random.default_rng(1238)
    n, M, k = 2000, 50000, 32
    symp = rng.random((M, 2 * n)) < 0.3
    symp[:, :k] = False
    H = PauliwordOp(symp, np.ones(M))
    rots = [(PauliwordOp((rng.random(2 * n) < 0.3).reshape(1, -1), [1]), np.pi / 2) for _ in range(16)]
    H = H.perform_rotations(rots)
    assert H.n_terms == M
    rows, n_xor = kernels.symmetry_kernel(H.packed, n)
    assert rows.shape[0] == k
    assert 7_000_000 < n_xor < 9_000_000
    S = IndependentOp.symmetry_generators(H, commuting_override=True)     # constructor re-checks independence on device
    assert S.n_terms == k and np.all(S.commutes_termwise(H)) and np.all(S.adjacency_matrix)
    # the planted generators Z_0..Z_31, pushed through the same rotations, span the same space
    planted = np.zeros((k, 2 * n), dtype=bool); planted[np.arange(k), n + np.arange(k)] = True
    Z = PauliwordOp(planted, np.ones(k)).perform_rotations(rots)
    _, mask = Z.generator_reconstruction(S)
    assert np.all(mask)


def test_cfg4_elimination_schedules_agree_full_size(monkeypatch):
    """BASELINE cfg4's matrix size (4000 x 54000, dense) and a sparse one of the same shape: the default schedule (selectors inside
    phase 0's launch, two-word panel window, full-row panel for sparse rows) against its two fall-backs, the separate-launch schedule
    (after a time-out) and the flag-per-row sweep (LDS attribute refused) — reduced matrix, pivots and reference row-XOR count identical."""
    rng = np.random.default_rng(77)
    for dens in (0.3, 0.0005):
        m = rng.random((4000, 54000)) < dens
        packed = packing.pack_bits(m)
        ref = kernels.rref(packed, want_pivots=True)
        for env in ({'SYMGPU_GF2_FUSED_SELECT': '0'}, {'SYMGPU_GF2_M4R': '0'}):
            for k, v in env.items(): monkeypatch.setenv(k, v)
            got = kernels.rref(packed, want_pivots=True)
            for k in env: monkeypatch.delenv(k)
            assert np.array_equal(ref[0], got[0]) and ref[1] == got[1] and np.array_equal(ref[2], got[2]), (dens, env)
        assert ref[1] > 0


# ---- round 4: the ORACLE at the sizes where the size-gated device paths run ----------------------------
def _cfg4_bench_operator(seed=1238):
    """bench.py's cfg4 operator (cfg4_operator): 2,000 qubits, 50,000 terms, 32 planted symmetries, 16 Clifford rotations."""
    rng = np.random.default_rng(seed)
    symp = rng.random((50000, 4000)) < 0.3
    symp[:, :32] = False
    H = DeviceOp.upload(packing.pack_rows(symp), np.ones(50000, dtype=complex))
    del symp
    for _ in range(16):
        q = packing.pack_rows((rng.random((1, 4000)) < 0.3))[0]
        res, allc = kernels.rotate_single_dev(H, q, np.pi / 2)
        if not allc:
            H.free(); H = res
    return H


def test_cfg4_full_size_against_the_c_oracle():
    """BASELINE cfg4 on the bench's own operator, against the C oracle (utils.py:292-315, independent_op.py:124-126):
    (1) symmetry_kernel: the 32 generators (rows and order) and the reference-order row-XOR count;
    (2) rref of the [4000 x 54000] matrix that symmetry_generators reduces: reduced matrix, pivot columns and XOR count — the device
    runs its default blocked schedule (in-launch barrier, look-ahead panel), the oracle the reference's row-by-row loop."""
    H = _cfg4_bench_operator()
    h_rows = H.download(with_coeff=False)
    H.free()
    n, M = 2000, 50000
    gen, n_xor = kernels.symmetry_kernel(h_rows, n)
    egen, e_xor = oc.symmetry_generators(h_rows, n)
    assert gen.shape == egen.shape == (32, 64) and np.array_equal(gen, egen)
    assert n_xor == e_xor and 7_000_000 < n_xor < 9_000_000
    # the matrix of independent_op.py:124: columns = [Z | X] of every term, then the identity; its transpose is row-reduced
    symp = packing.unpack_rows(h_rows, n)
    mat = np.hstack([np.hstack([symp[:, n:], symp[:, :n]]).T, np.eye(2 * n, dtype=bool)])
    assert mat.shape == (4000, 54000)
    packed = packing.pack_bits(mat)
    del symp, mat
    got = kernels.rref(packed, want_pivots=True)
    exp = oc.rref(packed, want_pivots=True)
    assert np.array_equal(got[0], exp[0]), 'reduced matrix differs from the oracle'
    assert got[1] == exp[1] == n_xor, 'reference-order XOR count differs'
    assert np.array_equal(got[2], exp[2]), 'pivot columns differ'


def test_cfg2_full_size_rotations_against_the_oracle():
    """BASELINE cfg2 on the bench's own operator (100,000 terms, 1,000 qubits) against the NumPy oracle of base.py:1090-1161: a
    non-Clifford and a Clifford rotation on the one-launch kernel (rows resident in LDS), then a non-Clifford rotation of the
    ~150,000-term result (the one-launch kernel's register-spill form, above its LDS capacity) — rows, row order, coefficients."""
    from oracle import oracle_np as onp
    n, N = 1000, 100000
    rng = np.random.default_rng(1236)
    P = DeviceOp.random(N, n, 0.3, seed=1236)
    qs = [(rng.random(2 * n) < 0.3) for _ in range(3)]
    qp = [packing.pack_rows(q.reshape(1, -1))[0] for q in qs]
    clean = kernels.cleanup_dev(P)                                         # duplicate status known: the one-launch kernel takes it
    P.free()
    rows0, c0 = clean.download()
    symp0 = packing.unpack_rows(rows0, n)
    cnt = ctypes.c_int64(0)

    def one_launch_count():
        _lib.check(_lib.lib().symgpu_debug_counter(1, ctypes.addressof(cnt)))
        return cnt.value

    def check(dev_in, symp_in, c_in, q, qpk, ang, exact):
        before = one_launch_count()
        res, allc = kernels.rotate_single_dev(dev_in, qpk, ang)
        assert not allc
        took = one_launch_count() - before
        r, c = res.download()
        er, ec = onp.rotate_by_single_pword(symp_in, c_in, q, ang)
        assert r.shape[0] == er.shape[0], (r.shape, er.shape)
        assert np.array_equal(r, packing.pack_rows(er)), 'rows / row order differ from the oracle'
        if exact:
            assert np.array_equal(c, ec)
        else:
            assert np.allclose(c, ec, rtol=0, atol=1e-12)
        return res, er, ec, took

    res1, er1, ec1, took = check(clean, symp0, c0, qs[0], qp[0], 0.3, False)
    assert took == 1, 'the one-launch rotation kernel did not run at BASELINE size'
    assert 1.4 * N < res1.n_terms < 1.6 * N
    res2, _, _, took = check(clean, symp0, c0, qs[1], qp[1], np.pi / 2, True)
    assert took == 1
    res2.free()
    # second rotation, of the 1.5e5-term result: above the kernel's LDS capacity at 1,000 qubits (rows spill to registers)
    res3, er3, _, took3 = check(res1, er1, ec1, qs[2], qp[2], -1.1, False)
    assert took3 == 1, 'the register-spill form of the one-launch kernel did not take the 150,000-term operator'
    assert res3.n_terms > 2.1 * N
    for h in (res1, res3, clean):
        h.free()
    # and through the drop-in API (fresh upload, duplicate status unknown: the multi-launch path with its own duplicate check)
    A = PauliwordOp._from_packed(rows0, n, c0)
    R = A._rotate_by_single_Pword(PauliwordOp(qs[0].reshape(1, -1), [1]), 0.3)
    assert np.array_equal(R.packed, packing.pack_rows(er1)) and np.allclose(R.coeff_vec, ec1, rtol=0, atol=1e-12)


@pytest.mark.parametrize('n,N', [(2000, 100000), (1000, 400000), (3000, 100000), (3000, 8000)])
def test_one_launch_rotation_with_the_rows_left_in_memory(n, N, monkeypatch):
    """Operators beyond the 38 MB of LDS + registers the one-launch rotation kernel holds (round 6: 1e5 terms of 2,000 qubits, 4e5 terms of
    1,000 qubits; 3,000 qubits: rows of 94 words, not a power-of-two number of chunks — analysed from memory — and, at 8,000 terms, rows of
    more than 64 words resident in LDS): only the per-row state stays on the chip, the rows are read again when they are written.  A non-Clifford and a Clifford
    rotation, then a non-Clifford rotation of the grown result: the one-launch kernel takes them (debug counter), rows / order / coefficients
    equal the NumPy oracle of base.py:1090-1161 and, bit for bit, the multi-launch path (SYMGPU_ROT_RESIDENT=0)."""
    from oracle import oracle_np as onp
    rng = np.random.default_rng(2026 + n)
    P = DeviceOp.random(N, n, 0.3, seed=77 + n)
    clean = kernels.cleanup_dev(P)
    P.free()
    rows0, c0 = clean.download()
    symp0 = packing.unpack_rows(rows0, n)
    cnt = ctypes.c_int64(0)

    def one_launch_count():
        _lib.check(_lib.lib().symgpu_debug_counter(1, ctypes.addressof(cnt)))
        return cnt.value

    q1 = rng.random(2 * n) < 0.3; q2 = rng.random(2 * n) < 0.3
    for q, ang, exact in ((q1, 0.3, False), (q2, np.pi / 2, True)):
        qpk = packing.pack_rows(q.reshape(1, -1))[0]
        before = one_launch_count()
        res, allc = kernels.rotate_single_dev(clean, qpk, ang)
        assert not allc and one_launch_count() - before == 1, 'the one-launch kernel did not take the operator'
        r, c = res.download()
        monkeypatch.setenv('SYMGPU_ROT_RESIDENT', '0')
        res_m, _ = kernels.rotate_single_dev(clean, qpk, ang)
        monkeypatch.delenv('SYMGPU_ROT_RESIDENT')
        rm, cm = res_m.download()
        res_m.free()
        assert np.array_equal(r, rm) and np.array_equal(c.view(np.uint64), cm.view(np.uint64)), 'differs from the multi-launch path'
        er, ec = onp.rotate_by_single_pword(symp0, c0, q, ang)
        assert np.array_equal(r, packing.pack_rows(er)), 'rows / row order differ from the oracle'
        assert np.array_equal(c, ec) if exact else np.allclose(c, ec, rtol=0, atol=1e-12)
        if not exact:                                             # the grown operator (hashes handed on by the kernel) once more
            q3 = rng.random(2 * n) < 0.3
            before = one_launch_count()
            res2, _ = kernels.rotate_single_dev(res, packing.pack_rows(q3.reshape(1, -1))[0], -0.7)
            assert one_launch_count() - before == 1
            r2, c2 = res2.download()
            er2, ec2 = onp.rotate_by_single_pword(er, ec, q3, -0.7)
            assert np.array_equal(r2, packing.pack_rows(er2)) and np.allclose(c2, ec2, rtol=0, atol=1e-12)
            res2.free()
        res.free()
    clean.free()


@pytest.mark.parametrize('shape', ['squared 3000', 'general 2500x2000', 'general 2000x2500'])
def test_product_cleanup_over_the_lazy_gate_against_the_c_oracle(shape):
    """Product + cleanup with more than 2^22 keys on the DEFAULT path (the lazy flow of cleanup.hip switches on there by itself;
    round 3 only compared it with the forced gate or with the round-2 flow), 100 qubits, against the C oracle (utils.py:230-279):
    a squared operator (4.5e6 keys of its 9e6 pairs) and general products in both operand orders (5e6 keys), dyadic coefficients:
    rows, first-occurrence order and coefficients bit for bit."""
    rng = np.random.default_rng(2200 + len(shape))
    n = 100
    if shape.startswith('squared'):
        A = PauliwordOp(rng.random((3000, 2 * n)) < 0.3, dyadic(rng, 3000)); B = A
    else:
        na, nb = (2500, 2000) if shape.endswith('2500x2000') else (2000, 2500)
        A = PauliwordOp(rng.random((na, 2 * n)) < 0.3, dyadic(rng, na)); B = PauliwordOp(rng.random((nb, 2 * n)) < 0.3, dyadic(rng, nb))
    R = A * B
    er, ec = oc.mul(A.packed, A.coeff_vec, B.packed, B.coeff_vec)
    assert R.n_terms == er.shape[0] > (1 << 21)
    assert np.array_equal(R.packed, er), 'rows / first-occurrence order differ from the oracle'
    assert np.array_equal(R.coeff_vec, ec)


@pytest.mark.parametrize('mode', ['default', 'sorted flag pass', 'full sort', 'give up', 'repeated rows', 'few rows', 'planted'])
def test_cleanup_flagged_key_flow_switches(mode, monkeypatch):
    """The cleanup of products — the pairs that share a key found from the bucketed operand hash tables (round 6, pair_dups.hip) or, with
    SYMGPU_CLEANUP_DIRECT=0 ('sorted flag pass') and for operands that do not fit a workgroup's LDS, behind a partial sort (round 4-5,
    k_find_suspects), only the flagged keys sorted completely — against the C oracle above
    the 2^22-key gate: the default, the full sort of all keys (SYMGPU_CLEANUP_SUSPECTS=0), the flow giving up after the flag pass and
    finishing the last radix pass on the whole array (forced, and reached by itself on operands full of repeated rows, where nearly
    every key has a partner; 'few rows': 3 x 2 distinct products, runs of ~10^6 equal keys — the flag pass itself gives up on a run it cannot
    read to its end; 'planted': rows that are products of two other rows, so that a few hundred product rows occur two or three times)."""
    rng = np.random.default_rng(4404)
    n, na, nb = 100, 2600, 2100
    if mode == 'full sort': monkeypatch.setenv('SYMGPU_CLEANUP_SUSPECTS', '0')
    if mode == 'give up': monkeypatch.setenv('SYMGPU_CLEANUP_SUSPECTS', '2')
    if mode == 'sorted flag pass': monkeypatch.setenv('SYMGPU_CLEANUP_DIRECT', '0')
    sa = rng.random((na, 2 * n)) < 0.3; sb = rng.random((nb, 2 * n)) < 0.3
    if mode == 'planted':                          # a few hundred products that occur twice or three times: rows that are products of two others
        for k in range(150):
            i, j, l = rng.integers(0, na, 3)
            sa[l] = sa[i] ^ sa[j]
            i, j, l = rng.integers(0, nb, 3)
            sb[l] = sb[i] ^ sb[j]
        sa = np.unique(sa, axis=0); sb = np.unique(sb, axis=0)
        na, nb = sa.shape[0], sb.shape[0]
    if mode == 'repeated rows':
        sa = sa[rng.integers(0, 400, na)]; sb = sb[rng.integers(0, 300, nb)]          # 400 x 300 distinct products, each ~45 times
    if mode == 'few rows':
        sa = sa[rng.integers(0, 3, na)]; sb = sb[rng.integers(0, 2, nb)]
    A = PauliwordOp(sa, dyadic(rng, na)); B = PauliwordOp(sb, dyadic(rng, nb))
    for X, Y in ((A, B), (A, A)):
        R = X * Y
        er, ec = oc.mul(X.packed, X.coeff_vec, Y.packed, Y.coeff_vec)
        assert np.array_equal(R.packed, er) and np.array_equal(R.coeff_vec, ec), mode


@pytest.mark.parametrize('planted', [False, True])
@pytest.mark.parametrize('shape', ['squared 2300', 'general 1900x1500'])
def test_marking_from_bytes_equals_marking_from_keys(shape, planted, monkeypatch):
    """Where the flag pass works from the operand hash tables the key kernel writes one byte per pair and k_mark_bytes marks the single terms
    from those (round 6): the same rows, order and coefficient bits as the 8-byte keys + k_mark_singles (SYMGPU_CLEANUP_KEYBYTES=0), as the
    sorted flag pass (SYMGPU_CLEANUP_DIRECT=0), and with every coefficient looked at (SYMGPU_CLEANUP_NOFLOOR=1: the per-pair decisions of
    k_mark_bytes) — O(1) Gaussian coefficients, and planted tiny ones whose products fall under the threshold; against the C oracle with the
    Gaussian rule of the parity tests."""
    rng = np.random.default_rng(515 + planted + len(shape))
    n = 64
    N, M = (2300, 2300) if shape.startswith('squared') else (1900, 1500)
    ca = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    if planted:
        ca[rng.integers(0, N, 60)] *= 1e-9
    sa = rng.random((N, 2 * n)) < 0.3
    for k in range(40):                                       # product rows that occur twice
        i, j, l = rng.integers(0, N, 3)
        sa[l] = sa[i] ^ sa[j]
    sa = np.unique(sa, axis=0); N = sa.shape[0]; ca = ca[:N]
    A = PauliwordOp(sa, ca)
    if shape.startswith('squared'):
        B = A
    else:
        cb = rng.standard_normal(M) + 1j * rng.standard_normal(M)
        if planted:
            cb[rng.integers(0, M, 60)] *= 1e-9
        B = PauliwordOp(rng.random((M, 2 * n)) < 0.3, cb)
    for thr in (1e-15, 1e-12):
        ref = kernels.mul_cleanup(A.packed, A.coeff_vec, B.packed, B.coeff_vec, True, thr)
        for env in ({'SYMGPU_CLEANUP_KEYBYTES': '0'}, {'SYMGPU_CLEANUP_DIRECT': '0'}, {'SYMGPU_CLEANUP_NOFLOOR': '1'}, {'SYMGPU_CLEANUP_NOFLOOR': '1', 'SYMGPU_CLEANUP_KEYBYTES': '0'}):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            got = kernels.mul_cleanup(A.packed, A.coeff_vec, B.packed, B.coeff_vec, True, thr)
            for k in env:
                monkeypatch.delenv(k)
            assert got[0].shape == ref[0].shape and np.array_equal(got[0], ref[0]) and np.array_equal(got[1].view(np.uint64), ref[1].view(np.uint64)), (env, thr)
    er, ec = oc.mul(A.packed, A.coeff_vec, B.packed, B.coeff_vec)
    R = A * B
    k1 = np.abs(R.coeff_vec) > 1e-12; k2 = np.abs(ec) > 1e-12
    assert np.array_equal(R.packed[k1], er[k2]) and np.allclose(R.coeff_vec[k1], ec[k2], rtol=0, atol=1e-12)


def test_squared_product_with_runs_of_more_than_a_thousand_keys(monkeypatch):
    """The flag pass of the cleanup (k_find_suspects) works on runs of keys that agree in 16 sorted hash bits: 763 keys on average at
    cfg3, up to 2,048 (products of 2^27 keys) before the library sorts completely instead.  12,000 terms squared = 7.2e7 keys, runs of
    ~1,100 keys that mostly need a second extension step: same rows, order and coefficients as the complete sort of all keys
    (SYMGPU_CLEANUP_SUSPECTS=0), and the count of kept terms is 1 + the commuting pairs (dyadic coefficients, nothing cancels by chance)."""
    rng = np.random.default_rng(8128)
    n, N = 100, 12000
    c = dyadic(rng, N); c[c == 0] = 0.5
    A = PauliwordOp(rng.random((N, 2 * n)) < 0.3, c)
    fast = A * A
    monkeypatch.setenv('SYMGPU_CLEANUP_SUSPECTS', '0')
    full = A * A
    monkeypatch.delenv('SYMGPU_CLEANUP_SUSPECTS')
    assert fast.n_terms == full.n_terms > 3.5e7
    assert np.array_equal(fast.packed, full.packed) and np.array_equal(fast.coeff_vec, full.coeff_vec)
    adj = A.adjacency_matrix
    assert fast.n_terms == 1 + (int(adj.sum()) - N) // 2


@pytest.mark.parametrize('coeffs', ['gaussian', 'dyadic'])
def test_cfg3_full_size_against_the_oracle(coeffs):
    """BASELINE cfg3 at full size — a 10,000-term, 1,000-qubit operator squared (10^8 pairs) + cleanup — against the reference's
    result assembled from the C oracle's 10^8 pair coefficients (tests/_expected.py, pinned to oracle_c.mul on the CPU: identity
    first, then the pairs o < i in index order with the two twins' coefficients added; utils.py:230-279): the term count, EVERY
    coefficient in output order bit for bit (Gaussian coefficients are all distinct, so that also fixes which pair every output
    row is), every row of five 2^20-row chunks bit for bit, and the XOR of all rows."""
    from _expected import squared_expected
    rng = np.random.default_rng(1237)
    n, N = 1000, 10000
    rows = packing.pack_rows(rng.random((N, 2 * n)) < 0.3)
    c = (rng.standard_normal(N) + 1j * rng.standard_normal(N)) if coeffs == 'gaussian' else dyadic(rng, N)
    A = DeviceOp.upload(rows, c)
    h = ctypes.c_void_p()
    _lib.check(_lib.lib().symgpu_mul_cleanup_dev(A.handle, A.handle, 1, 1e-15, 1, ctypes.byref(h)))
    R = DeviceOp(h)
    o, i, ec = squared_expected(rows, c)
    assert R.n_terms == ec.shape[0] and 2.4e7 < ec.shape[0] < 2.6e7
    x, _ = R.checksum()
    parity = (np.bincount(o[1:], minlength=N) + np.bincount(i[1:], minlength=N)) & 1          # the identity term contributes nothing
    assert np.array_equal(x, np.bitwise_xor.reduce(rows[parity.astype(bool)], axis=0)), 'XOR of all output rows'
    step = 1 << 20
    part = DeviceOp.alloc(step, 16, with_coeff=True)
    n_chunks = (R.n_terms + step - 1) // step
    row_chunks = {0, 1, n_chunks // 2, n_chunks - 2, n_chunks - 1}
    for k in range(n_chunks):
        lo = k * step; cnt = min(step, R.n_terms - lo)
        _lib.check(_lib.lib().symgpu_op_copy_rows(part.handle, 0, R.handle, lo, cnt)); part.set_rows(cnt)
        gr, gc = part.download()
        assert np.array_equal(gc, ec[lo:lo + cnt]), f'coefficients of chunk {k}'
        if k in row_chunks:
            assert np.array_equal(gr, rows[i[lo:lo + cnt]] ^ rows[o[lo:lo + cnt]]), f'rows of chunk {k}'
    for hnd in (part, R, A):
        hnd.free()


def test_cfg5_adjacency_slice_full_width():
    """2,000 qubits, 200,000 terms: a 4096-row block of the adjacency matrix against all terms — symmetric on the square
    sub-block, True on the diagonal, exact vs the C oracle on a 64 x 8192 corner, and byte count == bit-packed popcount."""
    n, N, rows = 2000, 200000, 4096
    A = DeviceOp.random(N, n, 0.3, seed=1239)
    lib = _lib.lib()
    buf = ctypes.c_void_p(); bits = ctypes.c_void_p()
    _lib.check(lib.symgpu_dev_alloc(rows * N, ctypes.byref(buf)))
    _lib.check(lib.symgpu_dev_alloc(rows * ((N + 63) // 64) * 8, ctypes.byref(bits)))
    _lib.check(lib.symgpu_commutes_dev(A.handle, 0, rows, A.handle, buf))
    _lib.check(lib.symgpu_commutes_bits_dev(A.handle, 0, rows, A.handle, bits))
    out = np.empty((rows, N), dtype=np.uint8)
    _lib.check(lib.symgpu_dev_download(buf, out.ctypes.data, out.nbytes))
    assert set(np.unique(out)) <= {0, 1}
    sq = out[:, :rows]
    assert np.array_equal(sq, sq.T) and np.all(np.diag(sq) == 1)
    a_rows = A.download(with_coeff=False)
    assert np.array_equal(out[:64, :8192].astype(bool), oc.commutes(a_rows[:64], a_rows[:8192]))
    s_bytes, s_bits = ctypes.c_uint64(0), ctypes.c_uint64(0)
    _lib.check(lib.symgpu_dev_checksum_u8(buf, rows * N, ctypes.addressof(s_bytes)))
    _lib.check(lib.symgpu_dev_popcount_u64(bits, rows * ((N + 63) // 64), ctypes.addressof(s_bits)))
    assert s_bytes.value == int(out.sum()) == s_bits.value
    assert 0.45 < s_bytes.value / (rows * N) < 0.55
    _lib.check(lib.symgpu_dev_free(buf)); _lib.check(lib.symgpu_dev_free(bits)); A.free()


def test_large_result_arrives_whole_through_the_api():
    """A commutation table of more than 2 GiB comes to the host in 1 GiB pieces whose pages are touched while the previous piece travels
    (context.hip download_pipelined): `P.commutes_termwise(P)` of 52,000 terms (2.7 GB) through the drop-in class
    equals the same table fetched band by band in copies below the gate, and random blocks equal the oracle's."""
    from symmer_amd import PauliwordOp
    lib = _lib.lib()
    n, T = 300, 52000
    A = DeviceOp.random(T, n, 0.3, seed=556)
    rows = A.download(with_coeff=False)
    P = PauliwordOp._from_device(A, n)
    C = P.commutes_termwise(P)
    assert C.shape == (T, T) and C.dtype == np.bool_ and C.nbytes > (2 << 30)
    buf = ctypes.c_void_p()
    _lib.check(lib.symgpu_dev_alloc(T * T, ctypes.byref(buf)))
    _lib.check(lib.symgpu_commutes_dev(A.handle, 0, T, A.handle, buf))
    band = 13000                                                          # 676 MB per copy: the single-call path
    part = np.empty((band, T), dtype=np.uint8)
    for r0 in range(0, T, band):
        _lib.check(lib.symgpu_dev_download(ctypes.c_void_p(buf.value + r0 * T), part.ctypes.data, band * T))
        assert np.array_equal(C[r0:r0 + band].view(np.uint8), part), r0
    _lib.check(lib.symgpu_dev_free(buf))
    rng = np.random.default_rng(10)
    for _ in range(8):
        r0, c0 = (int(v) for v in rng.integers(0, T - 300, 2))
        assert np.array_equal(C[r0:r0 + 300, c0:c0 + 300], oc.commutes(rows[r0:r0 + 300], rows[c0:c0 + 300]))


def test_bench_line_schema_small_workload():
    """bench.py prints ONE JSON line with the contract's keys (tiny workload; the roofline / cpu objects are present)."""
    import json, sys, io, contextlib, importlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    bench = importlib.import_module('bench')
    argv, buf = sys.argv, io.StringIO()
    sys.argv = ['bench.py', '--gpus', '1', '--steps', '2', '--warmup', '1', '--left-terms', '4000', '--right-terms', '3000', '--qubits', '200',
                '--no-extras', '--no-cpu']
    try:
        with contextlib.redirect_stdout(buf):                      # in-process: no exec from a process that holds the GPU
            bench.main()
    finally:
        sys.argv = argv

    class out:                                                      # same shape as a CompletedProcess for the checks below
        stdout = buf.getvalue()
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data',
                'config', 'roofline'):
        assert key in d, key
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['warmup'] == 1 and d['higher_is_better'] is True and d['value'] > 0
    assert d['config']['workload'] == 'allpairs_product' and d['config']['pairs_per_step'] == 4000 * 3000
    r = d['roofline']
    for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert key in r, key
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9
    # the line ENDS with a compact `summary` (the driver keeps the tail of stdout), the api legend appears once
    assert list(d)[-1] == 'summary' and list(d)[-2] == 'api_legend' and lines[0].count('first_call = fresh host arrays') == 1
    sm = d['summary']
    assert sm['product_1e5x1e5']['k'][0] == r['kernel'] and abs(sm['product_1e5x1e5']['k'][1] - r['frac']) < 5e-3
    assert len(json.dumps(sm)) < 1500


def test_bench_adjacency_workload_schema():
    """`bench.py --workload adjacency` (BASELINE cfg5 shape, reduced): one JSON line, strong scaling, T*T pairs per step."""
    import json, sys, io, contextlib, importlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    bench = importlib.import_module('bench')
    argv, buf = sys.argv, io.StringIO()
    sys.argv = ['bench.py', '--gpus', '1', '--steps', '2', '--warmup', '1', '--workload', 'adjacency', '--adj-terms', '30000', '--adj-qubits', '300',
                '--adj-slab-rows', '25000']
    try:
        with contextlib.redirect_stdout(buf):
            bench.main()
    finally:
        sys.argv = argv
    lines = [l for l in buf.getvalue().splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['scaling'] == 'strong' and d['config']['workload'] == 'commutes_termwise_adjacency' and d['config']['pairs_per_step'] == 30000 ** 2
    assert d['value'] > 0 and d['roofline']['bound'] == 'hbm' and d['roofline']['launches'] == 2 * 2      # two 25,000-row slabs per step
    assert 0 < d['roofline']['lds']['frac'] < 1


def _bit_column_counts(rows):
    """Number of set bits per bit column of packed rows uint64[T, W] -> int64[W * 64]."""
    bits = np.unpackbits(rows.view(np.uint8), axis=1, bitorder='little')
    return bits.sum(axis=0, dtype=np.int64)


def test_northstar_product_every_slab_of_the_bench_step():
    """What `bench.py` (default workload) runs — ALL 391 slabs of the 1,000-qubit 10^5 x 10^5 product, 256 outer rows each — checked
    through size-independent properties computed in O(N + M) from the operands:
      * popcount of every slab: sum_{i,o} |a_i ^ b_o| = sum_bits [cnt_a (n_o - cnt_b) + (N - cnt_a) cnt_b]   (exact, per slab)
      * XOR fold of every slab (linearity): 0 for even row counts, else the fold of the other operand
      * sum of the slab's coefficients against the C oracle's on 16 random slabs (phases included), and rows + coefficients
        bit for bit on a 2-row piece of each of them."""
    lib = _lib.lib()
    n, Ni, No, slab = 1000, 100000, 100000, 256
    A = DeviceOp.random(Ni, n, 0.3, seed=1234); B = DeviceOp.random(No, n, 0.3, seed=99991)
    a_rows, a_c = A.download(); b_rows, b_c = B.download()
    W = a_rows.shape[1]
    cnt_a = _bit_column_counts(a_rows)
    fold_a = np.bitwise_xor.reduce(a_rows, axis=0)
    out = DeviceOp.alloc(slab * Ni, W // 2, with_coeff=True)
    t, wq, cap = out.info()
    rng = np.random.default_rng(5)
    sampled = set(rng.choice((No + slab - 1) // slab, 16, replace=False).tolist())
    pop = ctypes.c_uint64(0)
    rows_ptr = ctypes.c_void_p()
    for k, o0 in enumerate(range(0, No, slab)):
        o1 = min(No, o0 + slab)
        _lib.check(lib.symgpu_mul_allpairs_dev(A.handle, B.handle, o0, o1, 1, out.handle))
        assert out.n_terms == (o1 - o0) * Ni
        cnt_b = _bit_column_counts(b_rows[o0:o1])
        expect_pop = int(np.sum(cnt_a * ((o1 - o0) - cnt_b) + (Ni - cnt_a) * cnt_b))
        _lib.check(lib.symgpu_op_popcount(out.handle, ctypes.addressof(pop)))
        assert pop.value == expect_pop, (k, pop.value, expect_pop)
        x, csum = out.checksum()
        expect = np.zeros(W, dtype='<u8')
        if (o1 - o0) % 2:
            expect ^= fold_a
        if Ni % 2:
            expect ^= np.bitwise_xor.reduce(b_rows[o0:o1], axis=0)
        assert np.array_equal(x, expect), k
        if k in sampled:
            ec = oc.mul_allpairs_coeff(a_rows, a_c, b_rows[o0:o1], b_c[o0:o1], True)
            assert abs(csum - ec.sum()) <= 1e-9 * np.abs(ec).sum(), k
    # exact rows + coefficients of a 2-row piece of 16 random places
    for o0 in rng.choice(No - 2, 16, replace=False).tolist():
        _lib.check(lib.symgpu_mul_allpairs_dev(A.handle, B.handle, o0, o0 + 2, 1, out.handle))
        rows, coeff = out.download()
        er, ec = oc.mul_allpairs(a_rows, a_c, b_rows[o0:o0 + 2], b_c[o0:o0 + 2], True)
        assert np.array_equal(rows, er) and np.array_equal(coeff, ec)
    for h in (A, B, out):
        h.free()


def test_cfg5_whole_adjacency_one_launch():
    """What `bench.py --workload adjacency` runs on one GPU — the 200,000 x 200,000 adjacency of a 2,000-qubit operator in ONE
    launch into a 40 GB np.bool_ result:
      * the byte sum of every 10,000-row band equals the popcount of the same band computed a second time as BIT-PACKED rows
        (another epilogue of the kernel) — and both lie near one half
      * symmetry: 24 sampled 256 x 256 blocks equal the transposes of their mirror blocks, diagonal blocks have a unit diagonal
      * 16 random 256 x 256 blocks are exactly the C oracle's commutes() of the corresponding rows."""
    lib = _lib.lib()
    n, T = 2000, 200000
    free_b, total_b = ctypes.c_int64(0), ctypes.c_int64(0)
    _lib.check(lib.symgpu_mem_info(ctypes.addressof(free_b), ctypes.addressof(total_b)))
    if free_b.value < 60 * (1 << 30):
        pytest.skip('needs 60 GB of free HBM')
    A = DeviceOp.random(T, n, 0.3, seed=555)
    a_rows = A.download(with_coeff=False)
    buf = ctypes.c_void_p()
    _lib.check(lib.symgpu_dev_alloc(T * T, ctypes.byref(buf)))
    _lib.check(lib.symgpu_commutes_dev(A.handle, 0, T, A.handle, buf))
    band = 10000
    wpr = (T + 63) // 64
    bits = ctypes.c_void_p()
    _lib.check(lib.symgpu_dev_alloc(band * wpr * 8, ctypes.byref(bits)))
    s_bytes, s_bits = ctypes.c_uint64(0), ctypes.c_uint64(0)
    total = 0
    for r0 in range(0, T, band):
        _lib.check(lib.symgpu_dev_checksum_u8(ctypes.c_void_p(buf.value + r0 * T), band * T, ctypes.addressof(s_bytes)))
        _lib.check(lib.symgpu_commutes_bits_dev(A.handle, r0, r0 + band, A.handle, bits))
        _lib.check(lib.symgpu_dev_popcount_u64(bits, band * wpr, ctypes.addressof(s_bits)))
        assert s_bytes.value == s_bits.value, r0
        total += s_bytes.value
    assert 0.49 < total / (T * T) < 0.51

    def block(r0, c0, h=256, w=256):
        out = np.empty((h, w), dtype=np.uint8)
        for k in range(h):
            _lib.check(lib.symgpu_dev_download(ctypes.c_void_p(buf.value + (r0 + k) * T + c0), out[k].ctypes.data, w))
        return out
    rng = np.random.default_rng(9)
    for _ in range(24):
        r0, c0 = (int(v) for v in rng.integers(0, T - 256, 2))
        assert np.array_equal(block(r0, c0), block(c0, r0).T)
    for d in (0, 77777, T - 256):
        assert np.all(np.diag(block(d, d)) == 1)
    for _ in range(16):
        r0, c0 = (int(v) for v in rng.integers(0, T - 256, 2))
        assert np.array_equal(block(r0, c0).astype(bool), oc.commutes(a_rows[r0:r0 + 256], a_rows[c0:c0 + 256]))
    _lib.check(lib.symgpu_dev_free(buf)); _lib.check(lib.symgpu_dev_free(bits)); A.free()


@pytest.mark.parametrize('workload,metric,kernel', [('mul_cleanup', 'pauli_term_pairs_per_sec', 'k_emit_fused'),
                                                    ('rotation', 'pauli_term_pairs_per_sec', 'k_rot_resident'),
                                                    ('gf2', 'gf2_row_xors_per_sec', 'k_sweep_m4r')])
def test_bench_workloads_of_the_other_baseline_configs(workload, metric, kernel):
    """`bench.py --workload mul_cleanup | rotation | gf2` (BASELINE cfg3 / cfg2 / cfg4 at full size): ONE JSON line
    with the contract's keys, an own roofline object measured with HIP events on the workload's dominant kernel, and the result
    of the step checked where the line carries it (terms out, generators found)."""
    import json, sys, io, contextlib, importlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    bench = importlib.import_module('bench')
    argv, buf = sys.argv, io.StringIO()
    sys.argv = ['bench.py', '--gpus', '1', '--steps', '1', '--warmup', '1', '--workload', workload, '--no-extras', '--no-cpu']
    try:
        with contextlib.redirect_stdout(buf):                      # in-process: no exec from a process that holds the GPU
            bench.main()
    finally:
        sys.argv = argv
    lines = [l for l in buf.getvalue().splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data',
                'config', 'roofline'):
        assert key in d, key
    r = d['roofline']
    assert d['metric'] == metric and d['value'] > 0 and d['n_gpus'] == 1 and d['vs_baseline'] is None and 'workload' in d['config']
    assert r['bound'] == 'hbm' and r['kernel'] == kernel and r['launches'] > 0 and 0 < r['frac'] < 1.2 and r['peak'] == 8000.0
    assert abs(r['achieved'] / r['peak'] - r['frac']) < 1e-9 and 'traffic' in r
    if workload == 'mul_cleanup':
        assert d['config']['pairs_per_step'] == 10 ** 8 and 2.4e7 < d['config']['terms_out'] < 2.6e7
    if workload == 'rotation':
        assert d['config']['terms'] == 100000 and 1.4e5 < d['config']['terms_out'] <= 1.5e5 and r['launches'] == 100
    if workload == 'gf2':
        assert d['config']['generators_found'] == 32 and d['config']['matrix'] == [4000, 54000] and 7e6 < d['config']['row_xors_per_step'] < 9e6


def test_f2_generator_reconstruction_full_size_against_the_c_oracle():
    """SURVEY 8f row f2 at cfg4's operator size: 50,000 terms on 2,000 qubits reconstructed in 60 independent generators — the device builds
    the transposed [4000 x 50,060] stack bit-packed, reduces it and reads (R, mask) out in pivot order (csrc/genrec.hip).  Expected values: the
    reference's lines (base.py:552-560, utils.py:317-359) on the C oracle's _rref_binary of the same packed matrix.  Neither operand is ever
    expanded to one byte per bit on the product's side.  Then PauliwordOp.generators of a 20,000-term operator against the C oracle's rref."""
    rng = np.random.default_rng(4004)
    n, T, g = 2000, 50000, 60
    gens = rng.random((g, 2 * n)) < 0.3
    combos = rng.random((T, g)) < 0.2
    inside = (combos.astype(np.float32) @ gens.astype(np.float32)) % 2 == 1
    symp = np.where((np.arange(T) % 2 == 0)[:, None], inside, rng.random((T, 2 * n)) < 0.3)          # even terms in the span, odd terms outside
    del inside
    G = PauliwordOp._from_packed(packing.pack_rows(gens), n, np.ones(g))
    M = PauliwordOp._from_packed(packing.pack_rows(symp), n, np.ones(T))
    R, mask = M.generator_reconstruction(G)
    assert G._symp is None and M._symp is None, 'an operand was expanded to one byte per bit'
    assert R.shape == (T, g) and R.dtype == np.int64 and mask.shape == (T,)
    # the reference's lines, with the C oracle's _rref_binary
    stack_t = packing.pack_bits(np.vstack([gens, symp]).T)                                           # [4000, ceil(50060 / 64)]
    red, _, piv = oc.rref(stack_t, want_pivots=True)
    has = np.flatnonzero(piv >= 0)
    order = np.concatenate([has[np.argsort(piv[has], kind='stable')], np.flatnonzero(piv < 0)])
    reduced = packing.unpack_bits(red[order], g + T).T                                               # cref_binary(vstack([G, M]))
    assert np.array_equal(R, reduced[g:, :g].astype(int))
    assert np.array_equal(mask, np.all(~reduced[g:, g:], axis=1))
    assert np.array_equal(mask, np.arange(T) % 2 == 0), 'exactly the terms built from the generators are reconstructed'
    assert np.array_equal(R[mask], combos[mask].astype(int)), 'independent generators: the reconstruction is the combination the term was built from'
    del reduced, red, stack_t
    # generators of a tall operator: 20,000 terms, rank 2n = 4,000
    sub = M[:20000]
    gen_op = sub.generators
    ered, _, epiv = oc.rref(packing.pack_rows(symp[:20000]), want_pivots=True)
    assert gen_op.n_terms == int((epiv >= 0).sum()) == 4000
    assert np.array_equal(gen_op.packed, ered[epiv >= 0])
