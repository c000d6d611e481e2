"""Device residency behind the drop-in classes ("device-resident operands ... so multi-step callers avoid PCIe").

Results of the kernels stay on the GPU (`PauliwordOp._dev`) and reach the host when `symp_matrix` / `packed` / `coeff_vec` are read;
operands are uploaded once.  The library counts the payload bytes it copies in either direction (`symgpu_debug_counter` 7-10), and
these tests assert through those counters that the reference's multi-step callers — `(P * P) * Q`, and rotate -> project -> cleanup
of `symmer/projection/base.py:44-124` — move their operator once in and once out, with results identical to the oracle's."""
import numpy as np
import pytest
from symmer_amd import PauliwordOp, IndependentOp, kernels, packing
from oracle import oracle_np as onp
from oracle import oracle_c as oc
from _golden import assert_op_equal

pytestmark = pytest.mark.gpu


def dyadic(rng, t):
    return (rng.integers(-8, 9, t) + 1j * rng.integers(-8, 9, t)) / 16.0


def op_bytes(op):
    return op.n_terms * (16 * packing.words_per_block(op.n_qubits) + 16)


class Traffic:
    """Deltas of the library's transfer counters around a block."""
    def __enter__(self):
        self.t0 = kernels.transfer_counters()
        return self

    def __exit__(self, *exc):
        t1 = kernels.transfer_counters()
        self.h2d, self.d2h, self.uploads, self.downloads = (b - a for a, b in zip(self.t0, t1))
        return False


def is_device_only(op):
    return op._dev is not None and op._symp is None and op._packed_cache is None and op._coeff is None


def test_square_then_multiply_moves_every_operand_once():
    rng = np.random.default_rng(501)
    n = 100
    P = PauliwordOp(rng.random((300, 2 * n)) < 0.3, dyadic(rng, 300))
    Q = PauliwordOp(rng.random((40, 2 * n)) < 0.3, dyadic(rng, 40))
    with Traffic() as t:
        R = (P * P) * Q
    assert is_device_only(R)
    assert t.uploads == 2 and t.h2d == op_bytes(P) + op_bytes(Q), (t.uploads, t.h2d)
    assert t.downloads == 0 and t.d2h == 0, 'an intermediate came back to the host'
    with Traffic() as t:
        rows, coeff = R.packed, R.coeff_vec
    assert t.downloads == 1 and t.d2h == op_bytes(R)
    r1, c1 = oc.mul(P.packed, P.coeff_vec, P.packed, P.coeff_vec)
    er, ec = oc.mul(r1, c1, Q.packed, Q.coeff_vec)
    assert np.array_equal(rows, er) and np.array_equal(coeff, ec)
    with Traffic() as t:                                                  # the operands are resident now: nothing goes up again but P's and
        R2 = P * Q                                                        # Q's coefficients (their host arrays were handed out two lines above)
    assert t.h2d == 16 * (P.n_terms + Q.n_terms) and t.d2h == 0
    with Traffic() as t:                                                  # ownership is an explicit flag, not a reference count: arrays that have
        R3 = P * Q                                                        # been handed out once are refreshed before every call (16 B per term)
    assert t.h2d == 16 * (P.n_terms + Q.n_terms) and t.d2h == 0
    P2 = PauliwordOp(P.symp_matrix, dyadic(rng, 300))                     # an operator whose coefficients nobody has seen: rows + coefficients go up once
    with Traffic() as t:
        P2 * Q; P2 * Q
    assert t.uploads == 3 and t.h2d == op_bytes(P2) + 2 * 16 * Q.n_terms, (t.uploads, t.h2d)     # P2 once + two refreshes of Q's coefficients
    er, ec = oc.mul(P.packed, P.coeff_vec, Q.packed, Q.coeff_vec)
    for R in (R2, R3):
        assert np.array_equal(R.packed, er) and np.array_equal(R.coeff_vec, ec)
    held = P.coeff_vec                                                    # a caller keeps the array ...
    P * Q
    held[0] = 3.0                                                         # ... and writes through it later: seen by the next call
    R4 = P * Q
    er, ec = oc.mul(P.packed, held, Q.packed, Q.coeff_vec)
    assert np.array_equal(R4.packed, er) and np.array_equal(R4.coeff_vec, ec)


def planted_symmetry_operator(rng, n, T, k):
    """T random terms on n qubits whose X bits vanish on qubits 0..k-1 (=> Z_0..Z_{k-1} commute with every term), scrambled by
    Clifford rotations as symmer/utils.py:141-149 does."""
    symp = rng.random((T, 2 * n)) < 0.3
    symp[:, :k] = False
    H = PauliwordOp(symp, rng.standard_normal(T) + 0j).cleanup()
    rots = [(PauliwordOp((rng.random((1, 2 * n)) < 0.1), [1]), np.pi / 2) for _ in range(6)]
    return H.perform_rotations(rots)


def test_tapering_workflow_moves_the_operator_once():
    """QubitTapering.taper_it (symmer/projection/qubit_tapering.py:54-106): generators from the resident operator, the rotation chain
    on the resident operator, projection + cleanup on the resident operator; only the tapered operator comes back."""
    from symmer_amd.projection import QubitTapering
    rng = np.random.default_rng(502)
    n, T, k = 200, 20000, 5
    scr = planted_symmetry_operator(rng, n, T, k)
    H = PauliwordOp(scr.symp_matrix.copy(), scr.coeff_vec.copy())         # a user's operator: host arrays
    # an operand of this size goes up in the reference layout (one byte per bit) and is packed by a device kernel: 8x the packed bytes
    # over PCIe, 10x faster than np.packbits on the host
    assert H.symp_matrix.size >= PauliwordOp._UPLOAD_BOOL_MIN_BYTES
    rows_bytes, coeff_bytes = H.symp_matrix.size, 16 * H.n_terms
    with Traffic() as t:
        tap = QubitTapering(H)
        assert tap.n_taper == k
        out = tap.taper_it(sector=[1, -1, 1, 1, -1])
    small = 1 << 16                                                       # the five stabilisers, their rotations, index lists
    assert t.h2d < rows_bytes + 2 * coeff_bytes + small, f'{t.h2d} bytes went up for an operator of {rows_bytes + coeff_bytes}'
    assert t.d2h < small, f'{t.d2h} bytes came back before anybody read the result'
    assert is_device_only(out) and out.n_qubits == n - k
    with Traffic() as t:
        got_rows, got_coeff = out.packed, out.coeff_vec
    assert t.downloads == 1 and t.d2h == op_bytes(out)
    # the same workflow with every intermediate taken through the host (fresh objects from host arrays at every step)
    gens = IndependentOp.symmetry_generators(PauliwordOp(H.symp_matrix, H.coeff_vec))
    gens.coeff_vec = np.array([1, -1, 1, 1, -1])
    rot_stab = gens.rotate_onto_single_qubit_paulis()
    rotated = PauliwordOp(H.symp_matrix, H.coeff_vec).perform_rotations(gens.stabilizer_rotations)
    rs, rc = rotated.symp_matrix, rotated.coeff_vec
    stab = rot_stab.symp_matrix
    commutes = onp.commutes_termwise(rs, stab)
    keep_rows = np.all(commutes, axis=1)
    cols = np.nonzero(stab)[1]
    ev = rs[keep_rows][:, cols] * np.asarray(rot_stab.coeff_vec)
    ev[ev == 0] = 1
    w = rc[keep_rows] * np.prod(ev, axis=1)
    free = np.setdiff1d(np.arange(n), cols % n)
    er, ec = onp.cleanup_op(rs[keep_rows][:, np.hstack([free, free + n])], w)
    assert_op_equal(packing.unpack_rows(got_rows, n - k), got_coeff, er, ec, exact=False, tol=1e-12)


@pytest.mark.parametrize('n', [1, 3, 63, 64, 65, 100, 130, 1000])
def test_reference_layout_is_packed_and_unpacked_on_the_device(n):
    rng = np.random.default_rng(503 + n)
    T = 257 if n < 1000 else 700                                           # 700 x 2000 bytes: above the 1 MiB gate of PauliwordOp._device
    symp = rng.random((T, 2 * n)) < 0.4
    coeff = dyadic(rng, T)
    dev = kernels.DeviceOp.upload_bool(symp, coeff)
    rows, c = dev.download()
    assert np.array_equal(rows, packing.pack_rows(symp)) and np.array_equal(c, coeff)
    assert np.array_equal(dev.download_bool(n), symp)
    P = PauliwordOp(symp, coeff)
    R = P.cleanup()                                                        # goes through P._device(): the ballot-kernel path for the large case
    er, ec = onp.cleanup_op(symp, coeff)
    assert is_device_only(R) and R.n_terms == er.shape[0]
    assert np.array_equal(R.symp_matrix, er) and np.array_equal(R.coeff_vec, ec)
    assert R._packed_cache is None, 'symp_matrix of a resident operator is unpacked on the device'
    assert np.array_equal(R.packed, packing.pack_rows(er))


def test_coefficients_changed_on_the_host_are_seen_by_the_next_call():
    """`op.coeff_vec[i] = x` and `op.coeff_vec *= -1` are reference idioms (base.py:742-748): the resident rows are reused, the
    coefficients are refreshed."""
    rng = np.random.default_rng(504)
    n = 70
    P = PauliwordOp(rng.random((120, 2 * n)) < 0.3, dyadic(rng, 120))
    Q = PauliwordOp(rng.random((9, 2 * n)) < 0.3, dyadic(rng, 9))

    def check(A, B):
        R = A * B
        er, ec = oc.mul(A.packed, A.coeff_vec, B.packed, B.coeff_vec)
        assert np.array_equal(R.packed, er) and np.array_equal(R.coeff_vec, ec)
        return R
    check(P, Q)
    P.coeff_vec[3] = 0.5 - 2j                                              # in place, behind the object's back
    check(P, Q)
    P.coeff_vec *= -1
    check(P, Q)
    P.coeff_vec = dyadic(rng, 120)
    R = check(P, Q)
    c = R.coeff_vec                                                        # a result's coefficients, handed out ...
    c *= 2                                                                 # ... and changed by the caller
    check(R, Q)
    twin = P.copy()
    twin.coeff_vec *= -1                                                   # the copy shares P's resident rows, not its coefficients
    check(P, Q)
    S = P + twin
    assert S.n_terms == 0
    # an alias held across device calls (invisible to any reference count on CPython 3.14 / PyPy): ownership is an explicit flag
    held = P.coeff_vec
    check(P, Q)                                                            # a device call with the alias outstanding
    held *= 3                                                              # ... then a write through it
    check(P, Q)
    held[5] = -7j
    check(P, Q)
    # the constructor copies its coefficient argument: the operator does not change behind `coeff_vec`'s back
    mine = dyadic(rng, 9)
    Q2 = PauliwordOp(Q.symp_matrix, mine)
    before = (P * Q2).coeff_vec.copy()
    mine *= 5
    assert np.array_equal((P * Q2).coeff_vec, before) and not np.shares_memory(Q2._coeff, mine)


def test_operators_pickle_from_the_device():
    """the reference's objects are plain NumPy and are pickled by its process pool and by users; a device-resident
    result comes to the host when it is pickled and the copy starts without a handle.  A bare DeviceOp refuses with a clear message."""
    import pickle
    from symmer_amd import IndependentOp
    rng = np.random.default_rng(511)
    n = 70
    P = PauliwordOp(rng.random((150, 2 * n)) < 0.3, dyadic(rng, 150))
    R = P * P
    assert is_device_only(R)
    back = pickle.loads(pickle.dumps(R))
    assert back._dev is None and np.array_equal(back.packed, R.packed) and np.array_equal(back.coeff_vec, R.coeff_vec)
    assert back == R and np.array_equal((back * P).packed, (R * P).packed)
    host_only = PauliwordOp(rng.random((7, 2 * n)) < 0.3, dyadic(rng, 7))
    again = pickle.loads(pickle.dumps(host_only))
    assert np.array_equal(again.symp_matrix, host_only.symp_matrix) and np.array_equal(again.coeff_vec, host_only.coeff_vec)
    G = IndependentOp.from_list(['ZI', 'IZ'])
    G2 = pickle.loads(pickle.dumps(G))
    assert type(G2) is IndependentOp and np.array_equal(G2.symp_matrix, G.symp_matrix) and list(G2.coeff_vec) == list(G.coeff_vec)
    with pytest.raises(TypeError, match='cannot be pickled'):
        pickle.dumps(R._dev)


def test_scaling_dagger_indexing_and_sums_stay_on_the_device():
    rng = np.random.default_rng(505)
    n = 100
    P = PauliwordOp(rng.random((200, 2 * n)) < 0.3, dyadic(rng, 200))
    R = P * P
    rows, coeff = (a.copy() for a in R._dev.download())
    with Traffic() as t:
        scaled, dag = R * (0.5 - 0.25j), R.dagger
        part, one, last, some = R[3:40:2], R[7], R[-1], R[[5, 1, 5]]
        mask = np.zeros(R.n_terms, dtype=bool); mask[::3] = True
        masked = R[mask]
        total = R + scaled
        diff = R - R
    assert t.d2h == 0 and t.downloads == 0, 'coefficient-only and index-only operations fetched the operator'
    assert all(is_device_only(x) for x in (scaled, dag, part, one, last, some, masked, total))
    assert np.array_equal(scaled.packed, rows) and np.array_equal(scaled.coeff_vec, coeff * (0.5 - 0.25j))
    assert np.array_equal(dag.packed, rows) and np.array_equal(dag.coeff_vec, coeff.conjugate())
    for got, idx in ((part, slice(3, 40, 2)), (one, [7]), (last, [-1]), (some, [5, 1, 5]), (masked, mask)):
        assert np.array_equal(got.packed, rows[idx]) and np.array_equal(got.coeff_vec, coeff[idx])
    er, ec = oc.cleanup(np.vstack([rows, rows]), np.hstack([coeff, coeff * (0.5 - 0.25j)]))
    assert np.array_equal(total.packed, er) and np.array_equal(total.coeff_vec, ec)
    assert diff.n_terms == 0 and diff.symp_matrix.shape == (0, 2 * n)
    assert np.array_equal(R.Y_count, oc.ycount(rows))
    assert np.array_equal(R.commutes_termwise(P), oc.commutes(rows, P.packed).astype(bool))


def test_rotation_results_are_resident_and_chain():
    rng = np.random.default_rng(506)
    n = 130
    P = PauliwordOp(rng.random((500, 2 * n)) < 0.3, dyadic(rng, 500)).cleanup()
    qs = [rng.random(2 * n) < 0.3 for _ in range(3)]
    with Traffic() as t:
        cur = P
        for q, ang in zip(qs, (0.3, np.pi / 2, -1.1)):
            cur = cur._rotate_by_single_Pword(PauliwordOp(q.reshape(1, -1), [1]), ang)
        chained = P.perform_rotations([(PauliwordOp(q.reshape(1, -1), [1]), a) for q, a in zip(qs, (0.3, np.pi / 2, -1.1))])
    assert is_device_only(cur) and is_device_only(chained)
    assert t.d2h < 4096, 'rotations read the operator back'
    es, ec = P.symp_matrix, P.coeff_vec
    for q, ang in zip(qs, (0.3, np.pi / 2, -1.1)):
        es, ec = onp.rotate_by_single_pword(es, ec, q, ang)
    assert_op_equal(cur.symp_matrix, cur.coeff_vec, es, ec, exact=False, tol=1e-12)
    es, ec = onp.perform_rotations(P.symp_matrix, P.coeff_vec, list(zip(qs, (0.3, np.pi / 2, -1.1))))
    assert_op_equal(chained.symp_matrix, chained.coeff_vec, es, ec, exact=False, tol=1e-12)


# ---------------------------------------------------------------- single-process multi-device mode (symmer_amd/multi.py), one device ----
def test_device_group_on_one_device_runs_the_grouped_rccl_path(monkeypatch):
    """The pool has one GPU per box: DeviceGroup over HipBackend(1) with SYMGPU_FORCE_COMM=1 goes through symgpu_init_all, ncclCommInitAll
    with one communicator, the grouped all-gather (symgpu_comm_allgather_ops) and the same block kernels as n devices would; the block
    arithmetic for 2 .. 8 devices is covered on the CPU by tests/test_multi_device.py.  Host sources and resident sources."""
    from symmer_amd import multi, _lib
    monkeypatch.setenv('SYMGPU_FORCE_COMM', '1')
    be = multi.HipBackend(1)
    assert be.n == 1 and be.degraded is None, be.degraded
    grp = multi.DeviceGroup(be)
    rng = np.random.default_rng(507)
    n, N, M = 130, 700, 333
    A = packing.pack_rows(rng.random((N, 2 * n)) < 0.3); B = packing.pack_rows(rng.random((M, 2 * n)) < 0.3)
    a, b = dyadic(rng, N), dyadic(rng, M)
    assert np.array_equal(grp.commutes((A, None), (B, None)), oc.commutes(A, B))
    assert np.array_equal(grp.commutes((A, None)), oc.commutes(A, A))
    dA, dB = kernels.DeviceOp.upload(A, a), kernels.DeviceOp.upload(B, b)
    assert np.array_equal(grp.commutes(dA, dB), oc.commutes(A, B))
    for src_a, src_b in (((A, a), (B, b)), (dA, dB)):
        for left in (True, False):
            res = grp.mul_cleanup(src_a, src_b, left, 1e-15)
            pr, pc = oc.mul_allpairs(A, a, B, b, left)
            er, ec = oc.cleanup(pr, pc, 1e-15)
            rows, coeff = res.download()
            assert np.array_equal(rows, er) and np.array_equal(coeff, ec)
    res = grp.mul_cleanup(dA, None, True, 1e-15, same=True)
    er, ec = oc.mul(A, a, A, a)
    rows, coeff = res.download()
    assert np.array_equal(rows, er) and np.array_equal(coeff, ec)
    # the thread's current device survives a sharded call, and an operator created afterwards meets the older ones
    home = _lib.current_device()
    grp.commutes(dA, dB)
    assert _lib.current_device() == home
    fresh = kernels.DeviceOp.upload(B, b)
    prod = kernels.mul_cleanup_handles(dA, fresh, True, 1e-15)
    pr, pc = oc.mul_allpairs(A, a, B, b, True)
    er, ec = oc.cleanup(pr, pc, 1e-15)
    rows, coeff = prod.download()
    assert np.array_equal(rows, er) and np.array_equal(coeff, ec)
    _lib.check(_lib.load().symgpu_comm_destroy())


def test_current_device_and_handle_devices():
    """symgpu_set_device / symgpu_current_device; a device that does not exist is refused; a handle keeps working whatever the thread's
    current device is (calls run on the handle's device)."""
    import threading
    from symmer_amd import _lib
    _lib.set_device(0)
    assert _lib.current_device() == 0
    with pytest.raises(_lib.SymgpuError):
        _lib.set_device(_lib.device_count())                            # one past the last device
    assert _lib.current_device() == 0
    rng = np.random.default_rng(508)
    rows = packing.pack_rows(rng.random((50, 140)) < 0.3)
    op = kernels.DeviceOp.upload(rows, dyadic(rng, 50))
    seen = {}

    def other_thread():                                                 # a fresh thread starts on the process' default device
        seen['dev'] = _lib.current_device()
        seen['rows'] = op.download(with_coeff=False)
    th = threading.Thread(target=other_thread); th.start(); th.join()
    assert seen['dev'] == 0 and np.array_equal(seen['rows'], rows)
