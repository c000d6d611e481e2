"""Single-process multi-device mode (symmer_amd/multi.py): the sharding arithmetic of ``DeviceGroup`` — block bounds, padded 1/G shards of
the gathered operand, placement of gathered rows and of result blocks, device-major merge of the product's parts — driven for 1 .. 8
"devices" on the CPU: a backend that keeps every "device operator" as a NumPy array and answers the kernel calls with the C oracle.  The
same DeviceGroup code runs on real devices through HipBackend (tests/test_gpu_resident.py covers it with one device)."""
import numpy as np
import pytest
from symmer_amd.multi import DeviceGroup
from oracle import oracle_c as oc
from oracle import oracle_np as onp


class FakeOp:
    def __init__(self, device, capacity, W, with_coeff):
        self.device, self.rows, self.T = device, np.zeros((capacity, W), dtype='<u8'), 0
        self.coeff = np.zeros(capacity, dtype=np.complex128) if with_coeff else None
        self.freed = False


class OracleBackend:
    """G pretend devices; every cross-device rule the real backend enforces is asserted here (an operator is only ever used on its device)."""
    degraded = None

    def __init__(self, n, home=0):
        self.n = n
        self.live = 0
        self.gathers = 0
        self.cur = home                                                # the calling thread's current device

    def use(self, d):
        self.cur = d

    def current(self):
        return self.cur

    def _src(self, source):
        return (source.rows[:source.T], None if source.coeff is None else source.coeff[:source.T]) if isinstance(source, FakeOp) else source

    def n_source_rows(self, source):
        return source.T if isinstance(source, FakeOp) else source[0].shape[0]

    def width(self, source):
        return (source.rows if isinstance(source, FakeOp) else source[0]).shape[1] // 2

    def sync(self, d):
        pass

    def shard_from(self, d, source, r0, r1, capacity, with_coeff):
        self.use(d)                                                    # (as HipBackend: every call moves the thread to the device it works on)
        rows, coeff = self._src(source)
        op = FakeOp(d, max(1, capacity), rows.shape[1], with_coeff)
        op.rows[:r1 - r0] = rows[r0:r1]
        if with_coeff:
            op.coeff[:r1 - r0] = coeff[r0:r1]
        op.T = r1 - r0
        self.live += 1
        return op

    def alloc(self, d, capacity, wq, with_coeff):
        self.use(d)
        self.live += 1
        return FakeOp(d, max(1, capacity), 2 * wq, with_coeff)

    def allgather(self, shards, fulls, n_rows_total):
        self.gathers += 1
        ts = shards[0].rows.shape[0]
        assert all(s.rows.shape[0] == ts for s in shards), 'shards of one gather have one capacity'
        for d, f in enumerate(fulls):
            assert f.device == d and shards[d].device == d and f.rows.shape[0] >= ts * self.n
            for s_idx, s in enumerate(shards):
                f.rows[s_idx * ts: s_idx * ts + ts] = 0
                f.rows[s_idx * ts: s_idx * ts + s.T] = s.rows[:s.T]
                if f.coeff is not None:
                    f.coeff[s_idx * ts: s_idx * ts + ts] = 0
                    f.coeff[s_idx * ts: s_idx * ts + s.T] = s.coeff[:s.T]
            f.T = n_rows_total

    def commutes_block(self, d, a, a0, a1, b):
        self.use(d)
        assert a.device == d and b.device == d
        return oc.commutes(a.rows[a0:a1], b.rows[:b.T]).astype(np.uint8) if a1 > a0 and b.T else np.zeros((a1 - a0, b.T), dtype=np.uint8)

    def fetch_blocks(self, bufs, out, bounds):
        for d, (b0, b1) in enumerate(bounds):
            out[b0:b1] = bufs[d]

    def mul_cleanup(self, d, inner, outer, inner_is_left, zero_threshold):
        self.use(d)
        assert inner.device == d and outer.device == d
        rows, coeff = oc.mul_allpairs(inner.rows[:inner.T], inner.coeff[:inner.T], outer.rows[:outer.T], outer.coeff[:outer.T], inner_is_left)
        r, c = oc.cleanup(rows, coeff, zero_threshold)
        op = FakeOp(d, max(1, r.shape[0]), r.shape[1], True)
        op.rows[:r.shape[0]] = r; op.coeff[:r.shape[0]] = c; op.T = r.shape[0]
        self.live += 1
        return op

    def concat_on(self, d, parts):
        self.use(d)
        total = sum(p.T for p in parts)
        op = FakeOp(d, max(1, total), parts[0].rows.shape[1], True)
        at = 0
        for p in parts:
            op.rows[at:at + p.T] = p.rows[:p.T]; op.coeff[at:at + p.T] = p.coeff[:p.T]
            at += p.T
        op.T = total
        self.live += 1
        return op

    def cleanup(self, d, op, zero_threshold):
        self.use(d)
        assert op.device == d
        r, c = oc.cleanup(op.rows[:op.T], op.coeff[:op.T], zero_threshold)
        out = FakeOp(d, max(1, r.shape[0]), op.rows.shape[1], True)
        out.rows[:r.shape[0]] = r; out.coeff[:r.shape[0]] = c; out.T = r.shape[0]
        self.live += 1
        return out

    def free(self, op):
        assert not op.freed, 'an operator was freed twice'
        op.freed = True
        self.live -= 1


def dyadic(rng, t):
    return (rng.integers(-8, 9, t) + 1j * rng.integers(-8, 9, t)) / 16.0


@pytest.mark.parametrize('G', [1, 2, 3, 8])
@pytest.mark.parametrize('n,N,M', [(100, 257, 64), (70, 5, 300), (130, 8, 8), (3, 50, 7), (64, 1000, 999)])
def test_commutation_blocks_over_devices(G, n, N, M):
    rng = np.random.default_rng(n * 7 + N + G)
    a = onp.pack_rows(rng.random((N, 2 * n)) < 0.3); b = onp.pack_rows(rng.random((M, 2 * n)) < 0.3)
    be = OracleBackend(G)
    grp = DeviceGroup(be)
    got = grp.commutes((a, None), (b, None))
    assert got.dtype == np.bool_ and np.array_equal(got, oc.commutes(a, b))
    assert be.live == 0 and be.gathers == 1, 'every device operator is released; one all-gather per call'
    got = grp.commutes((a, None))                                      # self-adjacency: the gather is also the distribution of the left blocks
    assert np.array_equal(got, oc.commutes(a, a)) and be.live == 0 and be.gathers == 2


@pytest.mark.parametrize('G', [1, 2, 3, 8])
@pytest.mark.parametrize('left', [True, False])
def test_product_cleanup_outer_blocks_over_devices(G, left):
    """Device-major order of the outer blocks is the reference's pair order (o * Ni + i, base.py:783-792): first-occurrence order and
    (dyadic) sums of the merged parts equal the single-device result — duplicates across the blocks included."""
    rng = np.random.default_rng(40 + G)
    n, Ni, No = 100, 120, 45
    A = onp.pack_rows(rng.random((Ni, 2 * n)) < 0.3); B = onp.pack_rows(rng.random((No, 2 * n)) < 0.3)
    A[Ni // 2: Ni // 2 + 10] = A[:10]; B[No // 2:] = B[: No - No // 2]            # repeated rows: terms that merge across devices
    a, b = dyadic(rng, Ni), dyadic(rng, No)
    be = OracleBackend(G)
    res = DeviceGroup(be).mul_cleanup((A, a), (B, b), left, 1e-15)
    pr, pc = oc.mul_allpairs(A, a, B, b, left)
    er, ec = oc.cleanup(pr, pc, 1e-15)
    assert res.device == 0 and np.array_equal(res.rows[:res.T], er) and np.array_equal(res.coeff[:res.T], ec)
    be.free(res)
    assert be.live == 0
    res = DeviceGroup(be).mul_cleanup((A, a), None, True, 1e-15, same=True)          # P * P: the blocks are slices of the gathered operand
    er, ec = oc.mul(A, a, A, a)
    assert np.array_equal(res.rows[:res.T], er) and np.array_equal(res.coeff[:res.T], ec)


def test_block_bounds_of_the_north_star_configuration():
    """cfg5 on 8 devices: 200,000 rows -> 25,000 per device, shards of the right operand 25,000 rows each (12.8 MB at 2,000 qubits)."""
    from symmer_amd.parallel import shard_bounds
    ts, bounds = shard_bounds(200000, 8)
    assert ts == 25000 and bounds[0] == (0, 25000) and bounds[7] == (175000, 200000)
    ts, bounds = shard_bounds(10, 8)                                   # fewer rows than devices: tail devices are empty
    assert ts == 2 and bounds[5] == (10, 10) and bounds[4] == (8, 10)


@pytest.mark.parametrize('home', [0, 2])
def test_sharded_calls_leave_the_current_device_alone(home):
    """a sharded call visits every device; the thread's current device — where the caller's next operator will be
    created — must be what it was, and the product lands on the caller's device, not on device 0."""
    rng = np.random.default_rng(5)
    n, N = 64, 90
    a = onp.pack_rows(rng.random((N, 2 * n)) < 0.3)
    c = dyadic(rng, N)
    be = OracleBackend(3, home=home)
    grp = DeviceGroup(be)
    grp.commutes((a, None))
    assert be.current() == home
    grp.commutes((a, None), (a[:7], None))
    assert be.current() == home
    res = grp.mul_cleanup((a, c), None, True, 1e-15, same=True)
    assert be.current() == home and res.device == home, 'the result lives where the caller works'
    be.free(res)
    assert be.live == 0


def test_launcher_ranks_do_not_build_a_device_group():
    from symmer_amd import multi
    assert multi.under_launcher({'WORLD_SIZE': '8', 'RANK': '3'}) and multi.under_launcher({'RANK': '0'})
    assert not multi.under_launcher({}) and not multi.under_launcher({'WORLD_SIZE': '1'})


def test_product_goes_to_the_device_group_only_on_request_or_for_size():
    """`A * B` stays on one device unless SYMGPU_DEVICES_PRODUCT=1 or one device cannot hold the product (rows +
    coefficients + 16 B of keys per pair against half of the device's memory); commutes_termwise keeps its own gate."""
    from symmer_amd import multi
    hbm = 288 << 30
    row = 256 + 16                                                     # 1,000 qubits
    assert not multi.product_uses_devices(10 ** 8, row, hbm, {})       # cfg3: 29 GB
    assert not multi.product_uses_devices(4 * 10 ** 8, row, hbm, {})   # 2 x 10^4 terms squared: 115 GB < 144 GB
    assert multi.product_uses_devices(6 * 10 ** 8, row, hbm, {})       # does not fit one MI355X
    assert multi.product_uses_devices(4 * 10 ** 8, row, hbm, {'SYMGPU_DEVICES_PRODUCT': '1'})
    assert not multi.product_uses_devices(1 << 20, row, hbm, {'SYMGPU_DEVICES_PRODUCT': '1'})      # below MIN_PAIRS_PRODUCT: never
