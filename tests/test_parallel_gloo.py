"""CPU, world_size 2 over gloo: the left-axis sharding + right-operand all-gather logic of symmer_amd/parallel.py.
The per-rank kernel is injected (the C oracle stands in for the HIP kernel, which needs a GPU), so this checks
exactly the multi-rank plumbing: shard bounds, padded gather, block placement."""
import os, socket, sys
import multiprocessing as mp
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        from symmer_amd import parallel
        from oracle import oracle_c as oc, oracle_np as onp
        comm = parallel.Communicator.from_env(data_plane='gloo-host', control='gloo')
        rng = np.random.default_rng(77)                       # same global operands on every rank
        n, N, M = 130, 37, 51
        A = onp.pack_rows(rng.random((N, 2 * n)) < 0.3); B = onp.pack_rows(rng.random((M, 2 * n)) < 0.3)
        a = (rng.integers(-8, 9, N) + 1j * rng.integers(-8, 9, N)) / 16.0; b = (rng.integers(-8, 9, M) + 1j * rng.integers(-8, 9, M)) / 16.0
        ts, bounds = parallel.shard_bounds(M, world)
        m0, m1 = bounds[rank]
        (r0, r1), blk = parallel.sharded_commutes(A, B[m0:m1], M, comm, kernel=oc.commutes)
        full = oc.commutes(A, B)
        ok = np.array_equal(blk, full[r0:r1]) and (r0, r1) == parallel.shard_bounds(N, world)[1][rank]
        (r0, r1), rows, coeff = parallel.sharded_product(A, a, B[m0:m1], b[m0:m1], M, comm, kernel=oc.mul_allpairs)
        erows, ecoeff = oc.mul_allpairs(A[r0:r1], a[r0:r1], B, b, True)
        ok = ok and np.array_equal(rows, erows) and np.array_equal(coeff, ecoeff)
        t = comm.max_over_ranks(float(rank + 1))
        ok = ok and t == float(world)
        comm.close()
        q.put((rank, bool(ok), ''))
    except Exception as e:                                    # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc()))


def test_shard_bounds():
    from symmer_amd.parallel import shard_bounds
    assert shard_bounds(10, 4) == (3, [(0, 3), (3, 6), (6, 9), (9, 10)])
    assert shard_bounds(2, 4) == (1, [(0, 1), (1, 2), (2, 2), (2, 2)])
    assert shard_bounds(200000, 8)[0] == 25000
    for n in (0, 1, 7, 64, 1001):
        for w in (1, 2, 3, 8):
            ts, b = shard_bounds(n, w)
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert all(e - s <= ts for s, e in b)


@pytest.mark.timeout(300)
def test_world_size_2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, err in res:
        assert ok, f'rank {rank} failed: {err}'


def _tcp_worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        from symmer_amd import parallel
        comm = parallel.Communicator.from_env(data_plane='none', control='tcp')     # bench.py's control plane, no torch
        payload = comm._bcast_bytes(bytes(range(128)), 128)
        ok = payload == bytes(range(128)) and comm.max_over_ranks(10.0 * (rank + 1)) == 10.0 * world
        comm.barrier()
        comm.close()
        q.put((rank, bool(ok) and 'torch' not in sys.modules, ''))
    except Exception:                                         # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc()))


@pytest.mark.timeout(120)
def test_world_size_3_tcp_control_plane():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tcp_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    for rank, ok, err in res:
        assert ok, f'rank {rank} failed: {err}'


@pytest.mark.timeout(120)
def test_tcp_control_plane_survives_an_occupied_port():
    """The first candidate port is held by an unrelated (silent) listener: rank 0 must move on, the peers must find it."""
    import socket
    port = _free_port()
    first = 1024 + (port + 7919) % 60000                      # Communicator.from_env's derived control port
    blocker = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    try:
        blocker.bind(('127.0.0.1', first))
    except OSError:
        pytest.skip('derived port already in use by something else')
    blocker.listen(4)
    try:
        ctx = mp.get_context('spawn')
        q = ctx.Queue()
        procs = [ctx.Process(target=_tcp_worker, args=(r, 2, port, q)) for r in range(2)]
        for p in procs:
            p.start()
        res = [q.get(timeout=100) for _ in procs]
        for p in procs:
            p.join(timeout=30)
        for rank, ok, err in res:
            assert ok, f'rank {rank} failed: {err}'
    finally:
        blocker.close()


def _rccl_fallback_worker(rank, world, port, q, disable_on):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        if rank in disable_on:
            os.environ['SYMGPU_RCCL_DISABLE'] = '1'             # this rank cannot load librccl
        from symmer_amd import parallel, _lib
        calls = []
        real = _lib.load().symgpu_comm_init
        # no GPU here: (a) a rank that cannot load librccl reports it BEFORE anybody enters the collective ncclCommInitRank,
        # (b) otherwise rank 0 cannot create a unique id; either way every rank must agree on the host-staged data plane
        # instead of hanging, and nobody may have called symgpu_comm_init
        _lib._lib.symgpu_comm_init = lambda *a: calls.append(a) or real(*a)
        comm = parallel.Communicator.from_env(data_plane='rccl', control='tcp')
        ok = comm.gathers and bool(comm.rccl_error) and comm.data_plane == 'host-staged' and bool(comm.degraded) and not calls
        # the fallback's transport: equal-sized byte strings, concatenated in rank order on every rank
        got = comm._allgather_bytes(bytes([rank + 1]) * 5000)
        ok = ok and got == b''.join(bytes([r + 1]) * 5000 for r in range(world))
        comm.barrier()
        comm.close()
        q.put((rank, ok, '' if ok else f'gathers={comm.gathers} plane={comm.data_plane} error={comm.rccl_error} init_calls={len(calls)}'))
    except Exception:                                         # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc()))


@pytest.mark.timeout(120)
@pytest.mark.parametrize('disable_on', [(), (1,), (0,)])
def test_rccl_failure_is_agreed_across_ranks(disable_on):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_fallback_worker, args=(r, 2, port, q, disable_on)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    for rank, ok, err in res:
        assert ok, f'rank {rank} failed: {err}'


def _first_gather_worker(rank, world, port, q, mode):
    """No GPU here: the collective itself is a stand-in callable; what is under test is the watchdog + agreement around the first
    all-gather (symmer_amd/parallel.py::Communicator._first_gather): 'error' = it raises on rank 1, 'hang' = it never returns on
    rank 1, 'ok' = it returns everywhere."""
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                          SYMGPU_RCCL_GATHER_TIMEOUT='2')
        import time
        from symmer_amd import parallel
        comm = parallel.Communicator.from_env(data_plane='none', control='tcp')
        comm.data_plane, comm.gathers = 'rccl', True             # as after a successful bring-up

        def gather():
            if rank == 1 and mode == 'error':
                raise RuntimeError('RCCL error 5 (unhandled system error) in ncclAllGather(rows)')
            if rank == 1 and mode == 'hang':
                time.sleep(3600)

        outcome = 'returned'
        try:
            comm._first_gather(gather)
        except parallel.CollectiveHang as exc:
            outcome = 'hang:' + str(exc)
        if mode == 'ok':
            ok = outcome == 'returned' and comm.data_plane == 'rccl' and comm.degraded is None and comm._gather_checked
        elif mode == 'error':
            ok = outcome == 'returned' and comm.data_plane == 'host-staged' and bool(comm.degraded) and 'all-gather failed' in comm.degraded
            # the fallback's transport still works on both ranks
            got = comm._allgather_bytes(bytes([rank + 7]) * 100)
            ok = ok and got == b''.join(bytes([r + 7]) * 100 for r in range(world))
        else:
            ok = outcome.startswith('hang:')                       # on BOTH ranks, although the gather returned on rank 0
            ok = ok and ((rank == 1) == (comm._hung_thread is not None))
        if mode != 'hang':
            comm.barrier()
        comm.data_plane = 'none' if mode == 'ok' else comm.data_plane      # nothing to destroy: there never was a communicator
        comm.close()
        ok = ok and comm.needs_hard_exit == (mode == 'hang' and rank == 1)
        q.put((rank, ok, '' if ok else f'outcome={outcome} plane={comm.data_plane} degraded={comm.degraded}'))
        if comm.needs_hard_exit:
            q.close(); q.join_thread()
            comm.hard_exit_if_hung()                               # default status: non-zero (a launcher reading exit codes sees the degraded run)
    except Exception:                                         # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc()))


@pytest.mark.timeout(120)
@pytest.mark.parametrize('mode', ['ok', 'error', 'hang'])
def test_first_gather_is_watched_and_agreed(mode):
    """the first ncclAllGather can hang or fail on one rank only; no rank may be left waiting."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_first_gather_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(timeout=30)
        assert not p.is_alive(), 'a rank did not leave'
    for rank, ok, err in res:
        assert ok, f'rank {rank} failed ({mode}): {err}'
    if mode == 'hang':
        from symmer_amd import parallel
        assert procs[1].exitcode == parallel.Communicator.HUNG_EXIT_STATUS != 0, 'a rank with a thread stuck inside RCCL must not exit with status 0'
        assert procs[0].exitcode == 0


def _mul_cleanup_worker(rank, world, port, q):
    """SURVEY 8e stretch: product + cleanup with the outer operand sharded — rows, ROW ORDER and coefficients
    must equal the single-process oracle's (dyadic coefficients: bit for bit).  The per-rank kernels are the C oracle's."""
    try:
        sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        from symmer_amd import parallel
        from oracle import oracle_c as oc, oracle_np as onp
        comm = parallel.Communicator.from_env(data_plane='gloo-host', control='gloo')
        ok = True
        for case, (n, N, M) in enumerate([(3, 60, 41), (70, 50, 33), (5, 7, 1), (130, 40, 40)]):
            rng = np.random.default_rng(900 + case)                   # same global operands on every rank
            A = onp.pack_rows(rng.random((N, 2 * n)) < 0.4); B = onp.pack_rows(rng.random((M, 2 * n)) < 0.4)
            if case == 0:
                A[10:20] = A[:10]; B[5:9] = B[:4]                     # duplicate rows inside and across the ranks' blocks
            a = (rng.integers(-8, 9, N) + 1j * rng.integers(-8, 9, N)) / 16.0; b = (rng.integers(-8, 9, M) + 1j * rng.integers(-8, 9, M)) / 16.0
            # reference order: the operand with fewer terms is the outer one (base.py:847-852); N >= M here, so B is outer, A inner and left
            _, bounds = parallel.shard_bounds(M, world)
            m0, m1 = bounds[rank]

            def mul_kernel(inner, ci, outer, co, left, thr):
                r, c = oc.mul_allpairs(inner, ci, outer, co, left) if outer.shape[0] else (np.zeros((0, inner.shape[1]), dtype='<u8'), np.zeros(0, dtype=complex))
                return oc.cleanup(r, c, thr) if r.shape[0] else (r, c)
            rows, coeff = parallel.sharded_mul_cleanup(A, a, B[m0:m1], b[m0:m1], comm, True, 1e-15, mul_kernel=mul_kernel,
                                                       cleanup_kernel=lambda r, c, thr: oc.cleanup(r, c, thr))
            er, ec = oc.mul(A, a, B, b)
            ok = ok and np.array_equal(rows, er) and np.array_equal(coeff, ec)
            # ---- the same product with the PAIRS partitioned by the linear class of their product row: both operands
            #      complete on every rank, nothing exchanged but each rank's share of the result.  Checker kernels on the oracle:
            #      first-occurrence unique with indices (utils.py:271) + np.add.at sums (:273-274)
            def indexed_cleanup(r, c, thr):
                first, inv = onp.first_occurrence_unique(np.ascontiguousarray(r).view(np.uint8).reshape(r.shape[0], -1))
                sums = np.zeros(first.shape[0], dtype=complex)
                np.add.at(sums, inv, c)
                keep = np.ones(first.shape[0], dtype=bool) if thr is None else np.abs(sums) > thr
                return r[first][keep], sums[keep], first[keep]

            def indexed_mul(inner, ci, outer, co, left):
                r, c = oc.mul_allpairs(inner, ci, outer, co, left)
                rr, cc, first = indexed_cleanup(r, c, None)
                return rr, cc, first % inner.shape[0], first // inner.shape[0]
            for (X, x, Y, y, left) in ((A, a, B, b, True), (A, a, A, a, True), (B, b, A, a, False)):
                st = {}
                rows, coeff = parallel.hash_partitioned_mul_cleanup(X, x, Y, y, comm, left, 1e-15, mul_kernel=indexed_mul, cleanup_kernel=indexed_cleanup, stats=st)
                rr, cc = oc.mul_allpairs(X, x, Y, y, left)
                er, ec = oc.cleanup(rr, cc, 1e-15)
                ok = ok and np.array_equal(rows, er) and np.array_equal(coeff, ec)
                # every pair has exactly one owner, and what a rank sends is its share of the RESULT — no key, no partial product row
                owned = np.frombuffer(comm._allgather_bytes(np.int64(st['pairs_owned']).tobytes()), dtype=np.int64)
                ok = ok and int(owned.sum()) == st['pairs_total'] == X.shape[0] * Y.shape[0]
                sent = np.frombuffer(comm._allgather_bytes(np.int64(st['bytes_sent']).tobytes()), dtype=np.int64)
                ok = ok and int(sent.sum()) == er.shape[0] * (er.shape[1] * 8 + 16 + 8)
                # (no key and no partial product is exchanged at all: the bytes a rank sends are exactly its kept terms — row, coefficient,
                # first pair index — i.e. the result is the only traffic, 1/G of it per rank on average)
        comm.close()
        q.put((rank, bool(ok), ''))
    except Exception:                                         # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc()))


@pytest.mark.timeout(300)
@pytest.mark.parametrize('world', [2, 3])
def test_sharded_mul_cleanup_equals_the_oracle(world):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mul_cleanup_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, err in res:
        assert ok, f'rank {rank} failed: {err}'
