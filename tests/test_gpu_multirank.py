"""One process per GPU, world_size 2, on real hardware (`-m gpu`).

* ``rccl``: needs two GPUs — RCCL all-gather over xGMI through ``Communicator.from_env`` + ``allgather_op`` (skipped on a
  one-GPU box; the driver's 8-GPU scaling run exercises the same path through bench.py).
* ``host-staged``: RCCL made unavailable on purpose (``SYMGPU_RCCL_DISABLE=1``) — the ranks must agree on the fallback
  before anybody enters ``ncclCommInitRank`` and gather through host memory; with one GPU both ranks share device 0.
* bench.py's launch line with two ranks (torch.distributed.run's environment), reduced workload.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _launch(argv, world, extra_env, timeout=300):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY='0', **extra_env)
        procs.append(subprocess.Popen([sys.executable] + argv, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, e))
    return outs


def _n_gpus():
    from symmer_amd import _lib
    return _lib.device_count()


@pytest.mark.timeout(600)
@pytest.mark.parametrize('plane', ['rccl', 'host-staged'])
def test_two_ranks_gather_and_blocks(plane):
    if plane == 'rccl' and _n_gpus() < 2:
        pytest.skip('RCCL needs one GPU per rank; this box has one')
    env = {'EXPECT_PLANE': plane}
    if plane == 'host-staged':
        env['SYMGPU_RCCL_DISABLE'] = '1'
    outs = _launch(['tests/_rank_worker.py'], 2, env)
    for r, (rc, o, e) in enumerate(outs):
        assert rc == 0 and f'RANK_OK {r} {plane}' in o, f'rank {r}: rc={rc}\n{o}\n{e[-3000:]}'


@pytest.mark.timeout(900)
def test_bench_two_ranks_prints_one_json_line():
    """bench.py under the driver's multi-rank environment: one JSON line on rank 0's stdout, n_gpus = 2, aggregate value, and the
    data plane named (RCCL with two GPUs; with one GPU both ranks share it, RCCL refuses that, and the line is flagged degraded)."""
    outs = _launch(['bench.py', '--gpus', '2', '--steps', '1', '--warmup', '1', '--left-terms', '20000', '--right-terms', '20000', '--no-extras', '--no-cpu'],
                   2, {}, timeout=800)
    for r, (rc, o, e) in enumerate(outs):
        assert rc == 0, f'rank {r}: rc={rc}\n{o}\n{e[-3000:]}'
    lines = [l for l in outs[0][1].splitlines() if l.strip()]
    assert len(lines) == 1 and not outs[1][1].strip(), (outs[0][1], outs[1][1])
    doc = json.loads(lines[0])
    assert doc['n_gpus'] == 2 and doc['config']['pairs_per_step'] == 2 * 20000 * 20000 and doc['value'] > 0
    if _n_gpus() >= 2:
        assert 'degraded' not in doc and 'rccl' in doc['config']['parallelism']
    else:
        assert 'host-staged' in doc['degraded'] and 'host-staged' in doc['config']['parallelism']


@pytest.mark.timeout(900)
def test_bench_adjacency_two_ranks():
    """BASELINE cfg5 through bench.py with two ranks (reduced size): the all-gather distributes the operator, every rank computes
    its block of the adjacency matrix; value counts all T*T pairs."""
    outs = _launch(['bench.py', '--gpus', '2', '--steps', '1', '--warmup', '1', '--workload', 'adjacency', '--adj-terms', '20000', '--adj-qubits', '500'],
                   2, {}, timeout=800)
    for r, (rc, o, e) in enumerate(outs):
        assert rc == 0, f'rank {r}: rc={rc}\n{o}\n{e[-3000:]}'
    lines = [l for l in outs[0][1].splitlines() if l.strip()]
    assert len(lines) == 1 and not outs[1][1].strip()
    doc = json.loads(lines[0])
    assert doc['n_gpus'] == 2 and doc['config']['rows_per_gpu'] == 10000 and doc['config']['pairs_per_step'] == 20000 ** 2 and doc['value'] > 0


@pytest.mark.timeout(300)
def test_rccl_self_check_detects_a_wrong_gather_and_falls_back():
    """Communicator.verify_allgather (bench.py runs it once before the timed region): RCCL with one rank (SYMGPU_FORCE_COMM=1);
    a flipped bit in the gathered operand switches every rank to the host-staged plane, flagged `degraded`, operand repaired."""
    outs = _launch(['tests/_rccl_selfcheck_worker.py'], 1, {'SYMGPU_FORCE_COMM': '1'})
    rc, o, e = outs[0]
    assert rc == 0 and 'SELFCHECK_OK' in o, f'rc={rc}\n{o}\n{e[-3000:]}'


@pytest.mark.timeout(300)
@pytest.mark.parametrize('worker', [['tests/_weak_hash_worker.py'], ['tests/_weak_hash_worker2.py', 'square'], ['tests/_weak_hash_worker2.py', 'pair'],
                                    ['tests/_weak_hash_worker2.py', 'rotate']])
def test_row_hash_collisions_are_caught_and_reseeded(worker):
    """The cleanup's exactness never rests on its 64-bit row hash: rows with equal hashes are compared word for word and a mismatch
    reseeds the hash and redoes the pass.  SYMGPU_HASH_WEAK_ODD=1 cuts the first seed's hash to 4 bits in a fresh process, so the
    guard (and the full-sort fallback for long mixed prefix runs) actually runs; results must still be the oracle's."""
    outs = _launch(worker, 1, {'SYMGPU_HASH_WEAK_ODD': '1'})
    rc, o, e = outs[0]
    assert rc == 0 and 'WEAK_HASH_OK' in o, f'rc={rc}\n{o}\n{e[-3000:]}'


@pytest.mark.timeout(900)
def test_bench_eight_ranks_rank_arithmetic():
    """bench.py at the driver's largest world size, reduced workloads (on a one-GPU box the eight ranks share the device and the
    host-staged plane runs): shard bounds with a remainder (20,003 right terms / 40,003 adjacency terms over 8 ranks), the TCP
    control plane with 8 peers, one JSON line from rank 0 only, aggregate pair counts."""
    for argv, pairs in ((['bench.py', '--gpus', '8', '--steps', '1', '--warmup', '1', '--left-terms', '6000', '--right-terms', '20003', '--no-extras', '--no-cpu'],
                         8 * 6000 * 20003),
                        (['bench.py', '--gpus', '8', '--steps', '1', '--warmup', '1', '--workload', 'adjacency', '--adj-terms', '40003', '--adj-qubits', '300'],
                         40003 ** 2)):
        outs = _launch(argv, 8, {}, timeout=800)
        for r, (rc, o, e) in enumerate(outs):
            assert rc == 0, f'rank {r}: rc={rc}\n{o}\n{e[-3000:]}'
            assert (r == 0) == bool(o.strip()), (r, o)
        doc = json.loads(outs[0][1].strip().splitlines()[-1])
        assert doc['n_gpus'] == 8 and doc['config']['pairs_per_step'] == pairs and doc['value'] > 0
        if _n_gpus() < 8:
            assert 'host-staged' in doc.get('degraded', '')


# ---------------------------------------------------------------- one process, several devices (symmer_amd/multi.py) ----
@pytest.mark.timeout(600)
def test_single_process_device_group_on_all_devices():
    """DeviceGroup over HipBackend on every visible device (symgpu_init_all, ncclCommInitAll, grouped all-gather, peer copies, one download
    thread per device) against the C oracle.  Needs two devices; the pool's boxes have one (tests/test_gpu_resident.py runs the same code with
    one device, tests/test_multi_device.py the block arithmetic for 8)."""
    if _n_gpus() < 2:
        pytest.skip('one process driving several devices needs several devices; this box has one')
    import numpy as np
    from symmer_amd import multi, packing, _lib
    from oracle import oracle_c as oc
    G = min(8, _n_gpus())
    be = multi.HipBackend(G)
    grp = multi.DeviceGroup(be)
    rng = np.random.default_rng(77)
    n, N, M = 300, 5000, 3100
    A = packing.pack_rows(rng.random((N, 2 * n)) < 0.3); B = packing.pack_rows(rng.random((M, 2 * n)) < 0.3)
    a = (rng.integers(-8, 9, N) + 1j * rng.integers(-8, 9, N)) / 16.0; b = (rng.integers(-8, 9, M) + 1j * rng.integers(-8, 9, M)) / 16.0
    assert np.array_equal(grp.commutes((A, None), (B, None)), oc.commutes(A, B))
    assert np.array_equal(grp.commutes((A, None)), oc.commutes(A, A))
    res = grp.mul_cleanup((A[:900], a[:900]), (B[:400], b[:400]), True, 1e-15)
    pr, pc = oc.mul_allpairs(A[:900], a[:900], B[:400], b[:400], True)
    er, ec = oc.cleanup(pr, pc, 1e-15)
    rows, coeff = res.download()
    assert np.array_equal(rows, er) and np.array_equal(coeff, ec)
    assert be.degraded is None, be.degraded
    _lib.check(_lib.load().symgpu_comm_destroy())


@pytest.mark.timeout(900)
def test_bench_single_process_line():
    """`bench.py --gpus N` WITHOUT a launcher: one process drives the devices and prints the contract line (N = the visible devices, at most 2
    here; with one device the same path runs through ncclCommInitAll with one communicator)."""
    import subprocess, sys
    n = min(2, _n_gpus())
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--single-process', '--steps', '1', '--warmup', '1', '--left-terms', '20000',
                        '--right-terms', '20000', '--no-extras', '--no-cpu'], capture_output=True, text=True, timeout=800, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    doc = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][-1])
    assert doc['n_gpus'] == n and doc['config']['pairs_per_step'] == n * 20000 * 20000 and doc['value'] > 0
    assert 'ONE process' in doc['config']['parallelism'] and doc['degraded_kernels'] == []
