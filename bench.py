"""bench.py — headline benchmark of the symplectic hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W     (N>1: launched by torch.distributed.run,
one rank per GPU).  Prints ONE JSON line on rank 0.

Workload (config.workload = "allpairs_product"): the north-star all-pairs symplectic product of
BASELINE.json — a 1,000-qubit operator, 10^5 left terms per GPU x 10^5 right terms, product rows + phase-
corrected coefficients streamed through a ring of output slabs in HBM (10^10 pairs, 2.72 TB written per step and
GPU).  One "step" = (N>1: RCCL all-gather of the packed right operand over xGMI, each rank contributes 1/N) + the
full product of this rank's left block against the whole right operand.  Inputs are resident in HBM before the
timed region.  Weak scaling: per-GPU work is fixed, value = all ranks' pairs / max-over-ranks time.

roofline     — dominant kernel k_mul_rows_e (HBM-write stream of the product rows, which also leaves one phase byte per
               pair): algorithmic bytes 16*Wq per pair (256 B at n=1000) x pairs per launch / average launch duration
               from HIP events recorded around every launch in the timed region (library stream).  A second streaming
               kernel (k_mul_coeff_expand) turns the phase bytes into the 16 B/pair coefficients right after it; the
               whole step moves 16*Wq+16 B/pair (`whole_step_GBps`).  peak = 8 TB/s (MI355X_MICROARCH.md).
cpu_baseline — the NumPy restatement of the reference algorithm (oracle/oracle_np.py: broadcast XOR on
               1-byte-per-bit matrices, per-bit popcount sums, complex outer product; base.py:783-792) timed on a
               bounded sample of the same workload on the host cores.
extras       — the other BASELINE configs measured on the same GPU (cfg3 product+cleanup, cfg2 rotation chain,
               cfg4 GF(2) symmetry kernel in row-XORs/s, cfg5 commutation slice).  N=1, rank 0 only.

Every BASELINE config is also a workload of its own with the same contract line (own `roofline` and `cpu_baseline`):
  --workload product      north star: all-pairs product, 1e5 x 1e5 terms, 1,000 qubits (default)
  --workload mul_cleanup  cfg3: P * P of a 1e4-term, 1,000-qubit operator (1e8 pairs) + cleanup   (reference base.py:821-859)
  --workload rotation     cfg2: one non-Clifford rotation of a 1e5-term, 1,000-qubit operator      (reference base.py:1090-1161)
  --workload gf2          cfg4: symmetry generators of a 5e4-term, 2,000-qubit operator             (reference independent_op.py:90-144)
  --workload adjacency    cfg5: adjacency matrix of a 2e5-term, 2,000-qubit operator                (reference base.py:938-971)
mul_cleanup / rotation / gf2 do not shard (replicas only): with --gpus N every rank runs the same workload on its
own GPU and `value` is the sum over the ranks.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
API_LEGEND = ('api objects: first_call = fresh host arrays in (upload included); a = operands resident, result object out (left on the GPU); '
              'b = a + result host arrays out; c_abi = the device-resident C-ABI step of the same line')
PROFILE_TAG = 'r06'                     # profiles/<tag>_*: the round whose rocprofv3 summaries belong to this bench.py
TRAFFIC_PROFILE = f'{PROFILE_TAG}_traffic.json'   # written by tools/pmc_product.sh from the rocprofv3 --pmc passes of this bench


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None, help='timed steps (default 3; the millisecond-scale workloads mul_cleanup / gf2: 20 / 10)')
    ap.add_argument('--warmup', type=int, default=None, help='untimed steps before them (default 1; mul_cleanup / gf2: 3 / 2)')
    ap.add_argument('--qubits', type=int, default=1000)
    ap.add_argument('--left-terms', type=int, default=100000, help='left terms PER GPU')
    ap.add_argument('--right-terms', type=int, default=100000)
    ap.add_argument('--slab-rows', type=int, default=256, help='outer rows per output slab')
    ap.add_argument('--workload', choices=['product', 'adjacency', 'mul_cleanup', 'rotation', 'gf2'], default='product',
                    help="product (default, the north-star metric); adjacency = BASELINE cfg5: commutes_termwise adjacency of a fixed "
                         "--adj-terms x --adj-qubits operator, left-term axis sharded over the ranks (strong scaling); mul_cleanup = cfg3; "
                         "rotation = cfg2; gf2 = cfg4 (replicas only: every rank runs the whole workload)")
    ap.add_argument('--adj-terms', type=int, default=200000)
    ap.add_argument('--adj-qubits', type=int, default=2000)
    ap.add_argument('--adj-slab-rows', type=int, default=0, help='adjacency: rows per launch / output slab (0: as many as a quarter of the free HBM holds)')
    ap.add_argument('--no-extras', action='store_true')
    ap.add_argument('--single-process', action='store_true', help='drive all --gpus devices from THIS process (symgpu_init_all + grouped RCCL all-gather); '
                    'the default whenever --gpus > 1 and no launcher has set WORLD_SIZE')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-api', action='store_true', help='skip the `api` objects (the reference\'s call spelling timed on the drop-in classes)')
    ap.add_argument('--cpu-full', action='store_true', help='gf2: time the CPU restatement on the full 4000 x 54000 matrix (~70 s) instead of citing the cached run')
    args = ap.parse_args()
    if os.environ.get('BENCH_CPU_FULL', '0') == '1':
        args.cpu_full = True                                      # the driver passes no flags: the environment switches the full-size CPU leg on
    # a millisecond-scale step needs more than three of them for a stable figure (clocks, allocator): mul_cleanup 20 + 3, gf2 10 + 2
    d_steps, d_warm = {'mul_cleanup': (20, 3), 'gf2': (10, 2)}.get(args.workload, (3, 1))
    if args.steps is None: args.steps = d_steps
    if args.warmup is None: args.warmup = d_warm
    return args


def main():
    args = parse()
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus or world == 1, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if world == 1 and (args.single_process or (args.gpus > 1 and 'WORLD_SIZE' not in os.environ)):
        # no launcher: ONE process drives all devices — same workload, same contract line
        out = single_process_product(args)
        emit(out)
        return

    from symmer_amd import _lib, kernels
    from symmer_amd.kernels import DeviceOp
    from symmer_amd import parallel

    _lib.init(local_rank)
    lib = _lib.lib()
    comm = parallel.Communicator.from_env()       # TCP control plane + RCCL data plane when world > 1
    if args.workload != 'product':
        fn = {'adjacency': adjacency, 'mul_cleanup': wl_mul_cleanup, 'rotation': wl_rotation, 'gf2': wl_gf2}[args.workload]
        try:
            out = fn(args, comm, rank, world, _lib, DeviceOp, parallel)
        except parallel.CollectiveHang as exc:
            collective_hang(exc, args, rank, world)
        comm.close()
        if rank == 0:
            out['degraded_kernels'] = _lib.degraded()     # fast paths that gave up in this process (in-kernel wait timed out): [] on a healthy box
            emit(out)
        comm.hard_exit_if_hung()
        return

    n, Ni, M = args.qubits, args.left_terms, args.right_terms
    wq = (n + 63) // 64
    left = DeviceOp.random(Ni, n, 0.3, seed=1234 + 7919 * rank)
    # right operand: every rank owns a 1/world shard; the step all-gathers it
    Ts = (M + world - 1) // world
    my_rows = max(0, min(Ts, M - rank * Ts))
    if comm.gathers:
        # the step all-gathers the right operand: RCCL over xGMI, or — if RCCL could not be brought up on every rank — the
        # same exchange staged through host memory (comm.degraded says so and the JSON line carries it).  There is no
        # silent "every rank generates everything" mode: a step without the exchange would not be the benchmark.
        shard = parallel.padded_random_shard(my_rows, Ts, n, 99991 + rank)
        right = DeviceOp.alloc(Ts * world, wq, with_coeff=True)
    else:
        assert world == 1, 'multi-rank run without a data plane'
        shard = right = DeviceOp.random(my_rows, n, 0.3, seed=99991 + rank)
    slab = max(1, min(args.slab_rows, M))

    def full_sync():
        # device-wide synchronisation == torch.cuda.synchronize(); torch's CUDA runtime is deliberately NOT initialised in
        # this process (PyTorch wheels bundle their own HIP runtime; the product uses the system one)
        _lib.check(lib.symgpu_device_sync())

    def timed_product(left_op, n_left, slab_rows, steps, warmup):
        """warmup + `steps` timed steps of (all-gather when world > 1) + the product of `left_op` against the whole right operand in
        output slabs of `slab_rows` outer rows -> (max-over-ranks seconds, launches of the row kernel, their summed HIP-event ms)."""
        ring = [DeviceOp.alloc(slab_rows * n_left, wq, with_coeff=True) for _ in range(2)]

        def step():
            if comm.gathers:
                comm.allgather_op(shard, right, M)
            k = 0
            for o0 in range(0, M, slab_rows):
                o1 = min(M, o0 + slab_rows)
                _lib.check(lib.symgpu_mul_allpairs_dev(left_op.handle, right.handle, o0, o1, 1, ring[k & 1].handle))
                k += 1
        for _ in range(warmup):
            step()
        full_sync(); comm.barrier()
        _lib.check(lib.symgpu_prof_enable(0, 1))
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        full_sync(); comm.barrier()
        dt_ = time.perf_counter() - t0
        _lib.check(lib.symgpu_prof_enable(0, 0))
        nl_, ms_ = prof_read(_lib, 0)
        for r in ring:
            r.free()
        return comm.max_over_ranks(dt_), nl_, ms_

    if comm.gathers:
        # one gather outside the timed region, under a watchdog and agreed on by all ranks (a collective that hangs on one rank ends
        # the job with an error line instead of a hung job), then checked bit for bit against the same shards gathered through host
        # memory: a data plane that delivers wrong rows is replaced by the host-staged one and the line is flagged `degraded`
        try:
            comm.allgather_op(shard, right, M)
        except parallel.CollectiveHang as exc:
            collective_hang(exc, args, rank, world)
        comm.verify_allgather(shard, right, M)
    dt, n_launch, tot_ms = timed_product(left, Ni, slab, args.steps, args.warmup)

    pairs_per_step_rank = Ni * M
    value = world * pairs_per_step_rank * args.steps / dt
    launch_pairs = pairs_per_step_rank * args.steps / max(1, n_launch)
    launch_ms = tot_ms / max(1, n_launch)
    per_pair = 16 * wq                                   # the dominant kernel streams the rows (+ 1 phase byte per pair, not counted)
    fused_ok = os.environ.get('SYMGPU_PRODUCT_FUSED', '1') != '0' and wq & (wq - 1) == 0 and wq <= 64
    ROW_KERNEL = 'k_mul_rows_e' if fused_ok else 'k_mul_rows'
    algo_bytes_launch = launch_pairs * per_pair
    achieved = algo_bytes_launch / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0

    out = {
        'metric': 'pauli_term_pairs_per_sec', 'value': value, 'unit': 'pairs/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'u64', 'data': 'synthetic',
        'config': {'workload': 'allpairs_product', 'n_qubits': n, 'left_terms_per_gpu': Ni, 'right_terms': M,
                   'pairs_per_step': world * pairs_per_step_rank, 'bytes_per_pair': 16 * wq + 16, 'slab_rows': slab,
                   'parallelism': (f'left-axis shard x{world}, all-gather of right rows ({comm.data_plane})' if world > 1 else 'single GPU')},
        'roofline': {'bound': 'hbm', 'kernel': ROW_KERNEL, 'bytes_per_pair': per_pair,
                     'note': 'one output row per block, grid.x a multiple of 8 so that every XCD keeps its eighth of the inner operand in L2; '
                             'the row stream also forms the phase sums (DPP + v_bcnt, VALU otherwise idle) and leaves 1 B/pair, '
                             'k_mul_coeff_expand streams the 16 B/pair coefficients after it (whole_step_GBps counts both kernels, 272 B/pair)', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBS, 'traffic': None, 'launches': n_launch, 'avg_launch_ms': launch_ms,
                     'algorithmic_bytes_per_launch': algo_bytes_launch,
                     'whole_step_GBps': pairs_per_step_rank * (16 * wq + 16) / (dt / args.steps) / 1e9},
    }
    if comm.degraded:
        out['degraded'] = comm.degraded          # top-level flag: the all-gather did not run over RCCL/xGMI
    # the plain row stream k_mul_rows (rows-only output slab: no phase bytes, no coefficients), outside the timed region: four warm
    # launches (a fresh 6.5 GB slab: first touch), then the best of three batches of eight
    if rank == 0:
        rows_only = DeviceOp.alloc(slab * Ni, wq, with_coeff=False)
        o1 = min(M, slab)
        for _ in range(4):
            _lib.check(lib.symgpu_mul_allpairs_dev(left.handle, right.handle, 0, o1, 1, rows_only.handle))
        kernels.sync()
        best = None
        for _ in range(3):
            _lib.check(lib.symgpu_prof_enable(0, 1))
            for _ in range(8):
                _lib.check(lib.symgpu_mul_allpairs_dev(left.handle, right.handle, 0, o1, 1, rows_only.handle))
            kernels.sync()
            _lib.check(lib.symgpu_prof_enable(0, 0))
            n2, ms2 = prof_read(_lib, 0)
            if n2 and (best is None or ms2 / n2 < best):
                best = ms2 / n2
        iso = (o1 * Ni * 16 * wq) / (best * 1e-3) / 1e9 if best else 0.0
        out['roofline']['isolated_GBps'] = iso
        out['roofline']['isolated_frac'] = iso / HBM_PEAK_GBS
        out['roofline']['isolated_avg_launch_ms'] = best
        rows_only.free()
    # the library's own one-shot 16-byte fill / copy over 4 GiB buffers: a PROBE of the box, not a ceiling (the row stream's
    # one-row-per-block write pattern beats the probe's grid-stride fill)
    if rank == 0:
        fill, cp = ctypes.c_double(0), ctypes.c_double(0)
        _lib.check(lib.symgpu_membw_probe(4 << 30, ctypes.addressof(fill), ctypes.addressof(cp)))
        out['roofline']['fill_probe_GBps'] = fill.value
        out['roofline']['copy_probe_GBps'] = cp.value
    # HBM traffic of the dominant kernel: hardware counters cannot be read from inside this process, so the figure comes from
    # the committed rocprofv3 --pmc profile of THIS workload — and only if that profile was taken on the kernel source that
    # is running now (sha256 of product.hip recorded next to it); otherwise traffic stays null.
    traffic_from_profile(out['roofline'], TRAFFIC_PROFILE, ['product.hip'],
                         {'n_qubits': n, 'left_terms_per_gpu': Ni, 'right_terms': M, 'slab_rows': slab})
    # STRONG scaling next to the weak line (world > 1): the same 10^5 x 10^5 product with the LEFT axis split over the ranks — 1/world of the
    # left terms per GPU, world x as many outer rows per output slab, so that a launch still writes the same ~6.5 GB.
    if world > 1:
        try:
            ni_s = (Ni + world - 1) // world
            left_s = DeviceOp.random(ni_s, n, 0.3, seed=4321 + 7919 * rank)
            dt_s, nl_s, ms_s = timed_product(left_s, ni_s, min(M, slab * world), args.steps, max(1, args.warmup))
            left_s.free()
            out['strong_scaling'] = {'scaling': 'strong', 'left_terms_per_gpu': ni_s, 'right_terms': M, 'slab_rows': min(M, slab * world),
                                     'pairs_per_step': world * ni_s * M, 'value': world * ni_s * M * args.steps / dt_s, 'unit': 'pairs/s',
                                     'ms_per_step': dt_s / args.steps * 1e3, 'row_kernel_avg_launch_ms': ms_s / max(1, nl_s),
                                     'note': 'total work fixed at 1e5 x 1e5 terms; `value` of the contract line above is the weak-scaling figure'}
        except Exception as exc:                                  # noqa: BLE001 - reported, never costs the headline line
            out['strong_scaling'] = {'error': f'{type(exc).__name__}: {exc}'}

    # the secondary measurements must never cost the headline line: a failure is reported in place of the numbers
    if rank == 0 and world == 1 and not args.no_extras:
        try:
            out['extras'] = extras(_lib, kernels, DeviceOp, comm, parallel, args, out)
        except Exception as exc:                                  # noqa: BLE001 - reported, not swallowed
            out['extras'] = {'error': f'{type(exc).__name__}: {exc}'}
    if rank == 0 and world == 1 and not args.no_cpu:
        try:
            out['cpu_baseline'] = cpu_baseline(n)
        except Exception as exc:                                  # noqa: BLE001
            out['cpu_baseline'] = {'error': f'{type(exc).__name__}: {exc}'}
    comm.close()
    if rank == 0:
        out['degraded_kernels'] = _lib.degraded()         # fast paths that gave up in this process (in-kernel wait timed out): [] on a healthy box
        emit(out)
    comm.hard_exit_if_hung()


def single_process_product(args):
    """The north-star product with `--gpus N` devices driven by ONE process (symmer_amd/multi.py's HipBackend: symgpu_init_all, one stream
    per device, ncclCommInitAll, the per-device all-gathers of a step issued from this thread as one RCCL group).  Per step and device: the
    grouped all-gather of the right operand's 1/N shards, then the product of the device's own 1e5 left terms against all right terms in
    output slabs; the launches of a slab go to every device before the next slab starts, so the devices run side by side.  Weak scaling,
    `value` = all devices' pairs / wall time between two all-device synchronisations."""
    from symmer_amd import _lib, multi, parallel
    from symmer_amd.kernels import DeviceOp
    G = max(1, args.gpus)
    if G == 1:
        os.environ.setdefault('SYMGPU_FORCE_COMM', '1')             # one device: still through ncclCommInitAll and the grouped all-gather
    if _lib.device_count() < G:
        return {'metric': 'pauli_term_pairs_per_sec', 'value': None, 'unit': 'pairs/s', 'n_gpus': G, 'steps': args.steps, 'warmup': args.warmup,
                'error': f'--gpus {G} in one process but only {_lib.device_count()} device(s) visible', 'config': {'workload': 'allpairs_product'}}
    be = multi.HipBackend(G)
    lib = _lib.load()
    n, Ni, M = args.qubits, args.left_terms, args.right_terms
    wq = (n + 63) // 64
    ts, bounds = parallel.shard_bounds(M, G)
    slab = max(1, min(args.slab_rows, M))
    lefts, shards, fulls, rings = [], [], [], []
    for d in range(G):
        be.use(d)
        lefts.append(DeviceOp.random(Ni, n, 0.3, seed=1234 + 7919 * d))
        sh = DeviceOp.random(ts, n, 0.3, seed=99991 + d); sh.set_rows(bounds[d][1] - bounds[d][0])
        shards.append(sh)
        fulls.append(DeviceOp.alloc(ts * G, wq, with_coeff=True))
        rings.append([DeviceOp.alloc(slab * Ni, wq, with_coeff=True) for _ in range(2)])

    def sync_all():
        for d in range(G):
            be.use(d)
            _lib.check(lib.symgpu_device_sync())

    def step():
        be.allgather(shards, fulls, M)
        k = 0
        for o0 in range(0, M, slab):
            o1 = min(M, o0 + slab)
            for d in range(G):
                _lib.check(lib.symgpu_mul_allpairs_dev(lefts[d].handle, fulls[d].handle, o0, o1, 1, rings[d][k & 1].handle))   # runs on the handles' device
            k += 1
    for _ in range(args.warmup):
        step()
    sync_all()
    for d in range(G):
        be.use(d); _lib.check(lib.symgpu_prof_enable(0, 1))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    dt = time.perf_counter() - t0
    n_launch, tot_ms = 0, 0.0
    for d in range(G):
        be.use(d); _lib.check(lib.symgpu_prof_enable(0, 0))
        nl, ms = prof_read(_lib, 0)
        n_launch += nl; tot_ms += ms
    launch_ms = tot_ms / max(1, n_launch)
    launch_pairs = G * Ni * M * args.steps / max(1, n_launch)
    achieved = launch_pairs * 16 * wq / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
    out = {'metric': 'pauli_term_pairs_per_sec', 'value': G * Ni * M * args.steps / dt, 'unit': 'pairs/s', 'n_gpus': G, 'steps': args.steps,
           'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'u64',
           'data': 'synthetic',
           'config': {'workload': 'allpairs_product', 'n_qubits': n, 'left_terms_per_gpu': Ni, 'right_terms': M, 'pairs_per_step': G * Ni * M,
                      'bytes_per_pair': 16 * wq + 16, 'slab_rows': slab,
                      'parallelism': f'ONE process driving {G} device(s): left-axis shard, grouped all-gather of right rows '
                                     f'({"rccl (ncclCommInitAll + ncclGroupStart/End)" if be.degraded is None else "host-staged"})'},
           'roofline': {'bound': 'hbm', 'kernel': 'k_mul_rows_e', 'bytes_per_pair': 16 * wq, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                        'frac': achieved / HBM_PEAK_GBS, 'traffic': None, 'launches': n_launch, 'avg_launch_ms': launch_ms,
                        'algorithmic_bytes_per_launch': launch_pairs * 16 * wq, 'note': 'per-launch HIP events on every device\'s own stream, averaged over the devices',
                        'whole_step_GBps_per_gpu': Ni * M * (16 * wq + 16) / (dt / args.steps) / 1e9},
           'degraded_kernels': _lib.degraded()}
    if be.degraded:
        out['degraded'] = be.degraded
    traffic_from_profile(out['roofline'], TRAFFIC_PROFILE, ['product.hip'], {'n_qubits': n, 'left_terms_per_gpu': Ni, 'right_terms': M, 'slab_rows': slab})
    if not args.no_cpu:
        try:
            out['cpu_baseline'] = cpu_baseline(n)
        except Exception as exc:                                      # noqa: BLE001
            out['cpu_baseline'] = {'error': f'{type(exc).__name__}: {exc}'}
    return out


def _sig(x, digits=4):
    """A number with `digits` significant figures (the summary has to stay short), None as it is."""
    if x is None or isinstance(x, (str, bool)):
        return x
    try:
        return float(f'{float(x):.{digits}g}')
    except (TypeError, ValueError):
        return None


def _roof(r):
    return [r.get('kernel'), _sig(r.get('frac'), 3)] if isinstance(r, dict) else None


def _api(a):
    return [_sig(a.get('a_result_object_seconds')), _sig(a.get('b_host_arrays_out_seconds'))] if isinstance(a, dict) and 'a_result_object_seconds' in a else None


def summary_of(out):
    """Every config's figure in one compact object, printed LAST in the line: the driver keeps the tail of stdout, and the line is > 8 KB.
    Per entry: v = value, u = unit, s = seconds per step (C ABI, operands resident), k = [dominant kernel, roofline fraction], api = [a, b]
    seconds (see `api_legend`), cpu = the CPU baseline's value in the same unit."""
    ex = out.get('extras') if isinstance(out.get('extras'), dict) else {}
    cpu = out.get('cpu_baseline') if isinstance(out.get('cpu_baseline'), dict) else {}
    oc_ = cpu.get('other_configs', {}) if isinstance(cpu.get('other_configs'), dict) else {}
    g = lambda d, *ks: (g(d.get(ks[0]), *ks[1:]) if len(ks) > 1 else d.get(ks[0])) if isinstance(d, dict) else None
    sm = {'legend': 'v value, u unit, s seconds/step, k [kernel, roofline frac], api [a, b] seconds, cpu = CPU baseline in u'}
    if out.get('config', {}).get('workload') == 'allpairs_product':
        sm['product_1e5x1e5'] = {'v': _sig(out.get('value')), 'u': 'pairs/s', 's': _sig(out.get('ms_per_step', 0) * 1e-3), 'k': _roof(out.get('roofline')),
                                 'cpu': _sig(cpu.get('value'))}
    else:
        sm[out.get('config', {}).get('workload', 'workload')] = {'v': _sig(out.get('value')), 'u': out.get('unit'), 's': _sig(out.get('ms_per_step', 0) * 1e-3),
                                                                  'k': _roof(out.get('roofline')), 'api': _api(out.get('api')) or _api(g(out, 'api', 'full')),
                                                                  'cpu': _sig(g(out, 'cpu_baseline', 'value'))}
        if g(out, 'roofline', 'lds', 'frac') is not None:
            sm[out['config']['workload']]['lds_frac'] = _sig(g(out, 'roofline', 'lds', 'frac'), 3)
    c1, c2, c3, c4, c5, ss = (ex.get(k) for k in ('cfg1_api_mul', 'cfg2_rotation', 'cfg3_mul_cleanup', 'cfg4_symmetry_kernel', 'cfg5_adjacency', 'strong_scaling_shard'))
    if isinstance(c1, dict) and 'error' not in c1:
        sm['cfg1_mul_500t_100q'] = {'v': _sig(c1.get('pairs_per_s')), 'u': 'pairs/s', 's': _sig(c1.get('seconds')), 'api': _api(c1.get('api')),
                                    'cpu': _sig(g(oc_, 'cfg1_mul_cleanup', 'pairs_per_s'))}
    if isinstance(c2, dict) and 'error' not in c2:
        sm['cfg2_rotation'] = {'v': _sig(c2.get('term_pairs_per_s')), 'u': 'term-pairs/s', 's': _sig(c2.get('seconds_per_rotation')), 'k': _roof(c2.get('roofline')),
                               'api': _api(c2.get('api')), 'cpu': _sig(g(oc_, 'cfg2_rotation', 'term_pairs_per_s')),
                               'clifford_s': _sig(g(c2, 'clifford', 'run_of_200_seconds_per_rotation')), 'saturated_chain_s': _sig(g(c2, 'saturated_chain', 'seconds_per_rotation'))}
    if isinstance(c3, dict) and 'error' not in c3:
        sm['cfg3_mul_cleanup'] = {'v': _sig(c3.get('pairs_per_s')), 'u': 'pairs/s', 's': _sig(c3.get('seconds')), 'k': _roof(c3.get('roofline')),
                                  'api': _api(c3.get('api')), 'cpu': _sig(g(oc_, 'cfg3_sample_mul_cleanup', 'pairs_per_s'))}
    if isinstance(c4, dict) and 'error' not in c4:
        sm['cfg4_gf2'] = {'v': _sig(c4.get('row_xors_per_s')), 'u': 'row-XORs/s', 's': _sig(c4.get('seconds')), 'k': _roof(c4.get('roofline')),
                          'api': _api(c4.get('api')), 'cpu': _sig(g(oc_, 'cfg4_full_size_cached', 'row_xors_per_s') or g(oc_, 'cfg4_sample_rref', 'row_xors_per_s'))}
    if isinstance(c5, dict) and 'error' not in c5:
        rs = c5.get('rank_share_25000_rows') or {}
        sm['cfg5_adjacency'] = {'v': _sig(c5.get('pairs_per_s')), 'u': 'pairs/s', 's': _sig(c5.get('seconds')), 'k': _roof(c5.get('roofline')),
                                'lds_frac': _sig(g(c5, 'roofline', 'lds', 'frac'), 3), 'api_full': _api(g(c5, 'api', 'full')), 'api_share': _api(g(c5, 'api', 'rank_share')),
                                'cpu': _sig(g(oc_, 'cfg5_sample_commutation', 'pairs_per_s')),
                                'rank_share_s': _sig(rs.get('seconds')), 'predicted_8gpu_x': _sig(rs.get('predicted_8gpu_strong_speedup_before_allgather'), 3)}
    if isinstance(ss, dict) and 'error' not in ss:
        sm['strong_scaling_shard'] = {'s': _sig(ss.get('ms_per_step', 0) * 1e-3), 'k': _roof(ss.get('roofline')),
                                      'predicted_8gpu_x': _sig(ss.get('predicted_8gpu_strong_speedup_before_allgather'), 3)}
    errs = [k for k, v in ex.items() if isinstance(v, dict) and 'error' in v]
    if errs:
        sm['failed_sections'] = errs
    return sm


def emit(out):
    """The ONE JSON line: `api_legend` once, `summary` last (the last 4 KB of stdout carry every config's figure)."""
    out.pop('summary', None)
    out['api_legend'] = API_LEGEND
    out['summary'] = summary_of(out)
    print(json.dumps(out))


def collective_hang(exc, args, rank, world):
    """A collective did not return on some rank (agreed by all ranks): the library stream of that rank is lost, there is nothing
    to fall back to.  Rank 0 prints an error line in place of the result and every rank leaves with status 3."""
    if rank == 0:
        print(json.dumps({'metric': 'pauli_term_pairs_per_sec', 'value': None, 'unit': 'pairs/s', 'n_gpus': world, 'steps': args.steps,
                          'warmup': args.warmup, 'error': f'CollectiveHang: {exc}', 'config': {'workload': args.workload}}))
    sys.stdout.flush(); sys.stderr.flush()
    os._exit(3)


def cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def traffic_from_profile(roof, name, sources, config):
    """roofline.traffic from the committed rocprofv3 --pmc profile `profiles/<name>` — only if it was taken on the kernel sources
    that are running now (sha256 recorded next to the counters) and on the same configuration; otherwise null with the reason."""
    import hashlib
    try:
        with open(os.path.join(ROOT, 'profiles', name)) as f:
            tr = json.load(f)
        h = hashlib.sha256()
        for src in sources:
            with open(os.path.join(ROOT, 'symmer_amd', 'csrc', src), 'rb') as f:
                h.update(f.read())
        if tr.get('config') != config:
            roof['traffic_source'] = f'null: profiles/{name} was taken on a different workload'
        elif tr.get('kernel_source_sha256') != h.hexdigest():
            roof['traffic_source'] = f'null: profiles/{name} was taken on a different revision of {"/".join(sources)}'
        else:
            roof['traffic'] = tr['write_bytes_per_launch'] + tr['fetch_bytes_per_launch_corrected_x2']
            roof['traffic_write_bytes'] = tr['write_bytes_per_launch']
            roof['traffic_fetch_bytes_x2'] = tr['fetch_bytes_per_launch_corrected_x2']
            roof['traffic_source'] = f'profiles/{name} (rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE, separate passes, same {"/".join(sources)})'
    except (OSError, KeyError, ValueError) as exc:
        roof['traffic_source'] = f'null: no usable traffic profile ({type(exc).__name__})'


def prof_read(_lib, cls):
    n, ms = ctypes.c_int64(0), ctypes.c_double(0)
    _lib.check(_lib.lib().symgpu_prof_read(cls, ctypes.addressof(n), ctypes.addressof(ms)))
    return n.value, ms.value


def contract_line(args, world, value, unit, metric, dt, dtype, workload_cfg, roofline, comm):
    out = {'metric': metric, 'value': value, 'unit': unit, 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
           'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': dtype,
           'data': 'synthetic', 'config': workload_cfg, 'roofline': roofline}
    if comm.degraded:
        out['degraded'] = comm.degraded
    return out


# ------------------------------------------------------------------------------------------------------------------
def wl_mul_cleanup(args, comm, rank, world, _lib, DeviceOp, parallel):
    """BASELINE cfg3 — the call `P * P` of the reference (base.py:821-859 -> _multiply_by_operator :764-794 -> symplectic_cleanup
    utils.py:230-279): a 10,000-term, 1,000-qubit operator squared (1e8 pairs) + cleanup, device resident.  Step = one fused
    product + cleanup (symgpu_mul_cleanup_dev); the 1e8 product rows are never materialised.
    roofline: the kernel that moves most of the step's bytes, the fused output stage k_emit_fused (every kept row and its coefficient
    written once: 16 Wq + 16 bytes per row, HBM-write bound; beside the stream it only reads bitmaps and the cache-resident operands).  SURVEY §8d's algorithmic bytes of the
    whole step (T (16 Wq + 16) read + U_kept (16 Wq + 16) written) are reported next to it — most of them never move here."""
    lib = _lib.lib()
    n, N = args.qubits, 10000
    wq = (n + 63) // 64
    A = DeviceOp.random(N, n, 0.3, seed=1237 + 7919 * rank)
    n_out = [0]

    def step():
        h = ctypes.c_void_p()
        _lib.check(lib.symgpu_mul_cleanup_dev(A.handle, A.handle, 1, 1e-15, 1, ctypes.byref(h)))
        r = DeviceOp(h); n_out[0] = r.n_terms; r.free()
    for _ in range(max(1, args.warmup)):
        step()
    _lib.check(lib.symgpu_device_sync()); comm.barrier()
    _lib.check(lib.symgpu_prof_enable(3, 1))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    _lib.check(lib.symgpu_device_sync()); comm.barrier()
    dt = comm.max_over_ranks(time.perf_counter() - t0)
    _lib.check(lib.symgpu_prof_enable(3, 0))
    nl, ms = prof_read(_lib, 3)
    pairs = N * N
    row_bytes = 16 * wq
    launch_bytes = n_out[0] * (row_bytes + 16) * args.steps / max(1, nl)   # rows + coefficients written by the one k_emit_fused launch of a step
    kt = ms / max(1, nl) * 1e-3
    algo_step = (pairs + n_out[0]) * (row_bytes + 16)
    roof = {'bound': 'hbm', 'kernel': 'k_emit_fused', 'achieved': launch_bytes / kt / 1e9 if kt else None, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': launch_bytes / kt / 1e9 / HBM_PEAK_GBS if kt else None, 'traffic': None, 'launches': nl, 'avg_launch_ms': kt * 1e3,
            'algorithmic_bytes_per_launch': launch_bytes,
            'note': 'output stage of the cleanup in one launch: 16*Wq + 16 bytes per kept row, written once (non-temporal stores), rows '
                    'gathered from the L2-resident operand; the other kernels of the step are key generation, the marking of single keys and the '
                    f'search for the pairs that share a product row (pair_dups.hip: operand hash tables in LDS, no sort) — profiles/{PROFILE_TAG}_cfg3_kernel_trace.txt',
            'whole_step': {'survey_8d_algorithmic_bytes': algo_step, 'algorithmic_GBps': algo_step / (dt / args.steps) / 1e9,
                           'physical_write_floor_ms': n_out[0] * (row_bytes + 16) / (HBM_PEAK_GBS * 1e9) * 1e3,
                           'note': 'SURVEY 8d counts T (16Wq+16) B read + U_kept (16Wq+16) B written; the T product rows and pair coefficients '
                                   'never exist in memory here (keys only), so this figure is not HBM traffic'}}
    traffic_from_profile(roof, f'{PROFILE_TAG}_cfg3_traffic.json', ['cleanup.hip'], {'workload': 'mul_cleanup', 'n_qubits': n, 'terms': N})
    out = contract_line(args, world, world * pairs * args.steps / dt, 'pairs/s', 'pauli_term_pairs_per_sec', dt, 'u64+c128',
                        {'workload': 'mul_cleanup_squared', 'n_qubits': n, 'terms': N, 'pairs_per_step': pairs, 'terms_out': n_out[0],
                         'call': 'P * P (symgpu_mul_cleanup_dev: fused product + cleanup, squared-operator path)',
                         'parallelism': f'{world} independent replicas' if world > 1 else 'single GPU'}, roof, comm)
    A.free()
    if rank == 0 and not getattr(args, 'no_api', False):
        # reference spelling: P = PauliwordOp.random(n, N); P * P   (base.py:821 -> :764 -> utils.py:230)
        out['api'] = guarded(lambda: api_block('P * P', lambda: host_operator(N, n, 1237), lambda P: P * P, lambda R: (R.packed, R.coeff_vec), 5,
                                               dt / args.steps, note='b reads the result as packed rows + coefficients (6.8 GB); its symp_matrix '
                                               'would be 50 GB of bools'))
    if rank == 0 and not args.no_cpu:
        out['cpu_baseline'] = guarded(lambda: cpu_mul_cleanup(n))
    return out


def wl_rotation(args, comm, rank, world, _lib, DeviceOp, parallel):
    """BASELINE cfg2 — `P._rotate_by_single_Pword(Q, 0.3)` (reference base.py:1090-1161) of a 100,000-term, 1,000-qubit operator,
    device resident, duplicate status and row hashes known (as inside perform_rotations from the second step on).  Step = one
    non-Clifford rotation = ONE persistent launch (k_rot_resident, rotate_resident.hip).  Unit: term pairs (N x 1) per second.
    roofline: SURVEY 8d's bytes of that launch, N (16 Wq + 16) read + N_out (16 Wq + 16) written, over its HIP-event duration."""
    from symmer_amd import kernels, packing
    lib = _lib.lib()
    n, N = args.qubits, 100000
    wq = (n + 63) // 64
    rng = np.random.default_rng(1236)
    P = DeviceOp.random(N, n, 0.3, seed=1236 + 7919 * rank)
    qs = [packing.pack_rows((rng.random((1, 2 * n)) < 0.3))[0] for _ in range(8)]
    kernels.rotate_single_dev(P, qs[0], 0.3)[0].free()                   # multi-launch path once: duplicate status + hashes of P
    n_out = [0]

    from symmer_amd.kernels import rotation_args
    cos_t, sin_t, ck = rotation_args(0.3)
    keep = []

    def step(k=0):
        # the C-ABI call itself (symgpu_rotate_single_dev) with its arguments prepared once, as perform_rotations' inner loop would
        h, allc = ctypes.c_void_p(), ctypes.c_int(0)
        _lib.check(lib.symgpu_rotate_single_dev(P.handle, qs[k % 8].ctypes.data, cos_t, sin_t, ck, 1e-15, ctypes.byref(h), ctypes.addressof(allc)))
        if keep:
            keep.pop().free()
        keep.append(DeviceOp(h))
    ROT = 100                                                            # per-rotation time over >= 100 rotations, operator device resident
    for k in range(max(1, args.warmup) * ROT):
        step(k)
    _lib.check(lib.symgpu_device_sync()); comm.barrier()
    t0 = time.perf_counter()
    for k in range(args.steps * ROT):
        step(k)
    _lib.check(lib.symgpu_device_sync()); comm.barrier()
    dt = comm.max_over_ranks(time.perf_counter() - t0)
    # the kernel's duration from HIP events around every launch: a second pass of the same steps, because the two event records per
    # launch are not free next to a 27 us kernel (they would sit inside every rotation of the timed region)
    _lib.check(lib.symgpu_prof_enable(4, 1))
    for k in range(args.steps * ROT):
        step(k)
    _lib.check(lib.symgpu_device_sync())
    _lib.check(lib.symgpu_prof_enable(4, 0))
    nl, ms = prof_read(_lib, 4)
    kt = ms / max(1, nl) * 1e-3
    n_out[0] = keep[-1].n_terms
    keep.pop().free()
    launch_bytes = (N + n_out[0]) * (16 * wq + 16)
    roof = {'bound': 'hbm', 'kernel': 'k_rot_resident', 'achieved': launch_bytes / kt / 1e9 if kt else None, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': launch_bytes / kt / 1e9 / HBM_PEAK_GBS if kt else None, 'traffic': None, 'launches': nl, 'avg_launch_ms': kt * 1e3,
            'algorithmic_bytes_per_launch': launch_bytes, 'whole_call_GBps': launch_bytes / (dt / (args.steps * ROT)) / 1e9,
            'events_pass': 'a second pass of the same steps right after the timed region (events inside every 27 us call would be part of it)',
            'note': 'one persistent launch per rotation, rows resident in LDS (one workgroup per CU); the launch is a chain of dependent phases '
                    '(rows in 7 us, join-table compare-and-swaps 3 us, two in-launch all-gathers, rows out 7 us), not a bandwidth-bound stream'}
    traffic_from_profile(roof, f'{PROFILE_TAG}_rotation_traffic.json', ['rotate_resident.hip'], {'workload': 'rotation', 'n_qubits': n, 'terms': N})
    out = contract_line(args, world, world * N * ROT * args.steps / dt, 'pairs/s', 'pauli_term_pairs_per_sec', dt, 'u64+c128',
                        {'workload': 'single_pauli_rotation_nonclifford', 'n_qubits': n, 'terms': N, 'terms_out': n_out[0], 'angle': 0.3,
                         'rotations_per_step': ROT, 'pairs_per_step': N * ROT, 'call': 'P._rotate_by_single_Pword(Q, 0.3) (symgpu_rotate_single_dev), one call per rotation',
                         'parallelism': f'{world} independent replicas' if world > 1 else 'single GPU'}, roof, comm)
    out['seconds_per_rotation'] = dt / (args.steps * ROT)
    # Clifford rotations of the same operator: one by one, and as one run (rows in registers + one sort per 40 rotations)
    if rank == 0 and not args.no_extras:
        def clifford():
            res = {}
            t0 = time.perf_counter()
            for k in range(20):
                kernels.rotate_single_dev(P, qs[k % 8], np.pi / 2)[0].free()
            kernels.sync()
            res['one_by_one_seconds_per_rotation'] = (time.perf_counter() - t0) / 20
            Pc = kernels.cleanup_dev(P)
            q200 = np.vstack([qs[k % 8] for k in range(200)]); k200 = (np.arange(200) % 4).astype(np.int32)
            kernels.rotate_clifford_chain_dev(Pc, q200[:45], k200[:45]).free(); kernels.sync()
            _lib.check(lib.symgpu_prof_enable(5, 1))
            t0 = time.perf_counter(); kernels.rotate_clifford_chain_dev(Pc, q200, k200).free(); kernels.sync()
            res['run_of_200_seconds_per_rotation'] = (time.perf_counter() - t0) / 200
            _lib.check(lib.symgpu_prof_enable(5, 0))
            nl5, ms5 = prof_read(_lib, 5)
            res['k_cchain_reg_launches'] = nl5
            res['k_cchain_reg_us_per_rotation'] = ms5 * 1e3 / 200
            Pc.free()
            return res
        out['clifford'] = guarded(clifford)

        def saturated_chain():
            # SURVEY 8d cfg2 asks for the per-rotation time over a CHAIN of >= 100 rotations.  A chain of non-Clifford rotations by random
            # Paulis grows the operator 1.5x per step; the realistic chain (Trotter circuits, symmer/evolution/exponentiation.py:26-38) cycles
            # through a fixed set of generators, under which the term set saturates: 800 seed terms x the products of 8 generators they reach (~10^5 terms).
            # Every rotation then MERGES (each anticommuting term's partner P.Q is already there): the term count stays constant.
            seed_op = DeviceOp.random(800, n, 0.3, seed=4242)
            cur, counts = seed_op, []
            for k in range(8 * 14):                                  # to the fixed point (the count stops growing after a few cycles)
                nxt = kernels.rotate_single_dev(cur, qs[k % 8], 0.3)[0]
                cur.free(); cur = nxt
                counts.append(cur.n_terms)
                if k % 8 == 7 and len(counts) > 8 and counts[-1] == counts[-9]:
                    break
            sat_terms, cycles = cur.n_terms, len(counts) // 8
            reps = 104
            for k in range(16):                                       # warm: allocator classes, join tables of this size
                nxt = kernels.rotate_single_dev(cur, qs[k % 8], 0.3)[0]; cur.free(); cur = nxt
            kernels.sync()
            t0 = time.perf_counter()
            for k in range(reps):
                nxt = kernels.rotate_single_dev(cur, qs[k % 8], 0.3)[0]; cur.free(); cur = nxt
            kernels.sync()
            t = (time.perf_counter() - t0) / reps
            end_terms = cur.n_terms
            cur.free()
            return {'generators': 8, 'seed_terms': 800, 'saturated_terms': sat_terms, 'cycles_to_saturation': cycles, 'timed_rotations': reps, 'terms_after': end_terms,
                    'seconds_per_rotation': t, 'term_pairs_per_s': sat_terms / t, 'angle': 0.3,
                    'call': 'cur = rotate(cur, Q_{k mod 8}, 0.3) 104 times on the saturated operator (kernels.rotate_single_dev: one C-ABI call per rotation, '
                            'device resident; every rotation merges rows, the term count is constant)'}
        out['saturated_chain'] = guarded(saturated_chain)
    P.free()
    if rank == 0 and not getattr(args, 'no_api', False):
        from symmer_amd import PauliwordOp
        Qs = [PauliwordOp(packing.unpack_rows(q.reshape(1, -1), n), [1]) for q in qs]
        out['api'] = guarded(lambda: api_block('P._rotate_by_single_Pword(Q, 0.3)', lambda: host_operator(N, n, 1236),
                                               lambda Pa: Pa._rotate_by_single_Pword(Qs[0], 0.3), lambda R: (R.packed, R.coeff_vec), 20,
                                               dt / (args.steps * ROT)))
        out['api_perform_rotations'] = guarded(lambda: api_block('P.perform_rotations([(Q_k, pi/2)] * 100)', lambda: host_operator(N, n, 1236),
                                                                 lambda Pa: Pa.perform_rotations([(Qs[k % 8], np.pi / 2) for k in range(100)]),
                                                                 lambda R: (R.packed, R.coeff_vec), 5, None,
                                                                 note='100 Clifford rotations in one call (a chain of 100 non-Clifford rotations by random Paulis '
                                                                      'grows the operator 1.5x per step)'))
    if rank == 0 and not args.no_cpu:
        out['cpu_baseline'] = guarded(lambda: cpu_rotation(n, N))
    return out


def cfg4_operator(_lib, DeviceOp, kernels, packing, seed):
    """2,000 qubits, 50,000 terms, 32 planted symmetries (X bits of qubits 0..31 zero), scrambled by 16 random
    Clifford rotations."""
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


def wl_gf2(args, comm, rank, world, _lib, DeviceOp, parallel):
    """BASELINE cfg4 — `IndependentOp.symmetry_generators(H)` (reference independent_op.py:90-144: _cref_binary of a 4000 x 54000 GF(2)
    matrix, utils.py:292-347).  Step = one symgpu_symmetry_kernel_dev call on the device-resident operator (matrix build, blocked
    elimination, read-out of the generators).  Unit: row-XORs per second, counted as the reference's loop performs them.
    roofline: the main sweep launches k_sweep_m4r — every launch reads and writes the whole matrix once (physical bytes); the
    blocked algorithm makes ~1/50 of the reference's passes, so SURVEY 8d's algorithmic figure (16 Wc bytes per row-XOR) is far above
    what moves and is reported separately."""
    from symmer_amd import kernels, packing
    lib = _lib.lib()
    H = cfg4_operator(_lib, DeviceOp, kernels, packing, 1238 + 7919 * rank)
    outg = np.zeros((4000, 64), dtype='<u8')
    k, nx = ctypes.c_int64(0), ctypes.c_int64(0)

    def step():
        _lib.check(lib.symgpu_symmetry_kernel_dev(H.handle, 2000, outg.ctypes.data, 4000, ctypes.addressof(k), ctypes.addressof(nx)))
    for _ in range(max(1, args.warmup)):
        step()
    _lib.check(lib.symgpu_device_sync()); comm.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    _lib.check(lib.symgpu_device_sync()); comm.barrier()
    dt = comm.max_over_ranks(time.perf_counter() - t0)
    # the sweep's duration from HIP events around every main sweep launch: a second pass of the same steps — 80 launches per step with
    # two event records each cost 15 % of the step, which would go into `value`
    _lib.check(lib.symgpu_prof_enable(2, 1))
    for _ in range(args.steps):
        step()
    _lib.check(lib.symgpu_device_sync())
    _lib.check(lib.symgpu_prof_enable(2, 0))
    nl, ms = prof_read(_lib, 2)
    kt = ms / max(1, nl) * 1e-3
    wc = (50000 + 63) // 64 + 64
    phys = 2 * 4000 * wc * 8                                               # one pass over the matrix: read + write
    algo_launch = nx.value * 16 * wc * args.steps / max(1, nl)
    roof = {'bound': 'hbm', 'kernel': 'k_sweep_m4r', 'achieved': phys / kt / 1e9 if kt else None, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': phys / kt / 1e9 / HBM_PEAK_GBS if kt else None, 'traffic': None, 'launches': nl, 'avg_launch_ms': kt * 1e3,
            'physical_bytes_per_launch': phys, 'survey_8d_algorithmic_bytes_per_launch': algo_launch,
            'events_pass': 'a second pass of the same steps right after the timed region (per-launch events cost 15 % of this step)',
            'survey_8d_algorithmic_GBps': nx.value * 16 * wc / (dt / args.steps) / 1e9,
            'note': 'achieved = bytes one sweep launch moves (the 27 MB matrix read and written once) / its duration; the matrix lives in the '
                    'Infinity Cache, and the launch is as long as the single-wavefront panel of the next block that it hides'}
    traffic_from_profile(roof, f'{PROFILE_TAG}_gf2_traffic.json', ['gf2.hip'], {'workload': 'gf2', 'rows': 4000, 'cols': 54000})
    out = contract_line(args, world, world * nx.value * args.steps / dt, 'row-XORs/s', 'gf2_row_xors_per_sec', dt, 'u64',
                        {'workload': 'symmetry_generators_gf2', 'n_qubits': 2000, 'terms': 50000, 'matrix': [4000, 54000], 'row_xors_per_step': nx.value,
                         'generators_found': k.value, 'call': 'IndependentOp.symmetry_generators (symgpu_symmetry_kernel_dev)',
                         'parallelism': f'{world} independent replicas' if world > 1 else 'single GPU'}, roof, comm)
    if rank == 0 and not getattr(args, 'no_api', False):
        from symmer_amd import PauliwordOp, IndependentOp

        def make():
            return PauliwordOp(H.download_bool(2000), H.download_coeff())
        out['api'] = guarded(lambda: api_block('IndependentOp.symmetry_generators(H, commuting_override=True)', make,
                                               lambda Ha: IndependentOp.symmetry_generators(Ha, commuting_override=True),
                                               lambda G: (G.symp_matrix, G.coeff_vec), 5, dt / args.steps))
    H.free()
    if rank == 0 and not args.no_cpu:
        out['cpu_baseline'] = guarded(lambda: cpu_gf2(getattr(args, 'cpu_full', False)))
    return out


def guarded(fn):
    try:
        return fn()
    except Exception as exc:                                      # noqa: BLE001 - reported in the JSON line
        return {'error': f'{type(exc).__name__}: {exc}'}


# ---- `api`: the reference's own call spelling (SURVEY 8d "API call timed per config") on the drop-in classes -----------------------
def host_operator(n_terms, n_qubits, seed):
    """A caller's operator as the reference holds it: NumPy arrays in the reference layout (bool [T, 2n], complex128 [T]) handed to the
    constructor.  The bits come from the library's device generator and are brought to the host (np.random.choice, which
    PauliwordOp.random uses, needs 40 s for cfg2's 2e8 bits); nothing of the operator is on the device when this returns."""
    from symmer_amd import PauliwordOp
    from symmer_amd.kernels import DeviceOp
    d = DeviceOp.random(n_terms, n_qubits, 0.3, seed=seed)
    symp, coeff = d.download_bool(n_qubits), d.download_coeff()
    d.free()
    return PauliwordOp(symp, coeff)


def api_block(spelling, make, call, fetch, reps, c_abi_seconds, note=None):
    """first_call: fresh host operands -> result object (includes their one-time upload; operands of 1 MiB and more go up in the
    reference layout and are packed by a device kernel).  (a) steady state, operands resident (they stay resident behind the drop-in
    classes): host objects in -> result object out, the result left on the device.  (b) = (a) + the result's host arrays read."""
    from symmer_amd import kernels
    pc = time.perf_counter
    obj = make()
    t0 = pc(); r = call(obj); kernels.sync(); first = pc() - t0
    del r
    ta, tb = [], []
    for _ in range(reps):                                             # (a) and (b) in loops of their own: a 40 MB read-back between two 30 us
        t0 = pc(); r = call(obj); kernels.sync(); ta.append(pc() - t0)  # calls leaves the GPU idle for milliseconds, and the next call pays for it
        del r
    for _ in range(reps):
        t0 = pc(); r = call(obj); fetch(r); tb.append(pc() - t0)
        del r
    med = lambda v: sorted(v)[len(v) // 2]
    out = {'call': spelling, 'first_call_seconds': first, 'a_result_object_seconds': med(ta), 'b_host_arrays_out_seconds': med(tb),
           'c_abi_seconds': c_abi_seconds, 'a_over_c_abi': med(ta) / c_abi_seconds if c_abi_seconds else None, 'reps': reps}   # (legend: top-level `api_legend`)
    if note:
        out['note'] = note
    return out


def cpu_mul_cleanup(n):
    """cfg3 cannot be materialised by the reference algorithm (200 GB of temporaries): 1e3 x 1e3 terms on 1,000 qubits."""
    from oracle import oracle_np as onp
    rng = np.random.default_rng(1234)
    A = rng.random((1000, 2 * n)) < 0.3; a = rng.standard_normal(1000) + 1j * rng.standard_normal(1000)
    t0 = time.perf_counter(); r, _ = onp.multiply_by_operator(A, a, A, a); t = time.perf_counter() - t0
    return {'value': 1e6 / t, 'unit': 'pairs/s', 'cores': 1, 'kind': 'port', 'cpu_model': cpu_model(), 'host_cores': os.cpu_count(), 'seconds': t,
            'sample': f'P * P of 1,000 terms on {n} qubits (1e6 pairs -> {r.shape[0]} terms), NumPy restatement of base.py:764-794 + utils.py:230-279, '
                      'single thread (the port\'s first-occurrence unique is a NumPy sort of 2,000-byte rows; the reference uses qiskit\'s Rust hash map)'}


def cpu_rotation(n, N):
    """cfg2 at full size: one non-Clifford rotation of 1e5 terms on 1,000 qubits."""
    from oracle import oracle_np as onp
    rng = np.random.default_rng(1234)
    P2 = rng.random((N, 2 * n)) < 0.3; c2 = rng.standard_normal(N) + 0j; q2 = rng.random(2 * n) < 0.3
    t0 = time.perf_counter(); onp.rotate_by_single_pword(P2, c2, q2, 0.3); t = time.perf_counter() - t0
    return {'value': N / t, 'unit': 'pairs/s', 'cores': 1, 'kind': 'port', 'cpu_model': cpu_model(), 'host_cores': os.cpu_count(), 'seconds': t,
            'sample': f'one rotation of {N} terms on {n} qubits at full size, NumPy restatement of base.py:1090-1161 (commutes -> split -> multiply -> cleanups)'}


def cpu_gf2(full=False):
    """cfg4: the row-XOR rate of the reference's loop depends on the row length and the row count, so SURVEY 8d asks for the full
    4000 x 54000 matrix (72 s on the box's EPYC 9575F).  The default leg is a bounded sample with the full row LENGTH and cites the
    full-size run of tools/cpu_cfg4_full.py cached in profiles/r03_cpu_cfg4_full.json (labelled with host and date); `--cpu-full`
    times the full matrix in this run instead."""
    from oracle import oracle_np as onp
    rng = np.random.default_rng(1234)
    if full:
        M4 = rng.random((4000, 54000)) < 0.5
        t0 = time.perf_counter(); _, nx = onp.rref_noswap(M4, count_xors=True); t = time.perf_counter() - t0
        return {'value': nx / t, 'unit': 'row-XORs/s', 'cores': 1, 'kind': 'port', 'cpu_model': cpu_model(), 'host_cores': os.cpu_count(), 'seconds': t,
                'sample': f'_rref_binary of the full 4000 x 54000 matrix ({int(nx)} row-XORs), NumPy restatement of utils.py:292-315, timed in this run'}
    M4 = rng.random((384, 54000)) < 0.5
    t0 = time.perf_counter(); _, nx = onp.rref_noswap(M4, count_xors=True); t = time.perf_counter() - t0
    out = {'value': nx / t, 'unit': 'row-XORs/s', 'cores': 1, 'kind': 'port', 'cpu_model': cpu_model(), 'host_cores': os.cpu_count(), 'seconds': t,
           'sample': f'_rref_binary of a 384 x 54000 matrix ({int(nx)} row-XORs of cfg4\'s row length), NumPy restatement of utils.py:292-315'}
    try:
        with open(os.path.join(ROOT, 'profiles', 'r03_cpu_cfg4_full.json')) as f:
            out['full_size_cached'] = json.load(f)
    except (OSError, ValueError):
        out['full_size_cached'] = None
    return out


# ------------------------------------------------------------------------------------------------------------------
def adjacency(args, comm, rank, world, _lib, DeviceOp, parallel):
    """BASELINE cfg5: adjacency matrix (commutes_termwise with itself) of ONE T-term operator, rows sharded over the ranks.
    Every rank owns T/world consecutive terms: they are its block of the left axis AND its shard of the right operand, so the
    all-gather that assembles the right operand is also the distribution step.  Step = all-gather + this rank's
    [T/world, T] block of np.bool_ results, written in 25,000-row slabs into a ring of two device buffers.  Strong scaling."""
    lib = _lib.lib()
    n, T = args.adj_qubits, args.adj_terms
    wq = (n + 63) // 64
    ts, bounds = parallel.shard_bounds(T, world)
    b0, b1 = bounds[rank]
    shard = parallel.padded_random_shard(b1 - b0, ts, n, 555 + rank)
    full = DeviceOp.alloc(ts * world, wq, with_coeff=True) if comm.gathers else shard
    # Output slabs: as many rows per launch as a quarter of the free HBM holds (a 25,000-row launch is 6.5 rounds of workgroups on
    # 256 CUs and costs 7; all 200,000 rows of one GPU in one launch: 50.1 rounds) — the whole 40 GB block of a single GPU fits.
    free_b, total_b = ctypes.c_int64(0), ctypes.c_int64(0)
    _lib.check(lib.symgpu_mem_info(ctypes.addressof(free_b), ctypes.addressof(total_b)))
    slab = min(max(1, b1 - b0), max(25000, int(free_b.value // 4 // max(1, T))))
    if args.adj_slab_rows:
        slab = min(max(1, b1 - b0), args.adj_slab_rows)
    ring = []
    for _ in range(1 if slab >= b1 - b0 else 2):
        p = ctypes.c_void_p()
        _lib.check(lib.symgpu_dev_alloc(slab * T, ctypes.byref(p)))
        ring.append(p)

    if comm.gathers:
        comm.allgather_op(shard, full, T)
        comm.verify_allgather(shard, full, T)       # bit-for-bit against a host-staged gather, outside the timed region

    def step():
        if comm.gathers:
            comm.allgather_op(shard, full, T)
        k = 0
        for r0 in range(0, b1 - b0, slab):
            r1 = min(b1 - b0, r0 + slab)
            _lib.check(lib.symgpu_commutes_dev(shard.handle, r0, r1, full.handle, ring[k % len(ring)]))
            k += 1

    for _ in range(args.warmup):
        step()
    _lib.check(lib.symgpu_device_sync()); comm.barrier()
    _lib.check(lib.symgpu_prof_enable(1, 1))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    _lib.check(lib.symgpu_device_sync()); comm.barrier()
    dt = comm.max_over_ranks(time.perf_counter() - t0)
    _lib.check(lib.symgpu_prof_enable(1, 0))
    nl, ms = ctypes.c_int64(0), ctypes.c_double(0)
    _lib.check(lib.symgpu_prof_read(1, ctypes.addressof(nl), ctypes.addressof(ms)))
    kt = ms.value / max(1, nl.value) * 1e-3
    pairs = T * T
    n_kblocks = m4r_groups(n)
    launch_rows = (b1 - b0) * args.steps / max(1, nl.value)
    lds_bytes = launch_rows * n_kblocks * (((T + 63) // 64 + 31) // 32) * 256.0
    out = {'metric': 'pauli_term_pairs_per_sec', 'value': pairs * args.steps / dt, 'unit': 'pairs/s', 'n_gpus': world, 'steps': args.steps,
           'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
           'dtype': 'u64', 'data': 'synthetic',
           'config': {'workload': 'commutes_termwise_adjacency', 'n_qubits': n, 'terms': T, 'rows_per_gpu': b1 - b0, 'pairs_per_step': pairs,
                      'bytes_per_pair': 1, 'slab_rows': slab,
                      'parallelism': (f'left-axis shard x{world}, all-gather of right rows ({comm.data_plane})' if world > 1 else 'single GPU')},
           # the contract's roofline object is HBM-side (1 B/pair np.bool_ output); the kernel itself is bound by the LDS table
           # reads of the Four-Russians product, reported next to it
           'roofline': {'bound': 'hbm', 'kernel': 'k_commutes_m4r7s', 'achieved': launch_rows * T / kt / 1e9 if kt else None, 'peak': HBM_PEAK_GBS,
                        'unit': 'GB/s', 'frac': launch_rows * T / kt / 1e9 / HBM_PEAK_GBS if kt else None, 'traffic': None,
                        'launches': nl.value, 'avg_launch_ms': kt * 1e3,
                        'note': 'not HBM-bound: Four-Russians GF(2) product, one 256-byte LDS table entry per row, 7-bit group and 2048-column tile; two '
                                'tables per step folded with v_bitop3, persistent workgroups over the (tile, step) space (csrc/commute_m4r7.hip); '
                                'the events cover the main launch and the (usually empty) fix-up launch behind it',
                        'lds': {'achieved_GBps': lds_bytes / kt / 1e9 if kt else None, 'peak_GBps': 157286.4,
                                'frac': lds_bytes / kt / 1e9 / 157286.4 if kt else None,
                                'peak_source': '256 CUs x 256 B/clk (ds_read_b128) x 2.4 GHz, MI355X_MICROARCH.md LDS table'}}}
    if comm.degraded:
        out['degraded'] = comm.degraded
    for p in ring:
        _lib.check(lib.symgpu_dev_free(p))
    traffic_from_profile(out['roofline'], f'{PROFILE_TAG}_adjacency_traffic.json', ['commute_m4r.hip', 'commute_m4r7.hip'], {'workload': 'adjacency', 'n_qubits': n, 'terms': T})
    if rank == 0 and world == 1 and not getattr(args, 'no_api', False):
        out['api'] = guarded(lambda: api_adjacency(n, T, dt / args.steps))
    if rank == 0 and not args.no_cpu:
        out['cpu_baseline'] = guarded(lambda: cpu_adjacency(n))
    return out


def m4r_groups(n):
    """Non-zero 7-bit groups of a packed n-qubit row (X bits 0 .. n-1, Z bits 64 Wq .. 64 Wq + n-1): the table look-ups of one row and one
    2048-column tile in csrc/commute_m4r7.hip, padded to whole pairs (a step reads two tables)."""
    wq = (n + 63) // 64
    groups = {b // 7 for b in range(n)} | {(64 * wq + b) // 7 for b in range(n)}
    return len(groups) + (len(groups) & 1)


def api_adjacency(n, T, c_abi_seconds):
    """`P.commutes_termwise(P)` (base.py:938-971) returns a NumPy bool [T, T]: 40 GB of host memory at cfg5, written by one D2H copy
    behind the 40 ms kernel.  Timed at full size when the host has the memory for it, and always on one rank's 25,000-row share."""
    avail = 0
    try:
        with open('/proc/meminfo') as f:
            for line in f:
                if line.startswith('MemAvailable'):
                    avail = int(line.split()[1]) * 1024
    except OSError:
        pass
    P = host_operator(T, n, 555)
    res = {}
    rows = min(25000, T)
    Ps = P[:rows]
    res['rank_share'] = api_block(f'Ps = P[:{rows}]; Ps.commutes_termwise(P)', lambda: (Ps, P), lambda ops: ops[0].commutes_termwise(ops[1]), lambda C: C, 3,
                                  c_abi_seconds * rows / T, note='the result IS a host array: a = b; c_abi scaled to the share')
    if avail > 3 * T * T:
        res['full'] = api_block('P.commutes_termwise(P)', lambda: P, lambda Pa: Pa.commutes_termwise(Pa), lambda C: C, 2, c_abi_seconds,
                                note=f'{T * T / 1e9:.0f} GB np.bool_ result on the host (MemAvailable {avail / 1e9:.0f} GB): a = b')
    else:
        res['full'] = {'skipped': f'host MemAvailable {avail / 1e9:.0f} GB < 3 x the {T * T / 1e9:.0f} GB result'}
    return res


def cpu_adjacency(n):
    """cfg5 cannot be materialised by the reference algorithm (320 GB of float64 temporaries): a 2,000 x 20,000 block of the 2,000-qubit
    commutation table, float64 dot % 2 as the reference (utils.py:63-78)."""
    from oracle import oracle_np as onp
    rng = np.random.default_rng(1234)
    C5 = rng.random((20000, 2 * n)) < 0.3
    t0 = time.perf_counter(); onp.commutes_termwise(C5[:2000], C5); t = time.perf_counter() - t0
    return {'value': 4e7 / t, 'unit': 'pairs/s', 'cores': 'BLAS default threads', 'kind': 'port', 'cpu_model': cpu_model(), 'host_cores': os.cpu_count(), 'seconds': t,
            'sample': f'P[:2000].commutes_termwise(P) of a 20,000-term, {n}-qubit operator (4e7 pairs), NumPy restatement of base.py:938-971 (float64 dot % 2)'}


def timed(fn, reps):
    from symmer_amd import kernels
    fn()
    kernels.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    kernels.sync()
    return (time.perf_counter() - t0) / reps


def extras(_lib, kernels, DeviceOp, comm, parallel, args, headline):
    """The other BASELINE.json configs on the same GPU.  cfg2 / cfg3 / cfg4 / cfg5 run the SAME code as their `--workload` lines
    (wl_rotation, wl_mul_cleanup, wl_gf2, adjacency) with a short timed region, and carry that line's `roofline` object (dominant
    kernel, HIP-event duration, algorithmic or physical bytes, fraction of peak; never above 1: bytes that do not move are not
    counted).  Every section runs on its own: one that fails reports its error and the others still run."""
    import types
    lib = _lib.lib()
    from symmer_amd.operators import PauliwordOp, check_independent
    from symmer_amd import packing
    ex = {}

    def section(fn):
        try:
            fn()
        except Exception as exc:                                  # noqa: BLE001 - reported in the JSON line
            ex[fn.__name__] = {'error': f'{type(exc).__name__}: {exc}'}

    def workload_line(fn, steps, warmup=1, **kw):
        a = types.SimpleNamespace(steps=steps, warmup=warmup, qubits=1000, no_cpu=True, no_extras=True, adj_terms=200000, adj_qubits=2000,
                                  adj_slab_rows=0, gpus=1, no_api=getattr(args, 'no_api', False), cpu_full=False)
        for k, v in kw.items():
            setattr(a, k, v)
        return fn(a, comm, 0, 1, _lib, DeviceOp, parallel)

    def wake_gpu():
        # the CPU baseline leg leaves the GPU idle for tens of seconds: its clocks are down when the (latency bound, sub-millisecond)
        # small cases below start — 50 ms of row-stream work first
        Aw = DeviceOp.random(20000, 1000, 0.3, seed=5); Bw = DeviceOp.random(256, 1000, 0.3, seed=6)
        ow = DeviceOp.alloc(256 * 20000, 16, with_coeff=True)
        for _ in range(250):
            _lib.check(lib.symgpu_mul_allpairs_dev(Aw.handle, Bw.handle, 0, 256, 1, ow.handle))
        kernels.sync()
        ow.free(); Aw.free(); Bw.free()
    section(wake_gpu)

    def strong_scaling_shard():
        # the per-rank shape of the 8-GPU STRONG-scaling run of the headline problem (1e5 x 1e5 terms in total): 12,500 left terms against all
        # 1e5 right terms, 2,048 outer rows per output slab so that a launch writes the same 6.5 GB as the 256-row slab of the full left operand
        n, Ni, M, slab = 1000, 12500, 100000, 2048
        left = DeviceOp.random(Ni, n, 0.3, seed=4321); right = DeviceOp.random(M, n, 0.3, seed=99991)
        ring = [DeviceOp.alloc(slab * Ni, 16, with_coeff=True) for _ in range(2)]

        def step():
            for k, o0 in enumerate(range(0, M, slab)):
                _lib.check(lib.symgpu_mul_allpairs_dev(left.handle, right.handle, o0, min(M, o0 + slab), 1, ring[k & 1].handle))
        step(); kernels.sync()
        _lib.check(lib.symgpu_prof_enable(0, 1))
        t0 = time.perf_counter()
        for _ in range(3):
            step()
        kernels.sync()
        dt = (time.perf_counter() - t0) / 3
        _lib.check(lib.symgpu_prof_enable(0, 0))
        nl, ms = prof_read(_lib, 0)
        kt = ms / max(1, nl) * 1e-3
        launch_bytes = Ni * M * 3 / max(1, nl) * 256
        full_ms = headline.get('ms_per_step')
        ex['strong_scaling_shard'] = {
            'left_terms': Ni, 'right_terms': M, 'slab_rows': slab, 'pairs_per_step': Ni * M, 'ms_per_step': dt * 1e3, 'pairs_per_s': Ni * M / dt,
            'roofline': {'bound': 'hbm', 'kernel': headline['roofline']['kernel'], 'achieved': launch_bytes / kt / 1e9 if kt else None, 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': launch_bytes / kt / 1e9 / HBM_PEAK_GBS if kt else None, 'traffic': None, 'launches': nl,
                         'avg_launch_ms': kt * 1e3, 'algorithmic_bytes_per_launch': launch_bytes},
            'predicted_8gpu_strong_speedup_before_allgather': (full_ms / (dt * 1e3)) if full_ms else None,
            'note': 'one rank\'s share of the fixed 1e5 x 1e5 product on 8 GPUs, timed on one GPU; the all-gather of the 27 MB right operand '
                    '(~0.1-0.3 ms over xGMI) comes on top of ms_per_step'}
        for h in ring + [left, right]:
            h.free()

    def cfg1_api_mul():
        # the reference's own CPU-runnable case through the drop-in API, host buffers in / host result out
        rng1 = np.random.default_rng(1235)
        P1 = PauliwordOp(rng1.random((500, 200)) < 0.3, rng1.standard_normal(500) + 1j * rng1.standard_normal(500))
        symp1, c1 = P1.symp_matrix, P1.coeff_vec.copy()
        for _ in range(4):                                          # the first calls of a process load the code objects of a dozen kernels
            (PauliwordOp(symp1, c1.copy()) * PauliwordOp(symp1, c1.copy()))
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            Pf = PauliwordOp(symp1, c1.copy())                      # a fresh operand every time: pack + upload + product/cleanup + download
            R1 = Pf * Pf
            rows_out, coeff_out = R1.symp_matrix, R1.coeff_vec
            ts.append(time.perf_counter() - t0)
        t = sorted(ts)[2]                                            # median of five: a sub-millisecond call through Python catches the odd GC pause
        ex['cfg1_api_mul'] = {'call': 'P = PauliwordOp(symp, coeff); R = P * P; R.symp_matrix, R.coeff_vec (fresh host arrays in, host arrays out; median of 5)', 'pairs': 250000,
                              'seconds': t, 'pairs_per_s': 250000 / t, 'terms_out': R1.n_terms,
                              'api': api_block('P * P', lambda: PauliwordOp(symp1, c1.copy()), lambda P: P * P, lambda R: (R.symp_matrix, R.coeff_vec), 5, None)}

    def cfg3_mul_cleanup():
        # 1,000 qubits, 10,000 terms squared (1e8 pairs) + cleanup: the `--workload mul_cleanup` line, 20 steps
        line = workload_line(wl_mul_cleanup, 20, 3)
        ex['cfg3_mul_cleanup'] = {'pairs': 10**8, 'seconds': line['ms_per_step'] * 1e-3, 'pairs_per_s': line['value'], 'terms_out': line['config']['terms_out'],
                                  'roofline': line['roofline'], 'api': line.get('api'),
                                  'note': 'squared operator: keys for the pairs with i >= o only (cleanup.hip); roofline = the output stage, the one kernel of the '
                                          'step that moves the result\'s bytes (SURVEY 8d\'s T (16 Wq + 16) bytes of product rows never exist here)'}

    def cfg2_rotation():
        # 1,000 qubits, 100,000 terms: the `--workload rotation` line (100 non-Clifford rotations per step) + the per-call time seen by the host
        line = workload_line(wl_rotation, 1, no_extras=False)
        rng = np.random.default_rng(1236)
        P = DeviceOp.random(100000, 1000, 0.3, seed=1236)
        qs = [packing.pack_rows((rng.random((1, 2000)) < 0.3))[0] for _ in range(8)]

        def chain4():
            terms, cur = [], P
            for q in qs[:4]:
                res, allc = kernels.rotate_single_dev(cur, q, 0.3)
                if cur is not P:
                    cur.free()
                cur = res
                terms.append(cur.n_terms)
            kernels.sync()
            cur.free()
            return terms
        chain4(); chain4()                                           # one-time costs: kernel modules, hash tables, the join tables of both paths
        t0 = time.perf_counter(); terms = chain4(); t_chain = time.perf_counter() - t0
        t1 = timed(lambda: kernels.rotate_single_dev(P, qs[0], 0.3)[0].free(), 20)
        P.free()
        ex['cfg2_rotation'] = {'terms_in': 100000, 'seconds_per_rotation': line['seconds_per_rotation'], 'term_pairs_per_s': line['value'],
                               'first_rotation_seconds': t1, 'first_rotation_note': 'through kernels.rotate_single_dev (Python wrapper + handle free per call), 20 calls back to back',
                               'roofline': line['roofline'], 'api': line.get('api'), 'api_perform_rotations': line.get('api_perform_rotations'), 'chain4_seconds': t_chain, 'chain_terms': terms, 'clifford': line.get('clifford'),
                               'saturated_chain': line.get('saturated_chain')}
        # README claim 1 (a depth-2,000 Clifford circuit on 1,000 qubits "in one second"): 2,000 Clifford rotations of a 64-term,
        # 1,000-qubit observable through perform_rotations — one single-workgroup launch for the whole run (rotate.hip)
        rng_c = np.random.default_rng(1240)
        obs = PauliwordOp(rng_c.random((64, 2000)) < 0.3, rng_c.standard_normal(64) + 0j).cleanup()
        rots = [(PauliwordOp(rng_c.random((1, 2000)) < 0.02, [1]), float(rng_c.integers(1, 4)) * np.pi / 2) for _ in range(2000)]
        obs.perform_rotations(rots[:20])
        t0 = time.perf_counter(); rot_obs = obs.perform_rotations(rots); t_circ = time.perf_counter() - t0
        ex['cfg2_rotation']['clifford_circuit_2000_rotations_64_terms'] = {
            'seconds': t_circ, 'seconds_per_rotation': t_circ / 2000, 'terms_out': rot_obs.n_terms,
            'call': 'PauliwordOp.perform_rotations (Python API: upload, one chain launch, download)'}

    def cfg5_adjacency():
        # 2,000 qubits, 200,000 terms: the whole adjacency matrix on this GPU (`--workload adjacency`, 1 step) and ONE rank's share of the
        # 8-GPU run, a 25,000 x 200,000 block, timed on its own
        line = workload_line(adjacency, 1)
        C = DeviceOp.random(200000, 2000, 0.3, seed=1239)
        nrow = 25000
        buf = ctypes.c_void_p()
        _lib.check(lib.symgpu_dev_alloc(nrow * 200000, ctypes.byref(buf)))
        # a 4 ms call is timed at the clocks of a busy GPU, as inside the 8-GPU step: the API legs above left the chip idle for seconds (the
        # first calls after that run 10 % slower, `tools/bench_adj_rows.py`), so ~60 ms of the same launches first, then the median of 10
        for _ in range(14):
            _lib.check(lib.symgpu_commutes_dev(C.handle, 0, nrow, C.handle, buf))
        kernels.sync()
        _lib.check(lib.symgpu_prof_enable(1, 1))
        ts = []
        for _ in range(10):
            t0 = time.perf_counter(); _lib.check(lib.symgpu_commutes_dev(C.handle, 0, nrow, C.handle, buf)); kernels.sync(); ts.append(time.perf_counter() - t0)
        t = sorted(ts)[len(ts) // 2]
        _lib.check(lib.symgpu_prof_enable(1, 0))
        nl, ms = prof_read(_lib, 1)
        kt = ms / max(1, nl) * 1e-3
        pairs = nrow * 200000
        n_kblocks = m4r_groups(2000)                                   # non-zero 7-bit groups of a 2,000-qubit row (X and Z halves)
        col_tiles = (200000 // 64 + 31) // 32
        lds_bytes = nrow * n_kblocks * col_tiles * 256.0
        full_s = line['ms_per_step'] * 1e-3
        ex['cfg5_adjacency'] = {'pairs': 200000 ** 2, 'seconds': full_s, 'pairs_per_s': line['value'], 'roofline': line['roofline'], 'api': line.get('api'),
                                'rank_share_25000_rows': {'pairs': pairs, 'seconds': t, 'pairs_per_s': pairs / t, 'kernel': 'k_commutes_m4r7s', 'kernel_seconds': kt,
                                                          'hbm_GBps_at_1B_per_pair': pairs / kt / 1e9 if kt else None,
                                                          'lds_read_frac_of_157TBps': lds_bytes / kt / 157.3e12 if kt else None,
                                                          'predicted_8gpu_strong_speedup_before_allgather': full_s / t}}
        _lib.check(lib.symgpu_dev_free(buf)); C.free()

    def cfg4_symmetry_kernel():
        # GF(2) symmetry kernel, 2,000 qubits x 50,000 terms, 32 planted symmetries, Clifford-scrambled: the `--workload gf2` line, 10 steps
        line = workload_line(wl_gf2, 10, 2)
        ex['cfg4_symmetry_kernel'] = {'rows': 4000, 'cols': 54000, 'generators_found': line['config']['generators_found'], 'row_xors': line['config']['row_xors_per_step'],
                                      'seconds': line['ms_per_step'] * 1e-3, 'row_xors_per_s': line['value'], 'roofline': line['roofline'], 'api': line.get('api')}

    def f2_generator_reconstruction():
        # SURVEY 8f row f2 at cfg4's operator size: 50,000 terms on 2,000 qubits reconstructed in 60 independent generators
        # (PauliwordOp.generator_reconstruction, base.py:523-560) and the generators of a 20,000-term operator (base.py:1436-1456)
        n2, T2, g2 = 2000, 50000, 60
        Mop = host_operator(T2, n2, 4004)
        Gop = host_operator(g2, n2, 4005)
        Mop.generator_reconstruction(Gop)
        t_rec = timed(lambda: Mop.generator_reconstruction(Gop), 3)
        t_ci = timed(lambda: check_independent(Gop), 5)
        sub = Mop[:20000]
        sub._device()
        t0 = time.perf_counter(); gens = PauliwordOp._from_device(kernels.generators_dev(sub._device()), n2); kernels.sync(); t_gen = time.perf_counter() - t0
        ex['f2_generator_reconstruction'] = {'terms': T2, 'n_qubits': n2, 'generators': g2, 'matrix': [2 * n2, g2 + T2],
                                             'seconds_generator_reconstruction': t_rec, 'seconds_check_independent': t_ci,
                                             'generators_of_20000_terms': {'seconds': t_gen, 'rank': gens.n_terms},
                                             'call': 'M.generator_reconstruction(G) -> (int64 [50000, 60], bool [50000]) on the host; operands resident, '
                                                     'transposed stack built / reduced / read out on the device (csrc/genrec.hip)'}

    def readme_claim1_clifford_circuit():
        # reference README.md:50-51: "expectation value of a 1,000-qubit Clifford circuit of depth 2,000" — CircuitSymmerlator: 2,000
        # random H / S / CX gates (4,000 single-Pauli rotations), a 64-term observable, evaluate() = <0| U^+ O U |0>
        from symmer_amd.evolution import CircuitSymmerlator
        rng1c = np.random.default_rng(1242)
        nq = 1000
        obs = PauliwordOp(rng1c.random((64, 2 * nq)) < 0.3, rng1c.standard_normal(64) + 0j)

        def build():
            C = CircuitSymmerlator(nq)
            for _ in range(2000):
                g = rng1c.integers(0, 3)
                if g == 0:
                    C.H(int(rng1c.integers(0, nq)))
                elif g == 1:
                    C.S(int(rng1c.integers(0, nq)))
                else:
                    a, b = rng1c.choice(nq, 2, replace=False)
                    C.CX(int(a), int(b))
            return C
        build().evaluate(obs)
        t0 = time.perf_counter(); C = build(); t_build = time.perf_counter() - t0
        t0 = time.perf_counter(); val = C.evaluate(obs); t_eval = time.perf_counter() - t0
        ex['readme_claim1_clifford_circuit'] = {'n_qubits': nq, 'gates': 2000, 'rotations': len(C.sequence), 'observable_terms': 64,
                                                'seconds_build_circuit': t_build, 'seconds_evaluate': t_eval, 'expectation': [float(np.real(val)), float(np.imag(val))],
                                                'call': 'CircuitSymmerlator.evaluate (Python API)'}

    def readme_claim3_square_1000q_500t():
        # reference README.md:53: "square a 1,000-qubit, 500-term operator incl. cleanup over 250,000 cross terms" (one second on a laptop;
        # 8.3 s for the reference code in the survey container) — through the drop-in API, host arrays in and out
        rng3 = np.random.default_rng(1241)
        P3 = PauliwordOp(rng3.random((500, 2000)) < 0.3, rng3.standard_normal(500) + 1j * rng3.standard_normal(500))
        symp3, c3 = P3.symp_matrix, P3.coeff_vec.copy()
        (P3 * P3)
        t0 = time.perf_counter()
        for _ in range(5):
            Pf = PauliwordOp(symp3, c3.copy())
            R3 = Pf * Pf
            rows_out, coeff_out = R3.packed, R3.coeff_vec
        t = (time.perf_counter() - t0) / 5
        ex['readme_claim3_square_1000q_500t'] = {'pairs': 250000, 'seconds': t, 'pairs_per_s': 250000 / t, 'terms_out': R3.n_terms,
                                                 'call': 'P = PauliwordOp(symp, coeff); R = P * P; R.packed, R.coeff_vec (fresh host arrays in, packed host arrays out)'}

    def readme_claim4_wide_product():
        # reference README.md:54: "multiply two 100,000,000-qubit Pauli terms" (in one second on a laptop).  Through the drop-in API:
        # host bool arrays in (2 x 2e8 bytes), packed, uploaded, fused product + cleanup on the word-parallel kernels (wide.hip),
        # one packed row back.
        nq = 100_000_000
        rngw = np.random.default_rng(1240)
        bits = lambda: np.unpackbits(rngw.integers(0, 256, 2 * nq // 8, dtype=np.uint8)).astype(bool).reshape(1, -1)
        A = PauliwordOp(bits(), [1.0]); B = PauliwordOp(bits(), [1.0])
        (A * B)
        A2 = PauliwordOp(A.symp_matrix, [1.0]); B2 = PauliwordOp(B.symp_matrix, [1.0])
        t0 = time.perf_counter(); R = A2 * B2; row_out = R.packed; t_all = time.perf_counter() - t0
        t0 = time.perf_counter(); R = A2 * B2; kernels.sync(); t_dev = time.perf_counter() - t0
        ex['readme_claim4_wide_product'] = {'n_qubits': nq, 'terms': '1 x 1', 'seconds_host_bool_arrays_in': t_all,
                                            'seconds_operands_resident': t_dev, 'terms_out': R.n_terms,
                                            'call': 'PauliwordOp * PauliwordOp (Python API)'}

    for fn in (strong_scaling_shard, cfg1_api_mul, cfg3_mul_cleanup, cfg2_rotation, cfg5_adjacency, cfg4_symmetry_kernel, f2_generator_reconstruction, readme_claim1_clifford_circuit, readme_claim3_square_1000q_500t, readme_claim4_wide_product):
        section(fn)
    return ex


def cpu_baseline(n):
    """NumPy port of the reference's product (no cleanup), bounded sample of the same 1,000-qubit workload."""
    from oracle import oracle_np as onp
    rng = np.random.default_rng(1234)
    Ns, Ms = 1000, 250
    A = rng.random((Ns, 2 * n)) < 0.3; B = rng.random((Ms, 2 * n)) < 0.3
    a = rng.standard_normal(Ns) + 1j * rng.standard_normal(Ns); b = rng.standard_normal(Ms) + 1j * rng.standard_normal(Ms)
    onp.product_rows_and_coeffs(A[:50], a[:50], B[:20], b[:20])
    reps, t_total = 0, 0.0
    while t_total < 10.0 and reps < 100:
        t0 = time.perf_counter()
        onp.product_rows_and_coeffs(A, a, B, b)
        t_total += time.perf_counter() - t0
        reps += 1
    v = Ns * Ms * reps / t_total
    other = {}
    # cfg1 at full size: product + cleanup of 500 x 500 terms on 100 qubits
    A1 = rng.random((500, 200)) < 0.3; a1 = rng.standard_normal(500) + 1j * rng.standard_normal(500)
    t0 = time.perf_counter(); onp.multiply_by_operator(A1, a1, A1, a1); t = time.perf_counter() - t0
    other['cfg1_mul_cleanup'] = {'pairs': 250000, 'seconds': t, 'pairs_per_s': 250000 / t}
    # cfg3 cannot be materialised by the reference algorithm (200 GB temporaries): 500 x 500 terms on 1,000 qubits
    # (the port's first-occurrence unique is a NumPy sort of 2,000-byte rows; the reference uses qiskit's Rust hash map there)
    t0 = time.perf_counter(); onp.multiply_by_operator(A[:500], a[:500], A[:500], a[:500]); t = time.perf_counter() - t0
    other['cfg3_sample_mul_cleanup'] = {'pairs': 250000, 'seconds': t, 'pairs_per_s': 250000 / t}
    # cfg2 at full size: one non-Clifford rotation of 1e5 terms on 1,000 qubits
    P2 = rng.random((100000, 2 * n)) < 0.3; c2 = rng.standard_normal(100000) + 0j; q2 = rng.random(2 * n) < 0.3
    t0 = time.perf_counter(); onp.rotate_by_single_pword(P2, c2, q2, 0.3); t = time.perf_counter() - t0
    other['cfg2_rotation'] = {'terms_in': 100000, 'seconds': t, 'term_pairs_per_s': 1e5 / t}
    del P2
    # cfg5 sample: 1,000 x 10,000 block of the 2,000-qubit commutation matrix (f64 dot % 2 as the reference)
    C5 = rng.random((10000, 4000)) < 0.3
    t0 = time.perf_counter(); onp.commutes_termwise(C5[:1000], C5); t = time.perf_counter() - t0
    other['cfg5_sample_commutation'] = {'pairs': 10**7, 'seconds': t, 'pairs_per_s': 1e7 / t}
    del C5
    # cfg4 sample: row-XOR rate depends on the row length, so keep cfg4's 54,000 columns and reduce the row count
    M4 = rng.random((256, 54000)) < 0.5
    t0 = time.perf_counter(); _, nx = onp.rref_noswap(M4, count_xors=True); t = time.perf_counter() - t0
    other['cfg4_sample_rref'] = {'rows': 256, 'cols': 54000, 'row_xors': int(nx), 'seconds': t, 'row_xors_per_s': nx / t}
    try:                                                              # the full 4000 x 54000 run of the same loop (~72 s), measured once by tools/cpu_cfg4_full.py
        with open(os.path.join(ROOT, 'profiles', 'r03_cpu_cfg4_full.json')) as f:
            other['cfg4_full_size_cached'] = json.load(f)
    except (OSError, ValueError):
        other['cfg4_full_size_cached'] = None
    model = 'unknown'
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    model = line.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    return {'cpu_model': model, 'host_cores': os.cpu_count(), 'other_configs': other, 'value': v, 'unit': 'pairs/s', 'cores': 1, 'kind': 'port',
            'sample': f'{reps} x ({Ns} x {Ms} terms, {n} qubits) all-pairs product, NumPy restatement of base.py:783-792 '
                      f'(1 byte per bit, single thread; host has {os.cpu_count()} cores)'}


if __name__ == '__main__':
    main()
