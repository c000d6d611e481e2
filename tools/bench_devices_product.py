# ad-hoc timing (not a test), VERDICT r5 item 2: what `DeviceGroup.mul_cleanup` (symmer_amd/multi.py) costs per device, measured on ONE GPU,
# next to the single-device call it would replace.  For P * P of an N-term, 1,000-qubit operator over G devices a device computes the GENERAL
# sub-product (all N inner terms) x (its N / G outer terms) without threshold (the squared-operator half-pairs path is lost), the cleaned
# parts are copied to the home device and their concatenation is cleaned once more there.
#     python tools/bench_devices_product.py [N] [G ...]
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from symmer_amd import kernels, _lib
from symmer_amd.kernels import DeviceOp

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
Gs = [int(a) for a in sys.argv[2:]] or [2, 4, 8]
n = 1000


def timed(fn, reps=3):
    r = fn(); kernels.sync()
    if r is not None:
        r.free()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); kernels.sync(); ts.append(time.perf_counter() - t0)
        if r is not None:
            r.free()
    return sorted(ts)[len(ts) // 2]


P = DeviceOp.random(N, n, 0.3, seed=77)
out = {'terms': N, 'n_qubits': n, 'pairs': N * N}
t_single = timed(lambda: kernels.mul_cleanup_handles(P, P, True, 1e-15))
res = kernels.mul_cleanup_handles(P, P, True, 1e-15)
out['single_device'] = {'seconds': t_single, 'terms_out': res.n_terms, 'call': 'kernels.mul_cleanup_handles(P, P): squared-operator path, tiled above MAX_PAIRS_PER_CALL'}
res.free()
print(json.dumps(out['single_device']), flush=True)
for G in Gs:
    blk = (N + G - 1) // G
    block = kernels.op_gather(P, np.arange(0, blk))
    try:
        t_share = timed(lambda: kernels.mul_cleanup_handles(P, block, True, None))
        part = kernels.mul_cleanup_handles(P, block, True, None)
        rows_part = part.n_terms
        entry = {'G': G, 'share_seconds': t_share, 'share_rows_out': rows_part, 'share_GB': rows_part * 272 / 1e9}
        # the home device's second cleanup: G parts of this size stacked (the parts of a real run differ, their sizes do not), if it fits
        free_b, total_b = kernels.ctypes.c_int64(0), kernels.ctypes.c_int64(0)
        _lib.check(_lib.lib().symgpu_mem_info(kernels.ctypes.addressof(free_b), kernels.ctypes.addressof(total_b)))
        need = G * rows_part * 272 * 2.2
        if need < free_b.value:
            cat = DeviceOp.alloc(G * rows_part, 16, True)
            for g in range(G):
                _lib.check(_lib.lib().symgpu_op_copy_rows(cat.handle, g * rows_part, part.handle, 0, rows_part))
            cat.set_rows(G * rows_part)
            t_copy = timed(lambda: [_lib.check(_lib.lib().symgpu_op_copy_rows(cat.handle, g * rows_part, part.handle, 0, rows_part)) for g in range(G)] and None, 2)
            t_clean = timed(lambda: kernels.cleanup_dev(cat, 1e-15), 2)
            entry.update({'concat_rows': G * rows_part, 'concat_copy_seconds_on_one_device': t_copy, 'home_cleanup_seconds': t_clean,
                          'note': 'G copies of one part: every row G-fold — the merge cost of the same number of rows'})
            cat.free()
        else:
            entry['home_cleanup'] = f'skipped: {need / 1e9:.0f} GB needed, {free_b.value / 1e9:.0f} GB free'
        entry['device_group_seconds_estimate'] = t_share + entry.get('home_cleanup_seconds', 0) + entry.get('concat_copy_seconds_on_one_device', 0)
        part.free()
    except Exception as exc:                                       # noqa: BLE001
        entry = {'G': G, 'error': f'{type(exc).__name__}: {exc}'}
    block.free()
    out[f'G{G}'] = entry
    print(json.dumps(entry), flush=True)
print(json.dumps(out))
