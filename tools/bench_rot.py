# ad-hoc: rotation timing only (for rocprof breakdowns): BASELINE cfg2 (1e5 terms, 1,000 qubits), one rotation per call
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from symmer_amd import kernels, packing
from symmer_amd.kernels import DeviceOp
rng = np.random.default_rng(5)
sizes = [(100000, 1000)] if os.environ.get('ROT_ONLY_CFG2') else [(100000, 1000), (1000, 1000), (20000, 1000), (140000, 1000), (100000, 100), (50000, 2000)]
def timed(fn, reps):
    fn(); kernels.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    kernels.sync(); return (time.perf_counter() - t0) / reps
for T, n in sizes:
    P = DeviceOp.random(T, n, 0.3, seed=1236)
    q = packing.pack_rows((rng.random((1, 2 * n)) < 0.3))[0]
    kernels.rotate_single_dev(P, q, 0.3)[0].free()           # multi-launch: duplicate status + hashes of P
    out = []
    for env in (None, '0'):
        if env is None: os.environ.pop('SYMGPU_ROT_RESIDENT', None)
        else: os.environ['SYMGPU_ROT_RESIDENT'] = env
        t = timed(lambda: kernels.rotate_single_dev(P, q, 0.3)[0].free(), 20)
        t2 = timed(lambda: kernels.rotate_single_dev(P, q, np.pi / 2)[0].free(), 20)
        out.append((t, t2))
    os.environ.pop('SYMGPU_ROT_RESIDENT', None)
    print(f'rotation {T} terms, {n} qubits: non-Clifford {out[0][0]*1e3:.4f} ms, Clifford {out[0][1]*1e3:.4f} ms   (multi-launch paths: {out[1][0]*1e3:.4f} / {out[1][1]*1e3:.4f} ms)', flush=True)
    P.free()

# phase stamps of the one-launch kernel (100 MHz wall clock): per phase, the time at which the LAST workgroup passed it
if os.environ.get('ROT_TRACE'):
    import ctypes
    from symmer_amd import _lib
    P = DeviceOp.random(100000, 1000, 0.3, seed=1236)
    q = packing.pack_rows((rng.random((1, 2000)) < 0.3))[0]
    kernels.rotate_single_dev(P, q, 0.3)[0].free()
    os.environ['SYMGPU_RES_TRACE'] = '1'
    for ang, name in ((0.3, 'non-Clifford'), (np.pi / 2, 'Clifford')):
        for rep in range(3):
            kernels.rotate_single_dev(P, q, ang)[0].free()
        buf = np.zeros((256, 16), dtype=np.uint64); n = ctypes.c_int(0)
        _lib.check(_lib.lib().symgpu_debug_rotation_trace(buf.ctypes.data, 256, ctypes.addressof(n)))
        t = buf[:n.value, :8].astype(np.int64)
        t0 = t[:, 0].min()
        print(f'trace {name}: {n.value} workgroups; start spread {(t[:,0].max()-t0)/100:.2f} us; phase end (last / median workgroup) us:',
              ' '.join(f'{(t[:,i].max()-t0)/100:.2f}/{(np.median(t[:,i])-t0)/100:.2f}' for i in range(1, 8)), flush=True)
