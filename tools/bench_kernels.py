# ad-hoc kernel timing (not a test): commutation slice, product slab, cfg3 fused product+cleanup
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from symmer_amd import kernels, _lib
from symmer_amd.kernels import DeviceOp
lib = _lib.lib()

def timed(fn, reps=3):
    fn(); kernels.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    kernels.sync(); return (time.perf_counter() - t0) / reps

what = sys.argv[1:] or ['commute', 'product', 'cfg3', 'rotate']
if 'commute' in what:
    for n in (2000, 1000, 100):
        C = DeviceOp.random(200000, n, 0.3, seed=1239); nrow = 25000
        buf = ctypes.c_void_p(); _lib.check(lib.symgpu_dev_alloc(nrow * 200000, ctypes.byref(buf)))
        t = timed(lambda: _lib.check(lib.symgpu_commutes_dev(C.handle, 0, nrow, C.handle, buf)))
        wq = (n + 63) // 64
        print(f'commute n={n}: {t*1e3:.2f} ms  {nrow*200000/t:.3e} pairs/s  bitop3 rate {nrow*200000/t*4*wq:.3e}/s', flush=True)
        _lib.check(lib.symgpu_dev_free(buf)); C.free()
if 'product' in what:
    A = DeviceOp.random(100000, 1000, 0.3, seed=1); B = DeviceOp.random(4096, 1000, 0.3, seed=2)
    out = DeviceOp.alloc(256 * 100000, 16, True); outr = DeviceOp.alloc(256 * 100000, 16, False)
    t = timed(lambda: _lib.check(lib.symgpu_mul_allpairs_dev(A.handle, B.handle, 0, 256, 1, out.handle)), 10)
    t2 = timed(lambda: _lib.check(lib.symgpu_mul_allpairs_dev(A.handle, B.handle, 0, 256, 1, outr.handle)), 10)
    print(f'product slab 256x1e5: rows+coeff {t*1e3:.3f} ms ({2.56e7*272/t/1e12:.2f} TB/s), rows only {t2*1e3:.3f} ms ({2.56e7*256/t2/1e12:.2f} TB/s), coeff {1e3*(t-t2):.3f} ms', flush=True)
    A.free(); B.free(); out.free(); outr.free()
if 'cfg3' in what:
    A = DeviceOp.random(10000, 1000, 0.3, seed=1237)
    def cfg3():
        h = ctypes.c_void_p(); _lib.check(lib.symgpu_mul_cleanup_dev(A.handle, A.handle, 1, 1e-15, 1, ctypes.byref(h))); DeviceOp(h).free()
    t = timed(cfg3, 3); print(f'cfg3 mul+cleanup 1e8 pairs: {t*1e3:.2f} ms  {1e8/t:.3e} pairs/s', flush=True)
    A.free()
if 'cfg3x' in what:
    A = DeviceOp.random(10000, 1000, 0.3, seed=1237)
    def cfg3x():
        h = ctypes.c_void_p(); _lib.check(lib.symgpu_mul_cleanup_dev(A.handle, A.handle, 1, -1.0, 1, ctypes.byref(h))); DeviceOp(h).free()
    t = timed(cfg3x, 3); print(f'cfg3x (no atomics experiment) {t*1e3:.2f} ms', flush=True)
    A.free()
if 'rotate' in what:
    from symmer_amd import packing
    rng = np.random.default_rng(5)
    P = DeviceOp.random(100000, 1000, 0.3, seed=1236)
    q = packing.pack_rows((rng.random((1, 2000)) < 0.3))[0]
    t = timed(lambda: kernels.rotate_single_dev(P, q, 0.3)[0].free(), 10)
    t2 = timed(lambda: kernels.rotate_single_dev(P, q, np.pi / 2)[0].free(), 10)
    print(f'rotation 1e5 terms: non-Clifford {t*1e3:.3f} ms, Clifford {t2*1e3:.3f} ms', flush=True)
