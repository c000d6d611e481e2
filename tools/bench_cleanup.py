# ad-hoc: plain cleanup (device resident) timing at several sizes
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
for n, T, distinct in ((100, 10_000_000, 4_000_000), (1000, 10_000_000, 4_000_000), (1000, 1_000_000, 400_000), (100, 100_000, 60_000)):
    base = DeviceOp.random(distinct, n, 0.3, seed=7)
    rows, coeff = base.download()
    rng = np.random.default_rng(1)
    pick = rng.integers(0, distinct, T)
    op = DeviceOp.upload(rows[pick], rng.standard_normal(T) + 0j)
    def run():
        h = ctypes.c_void_p(); _lib.check(lib.symgpu_cleanup_dev(op.handle, 1e-15, 1, ctypes.byref(h))); r = DeviceOp(h); run.n = r.n_terms; r.free()
    t = timed(run)
    wq = (n + 63) // 64
    print(f'cleanup n={n} T={T:.0e} -> {run.n} terms: {t*1e3:.2f} ms  {T/t:.3e} terms/s  algorithmic {(T + run.n) * (16*wq+16) / t / 1e12:.2f} TB/s', flush=True)
    op.free(); base.free()
