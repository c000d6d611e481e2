# ad-hoc: plain cleanup of 1e7 1,000-qubit terms (3.7e6 distinct), for rocprof breakdowns
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from symmer_amd import kernels, _lib
from symmer_amd.kernels import DeviceOp
lib = _lib.lib()
n, T, distinct = 1000, 10_000_000, 4_000_000
base = DeviceOp.random(distinct, n, 0.3, seed=7)
rows, coeff = base.download()
rng = np.random.default_rng(1)
pick = rng.integers(0, distinct, T)
op = DeviceOp.upload(rows[pick], rng.standard_normal(T) + 0j)
def run():
    h = ctypes.c_void_p(); _lib.check(lib.symgpu_cleanup_dev(op.handle, 1e-15, 1, ctypes.byref(h))); r = DeviceOp(h); run.n = r.n_terms; r.free()
run(); kernels.sync(); t0 = time.perf_counter()
for _ in range(3): run()
kernels.sync(); t = (time.perf_counter() - t0) / 3
print(f'cleanup n={n} T={T:.0e} -> {run.n} terms: {t*1e3:.2f} ms', flush=True)
