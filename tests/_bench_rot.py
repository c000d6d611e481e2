# ad-hoc: rotation timing only (for rocprof breakdowns)
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from symmer_amd import kernels, packing
from symmer_amd.kernels import DeviceOp
rng = np.random.default_rng(5)
P = DeviceOp.random(100000, 1000, 0.3, seed=1236)
q = packing.pack_rows((rng.random((1, 2000)) < 0.3))[0]
def timed(fn, reps):
    fn(); kernels.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    kernels.sync(); return (time.perf_counter() - t0) / reps
t = timed(lambda: kernels.rotate_single_dev(P, q, 0.3)[0].free(), 20)
t2 = timed(lambda: kernels.rotate_single_dev(P, q, np.pi / 2)[0].free(), 20)
print(f'rotation 1e5 terms: non-Clifford {t*1e3:.3f} ms, Clifford {t2*1e3:.3f} ms', flush=True)
