import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from symmer_amd import kernels, packing
from symmer_amd.kernels import DeviceOp
rng = np.random.default_rng(5)
def timed(fn, reps):
    fn(); kernels.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    kernels.sync(); return (time.perf_counter() - t0) / reps
for T, n in ((200000, 1000), (1000000, 1000), (4000000, 200)):
    P0 = DeviceOp.random(T, n, 0.3, seed=1236); P = kernels.cleanup_dev(P0); P0.free()
    q = packing.pack_rows((rng.random((1, 2 * n)) < 0.3))[0]
    t = timed(lambda: kernels.rotate_single_dev(P, q, 0.3)[0].free(), 10)
    t2 = timed(lambda: kernels.rotate_single_dev(P, q, np.pi / 2)[0].free(), 10)
    wq = (n + 63) // 64
    by = T * (16 * wq + 16) * 2.5
    byc = T * (16 * wq + 16) * 2
    print(f'rotation {T} terms, {n} qubits: non-Clifford {t*1e3:.3f} ms = {by/t/1e12:.2f} TB/s, Clifford {t2*1e3:.3f} ms = {byc/t2/1e12:.2f} TB/s', flush=True)
    P.free()
