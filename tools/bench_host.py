# ad-hoc: where the wall time of a one-by-one rotation goes on the host (debug counters 4-6: preparation, launch call, wait)
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from symmer_amd import kernels, packing, _lib
from symmer_amd.kernels import DeviceOp
rng = np.random.default_rng(5)
def counter(i):
    v = ctypes.c_int64(0); _lib.check(_lib.lib().symgpu_debug_counter(i, ctypes.byref(v))); return v.value
for T, n in ((100000, 1000), (1000, 1000)):
    P0 = DeviceOp.random(T, n, 0.3, seed=1236)
    P = kernels.cleanup_dev(P0); P0.free()
    q = packing.pack_rows((rng.random((1, 2 * n)) < 0.3))[0]
    for ang, name in ((0.3, 'non-Clifford'), (np.pi / 2, 'Clifford')):
        for _ in range(5): kernels.rotate_single_dev(P, q, ang)[0].free()
        kernels.sync()
        c0 = [counter(i) for i in (4, 5, 6)]; reps = 200
        t0 = time.perf_counter()
        for _ in range(reps): kernels.rotate_single_dev(P, q, ang)[0].free()
        kernels.sync(); t = time.perf_counter() - t0
        c1 = [counter(i) for i in (4, 5, 6)]
        d = [(b - a) / reps / 1e3 for a, b in zip(c0, c1)]
        print(f'{T} terms {name}: {t/reps*1e6:.1f} us per call = prep {d[0]:.2f} + launch call {d[1]:.2f} + wait {d[2]:.2f} + rest (python, free) {t/reps*1e6-sum(d):.2f}', flush=True)
    P.free()
