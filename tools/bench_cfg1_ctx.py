import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from symmer_amd import PauliwordOp, kernels, _lib
from symmer_amd.kernels import DeviceOp
def counter(i):
    v = ctypes.c_int64(0); _lib.check(_lib.lib().symgpu_debug_counter(i, ctypes.byref(v))); return v.value
rng = np.random.default_rng(1235)
P = PauliwordOp(rng.random((500, 200)) < 0.3, rng.standard_normal(500) + 1j * rng.standard_normal(500))
def run():
    P._packed_cache = None
    return P * P
def measure(tag):
    for _ in range(4): run()
    c0 = counter(3); ts = []
    for _ in range(8):
        t0 = time.perf_counter(); run(); ts.append((time.perf_counter() - t0) * 1e3)
    print(tag, ' '.join(f'{x:.2f}' for x in ts), '| hipMalloc', counter(3) - c0, flush=True)
measure('fresh      :')
# what bench.py does first: a big product slab (6.5 GB output) and its scratch
A = DeviceOp.random(100000, 1000, 0.3, seed=1); B = DeviceOp.random(100000, 1000, 0.3, seed=2)
out = DeviceOp.alloc(256 * 100000, 16, with_coeff=True)
_lib.check(_lib.lib().symgpu_mul_allpairs_dev(A.handle, B.handle, 0, 256, 1, out.handle)); kernels.sync()
measure('after slab :')
out.free(); A.free(); B.free()
measure('after free :')
# cfg3-size cleanup as in the extras
C = DeviceOp.random(10000, 1000, 0.3, seed=3)
h = ctypes.c_void_p(); _lib.check(_lib.lib().symgpu_mul_cleanup_dev(C.handle, C.handle, 1, 1e-15, 1, ctypes.byref(h))); DeviceOp(h).free(); C.free()
measure('after cfg3 :')
# what bench.py's cpu_baseline leg does on the host before the extras: large NumPy temporaries
big = [np.random.default_rng(i).random((1000, 250 * 8)) for i in range(40)]
s = sum(float((b[:, None, :64] < 0.5).sum()) for b in big[:4])
del big
measure('after numpy:')
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20): run()
pr.disable(); pstats.Stats(pr).sort_stats('tottime').print_stats(12)
