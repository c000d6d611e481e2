import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
from symmer_amd import kernels, _lib
from symmer_amd.kernels import DeviceOp
pairs = [tuple(int(v) for v in x.split("x")) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else ((1000, 600), (1200, 1000), (1500, 1000), (2000, 1500), (2500, 1600), (3000, 2000))
for Na, Nb in pairs:
    A = DeviceOp.random(Na, 1000, 0.3, seed=5); B = DeviceOp.random(Nb, 1000, 0.3, seed=6)
    ts = []
    for rep in range(4):
        h = ctypes.c_void_p()
        kernels.sync(); t0 = time.perf_counter()
        _lib.check(_lib.lib().symgpu_mul_cleanup_dev(A.handle, B.handle, 1, 1e-15, 1, ctypes.byref(h)))
        kernels.sync(); ts.append(time.perf_counter() - t0)
        R = DeviceOp(h); nt = R.n_terms; R.free()
    print(Na, Nb, 'keys', Na * Nb, 'terms', nt, 'ms', round(min(ts) * 1e3, 3), flush=True)
    A.free(); B.free()
