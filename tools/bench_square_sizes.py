import sys, time
sys.path.insert(0, '/root/repo')
import ctypes
from symmer_amd import kernels, _lib
from symmer_amd.kernels import DeviceOp
for N in (8000, 10000, 11000, 11500, 12000, 14000):
    A = DeviceOp.random(N, 1000, 0.3, seed=5)
    ts = []
    for rep in range(4):
        h = ctypes.c_void_p()
        kernels.sync(); t0 = time.perf_counter()
        _lib.check(_lib.lib().symgpu_mul_cleanup_dev(A.handle, A.handle, 1, 1e-15, 1, ctypes.byref(h)))
        kernels.sync(); ts.append(time.perf_counter() - t0)
        R = DeviceOp(h); nt = R.n_terms; R.free()
    print(N, 'keys', N * (N + 1) // 2, 'terms', nt, 'ms', round(min(ts) * 1e3, 3), 'ns/pair', round(min(ts) * 1e9 / (N * N), 3), flush=True)
    A.free()
