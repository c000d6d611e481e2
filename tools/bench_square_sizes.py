import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes   # squared product + cleanup, 1,000 qubits, device resident: python tools/bench_square_sizes.py [N1,N2,...]
from symmer_amd import kernels, _lib
from symmer_amd.kernels import DeviceOp
Ns = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else (8000, 10000, 11000, 11500, 12000, 14000)
for N in Ns:
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
