"""Does the cleanup's output stage depend on WHERE its output lands?  cfg3's P * P with dummy blocks of different sizes allocated first
(they shift every later block inside the allocator's arena); prints the step time per shift (the output stage is the part that moves)."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from symmer_amd import kernels, _lib
from symmer_amd.kernels import DeviceOp
A = DeviceOp.random(10000, 1000, 0.3, seed=1236)
def step():
    h = ctypes.c_void_p()
    _lib.check(_lib.lib().symgpu_mul_cleanup_dev(A.handle, A.handle, 1, 1e-15, 1, ctypes.byref(h)))
    return DeviceOp(h)
for mb in (0, 1, 3, 17, 64, 65, 129, 300, 513, 1000, 1025, 2049):
    dummy = DeviceOp.alloc(max(1, mb * 1024 * 1024 // 272), 16, with_coeff=True) if mb else None
    ts = []
    for rep in range(6):
        kernels.sync(); t0 = time.perf_counter()
        R = step()
        kernels.sync(); ts.append(time.perf_counter() - t0)
        R.free()
    print(f'dummy {mb:5d} MB: step {min(ts) * 1e3:.3f} ms (median {sorted(ts)[3] * 1e3:.3f})', flush=True)
    if dummy is not None: dummy.free()
