"""The cleanup's output stage has a fast and a slow mode.  Does the mode change when the library's arenas are given back and
allocated again inside ONE process (symgpu_shutdown + symgpu_init)?  cfg3's P * P, min step time per life of the context."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from symmer_amd import kernels, _lib
from symmer_amd.kernels import DeviceOp
for life in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    _lib.init()
    A = DeviceOp.random(10000, 1000, 0.3, seed=1236)
    ts = []
    for rep in range(8):
        h = ctypes.c_void_p()
        kernels.sync(); t0 = time.perf_counter()
        _lib.check(_lib.lib().symgpu_mul_cleanup_dev(A.handle, A.handle, 1, 1e-15, 1, ctypes.byref(h)))
        kernels.sync(); ts.append(time.perf_counter() - t0)
        DeviceOp(h).free()
    A.free()
    print(f'life {life}: step {min(ts) * 1e3:.3f} ms (median {sorted(ts)[4] * 1e3:.3f})', flush=True)
    _lib.check(_lib.load().symgpu_shutdown())
    _lib._initialised_device = None
