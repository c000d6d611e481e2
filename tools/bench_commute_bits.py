# ad-hoc: cfg5 slice, np.bool_ bytes vs bit-packed output of the Four-Russians commutation kernel
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from symmer_amd import kernels, _lib
from symmer_amd.kernels import DeviceOp
lib = _lib.lib()
def timed(fn, reps=5):
    fn(); kernels.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    kernels.sync(); return (time.perf_counter() - t0) / reps
n, T, nrow = 2000, 200000, 25000
C = DeviceOp.random(T, n, 0.3, seed=1239)
buf = ctypes.c_void_p(); _lib.check(lib.symgpu_dev_alloc(nrow * T, ctypes.byref(buf)))
t = timed(lambda: _lib.check(lib.symgpu_commutes_dev(C.handle, 0, nrow, C.handle, buf)))
t2 = timed(lambda: _lib.check(lib.symgpu_commutes_bits_dev(C.handle, 0, nrow, C.handle, buf)))
print(f'bytes {t*1e3:.2f} ms ({nrow*T/t:.3e} pairs/s)   bits {t2*1e3:.2f} ms ({nrow*T/t2:.3e} pairs/s)')
