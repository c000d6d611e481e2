# ad-hoc timing (not a test): Four-Russians commutation of the first `rows` terms of a 200,000-term / 2,000-qubit operator against all of it
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from symmer_amd import kernels, _lib
from symmer_amd.kernels import DeviceOp
lib = _lib.lib()
T, n = 200000, 2000
C = DeviceOp.random(T, n, 0.3, seed=1239)
buf = ctypes.c_void_p(); _lib.check(lib.symgpu_dev_alloc(T * T, ctypes.byref(buf)))
for rows in [int(a) for a in sys.argv[1:]] or [12500, 25000, 50000, 100000, 200000]:
    fn = lambda: _lib.check(lib.symgpu_commutes_dev(C.handle, 0, rows, C.handle, buf))
    fn(); kernels.sync()
    _lib.check(lib.symgpu_prof_enable(1, 1))
    t0 = time.perf_counter()
    for _ in range(5): fn()
    kernels.sync(); t = (time.perf_counter() - t0) / 5
    nl, ms = ctypes.c_int64(0), ctypes.c_double(0)
    _lib.check(lib.symgpu_prof_enable(1, 0)); _lib.check(lib.symgpu_prof_read(1, ctypes.addressof(nl), ctypes.addressof(ms)))
    print(f'rows {rows:7d}: call {t*1e3:8.3f} ms  main {ms.value/max(1,nl.value):8.3f} ms   per 25,000 rows {t*1e3*25000/rows:7.3f} ms', flush=True)
