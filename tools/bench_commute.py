# ad-hoc timing (not a test): commutation kernels, register-tile vs Four-Russians, whole call and main-kernel events
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from symmer_amd import kernels, _lib
from symmer_amd.kernels import DeviceOp
lib = _lib.lib()

def timed(fn, reps=3):
    fn(); kernels.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    kernels.sync(); return (time.perf_counter() - t0) / reps

only = os.environ.get('BENCH_ONLY')
shapes = [(2000, 25000, 200000), (1000, 25000, 200000), (100, 25000, 200000), (2000, 4096, 65536), (2000, 1024, 16384), (1000, 100000, 1)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split(',')) for a in sys.argv[1:]]
for n, N, M in shapes:
    C = DeviceOp.random(max(N, M), n, 0.3, seed=1239)
    buf = ctypes.c_void_p(); _lib.check(lib.symgpu_dev_alloc(N * M, ctypes.byref(buf)))
    B = C
    if M != max(N, M):
        B = DeviceOp.random(M, n, 0.3, seed=77)
    res = {}
    sums = {}
    for mode in ('0', '1'):
        os.environ['SYMGPU_COMMUTE_M4R'] = mode
        for r in ((None,) if mode == '0' else ('16', '24', '48')):
            if only and (mode == '0' or r != only): continue
            if r: os.environ['SYMGPU_M4R_R'] = r
            _lib.check(lib.symgpu_prof_enable(1, 1))
            t = timed(lambda: _lib.check(lib.symgpu_commutes_dev(C.handle, 0, N, B.handle, buf)))
            nl, ms = ctypes.c_int64(0), ctypes.c_double(0)
            _lib.check(lib.symgpu_prof_enable(1, 0)); _lib.check(lib.symgpu_prof_read(1, ctypes.addressof(nl), ctypes.addressof(ms)))
            s = ctypes.c_uint64(0)
            _lib.check(lib.symgpu_dev_checksum_u8(buf, N * M, ctypes.addressof(s)))
            key = 'tile' if mode == '0' else 'm4r R=' + r
            sums[key] = s.value
            print(f'n={n} {N}x{M} {key:9s}: call {t*1e3:8.3f} ms  {N*M/t:.3e} pairs/s   main kernel {ms.value/max(1,nl.value):8.3f} ms   checksum {s.value}', flush=True)
    assert len(set(sums.values())) <= 1 or os.environ.get('SYMGPU_M4R_DBG'), sums
    _lib.check(lib.symgpu_dev_free(buf)); C.free()
    if B is not C: B.free()
os.environ.pop('SYMGPU_COMMUTE_M4R', None); os.environ.pop('SYMGPU_M4R_R', None)
