# ad-hoc timing (not a test): the 200,000-term / 2,000-qubit adjacency matrix and one rank's 25,000-row share of it, Four-Russians kernel,
# under the switches given as NAME=VALUE sets on the command line (sets separated by '/'); checksums must agree between the sets
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from symmer_amd import kernels, _lib
from symmer_amd.kernels import DeviceOp
lib = _lib.lib()

def timed(fn, reps=3):
    fn(); kernels.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    kernels.sync(); return (time.perf_counter() - t0) / reps

T = int(os.environ.get('ADJ_TERMS', 200000)); n = int(os.environ.get('ADJ_QUBITS', 2000))
sets = [dict(kv.split('=') for kv in s.split(',') if kv) for s in (sys.argv[1] if len(sys.argv) > 1 else '').split('/')]
C = DeviceOp.random(T, n, 0.3, seed=1239)
shapes = [(T // 8, 'share'), (T, 'full')]
bufs = {}
for rows, name in shapes:
    p = ctypes.c_void_p(); _lib.check(lib.symgpu_dev_alloc(rows * T, ctypes.byref(p))); bufs[name] = p
sums = {}
for env in sets:
    for k, v in env.items(): os.environ[k] = v
    for rows, name in shapes:
        _lib.check(lib.symgpu_prof_enable(1, 1))
        t = timed(lambda: _lib.check(lib.symgpu_commutes_dev(C.handle, 0, rows, C.handle, bufs[name])))
        nl, ms = ctypes.c_int64(0), ctypes.c_double(0)
        _lib.check(lib.symgpu_prof_enable(1, 0)); _lib.check(lib.symgpu_prof_read(1, ctypes.addressof(nl), ctypes.addressof(ms)))
        s = ctypes.c_uint64(0); _lib.check(lib.symgpu_dev_checksum_u8(bufs[name], rows * T, ctypes.addressof(s)))
        sums.setdefault(name, set()).add(s.value)
        print(f'{env} {name:5s} {rows}x{T}: call {t*1e3:8.3f} ms  main {ms.value/max(1,nl.value):8.3f} ms  checksum {s.value}', flush=True)
    for k in env: os.environ.pop(k, None)
print('checksums agree' if all(len(v) == 1 for v in sums.values()) else f'CHECKSUM MISMATCH {sums}')
