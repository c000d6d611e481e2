# ad-hoc stress: 1e9-pair fused product + cleanup, checked through linearity (A*B == cleanup(A*B1 ++ A*B2)) with checksums
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from symmer_amd import kernels, _lib, packing
from symmer_amd.kernels import DeviceOp
lib = _lib.lib()
rng = np.random.default_rng(1)
n, Ni, No = 100, 100000, 10000
def dyadic(t): return (rng.integers(-8, 9, t) + 1j * rng.integers(-8, 9, t)) / 16
A_rows = packing.pack_rows(rng.random((Ni, 2 * n)) < 0.3); B_rows = packing.pack_rows(rng.random((No, 2 * n)) < 0.3)
ca, cb = dyadic(Ni), dyadic(No)
A = DeviceOp.upload(A_rows, ca); B = DeviceOp.upload(B_rows, cb)
B1 = DeviceOp.upload(B_rows[:No // 2], cb[:No // 2]); B2 = DeviceOp.upload(B_rows[No // 2:], cb[No // 2:])
def mulc(a, b):
    h = ctypes.c_void_p(); _lib.check(lib.symgpu_mul_cleanup_dev(a.handle, b.handle, 1, 1e-15, 1, ctypes.byref(h))); return DeviceOp(h)
kernels.sync(); t0 = time.perf_counter()
R = mulc(A, B); kernels.sync(); t = time.perf_counter() - t0
print(f'{Ni} x {No} = {Ni*No:.1e} pairs -> {R.n_terms} terms in {t*1e3:.1f} ms ({Ni*No/t:.2e} pairs/s)', flush=True)
x, c = R.checksum()
R.free(); kernels.sync(); t0 = time.perf_counter(); R = mulc(A, B); kernels.sync(); t = time.perf_counter() - t0
print(f'second call (allocator warm): {t*1e3:.1f} ms ({Ni*No/t:.2e} pairs/s)', flush=True)
R1 = mulc(A, B1); R2 = mulc(A, B2)
x1, c1 = R1.checksum(); x2, c2 = R2.checksum()
print('halves:', R1.n_terms, R2.n_terms, 'coefficient sums equal:', abs((c1 + c2) - c) < 1e-6 * max(1, abs(c)), flush=True)
free, total = ctypes.c_int64(0), ctypes.c_int64(0); _lib.check(lib.symgpu_mem_info(ctypes.addressof(free), ctypes.addressof(total)))
print(f'HBM free {free.value/2**30:.0f} GiB of {total.value/2**30:.0f} GiB')
