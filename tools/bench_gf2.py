import sys, os, time, ctypes, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from symmer_amd import kernels, packing, _lib
from symmer_amd.kernels import DeviceOp
lib = _lib.lib()
rng = np.random.default_rng(1238)
import os
for (n, M) in (((2000, 50000),) if os.environ.get("GF2_ONLY_CFG4") else ((2000, 50000), (1000, 20000), (500, 10000))):
    symp = rng.random((M, 2 * n)) < 0.3
    symp[:, :32] = False
    H = DeviceOp.upload(packing.pack_rows(symp), np.ones(M, dtype=complex))
    for _ in range(8):
        q = packing.pack_rows((rng.random((1, 2 * n)) < 0.3))[0]
        res, allc = kernels.rotate_single_dev(H, q, np.pi / 2)
        if not allc:
            H.free(); H = res
    wq = (n + 63) // 64
    outg = np.zeros((2 * n, 2 * wq), dtype='<u8'); k = ctypes.c_int64(0); nx = ctypes.c_int64(0)
    def run():
        _lib.check(lib.symgpu_symmetry_kernel_dev(H.handle, n, outg.ctypes.data, 2 * n, ctypes.addressof(k), ctypes.addressof(nx)))
    run(); kernels.sync()
    t0 = time.perf_counter()
    for _ in range(3): run()
    kernels.sync(); t = (time.perf_counter() - t0) / 3
    print(f'n={n} M={M}: {t*1e3:.2f} ms, generators={k.value}, row_xors={nx.value}, {nx.value/t:.3e} row-XORs/s', flush=True)
