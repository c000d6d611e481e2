# ad-hoc: small operators — LDS-resident single-workgroup chain kernel against the register chain (SYMGPU_CHAIN_LOCAL_T=0)
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from symmer_amd import kernels, packing
from symmer_amd.kernels import DeviceOp
rng = np.random.default_rng(7)
K, n = 2000, 1000
qs = packing.pack_rows(rng.random((K, 2 * n)) < 0.02)
ks = rng.integers(0, 4, K).astype(np.int32)
for T in (1, 16, 64, 128, 256, 512):
    raw = DeviceOp.random(T, n, 0.3, seed=11); dev = kernels.cleanup_dev(raw); raw.free()
    res = []
    for env in (None, '0'):
        if env is None: os.environ.pop('SYMGPU_CHAIN_LOCAL_T', None)
        else: os.environ['SYMGPU_CHAIN_LOCAL_T'] = env
        kernels.rotate_clifford_chain_dev(dev, qs[:45], ks[:45]).free(); kernels.sync()
        t0 = time.perf_counter(); out = kernels.rotate_clifford_chain_dev(dev, qs, ks); kernels.sync(); t1 = time.perf_counter() - t0
        out.free(); res.append(t1 / K * 1e6)
    os.environ.pop('SYMGPU_CHAIN_LOCAL_T', None)
    print(f'chain T={T:4d}: default {res[0]:6.2f} us per rotation, register chain {res[1]:6.2f} us per rotation', flush=True)
    dev.free()
