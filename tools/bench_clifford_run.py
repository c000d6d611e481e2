# ad-hoc: a run of Clifford rotations, register chain (rotate_chain.hip) against the multi-launch forms
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from symmer_amd import kernels, packing
from symmer_amd.kernels import DeviceOp
rng = np.random.default_rng(7)
K = 200
for T, n in [(100000, 1000), (8000, 1000), (1000, 1000), (300, 1000), (100000, 100), (50000, 2000)]:
    qs = packing.pack_rows(rng.random((K, 2 * n)) < 0.3)
    ks = rng.integers(0, 4, K).astype(np.int32)
    raw = DeviceOp.random(T, n, 0.3, seed=11)
    dev = kernels.cleanup_dev(raw); raw.free()
    res = []
    for env in (None, '0'):
        if env is None: os.environ.pop('SYMGPU_CHAIN_REG', None)
        else: os.environ['SYMGPU_CHAIN_REG'] = env
        kernels.rotate_clifford_chain_dev(dev, qs[:45], ks[:45]).free(); kernels.sync()
        t0 = time.perf_counter(); out = kernels.rotate_clifford_chain_dev(dev, qs, ks); kernels.sync(); t1 = time.perf_counter() - t0
        out.free(); res.append(t1 / K * 1e6)
    os.environ.pop('SYMGPU_CHAIN_REG', None)
    print(f'chain T={T:6d} n={n:5d}: register chain {res[0]:7.2f} us per rotation, multi-launch {res[1]:7.2f} us per rotation', flush=True)
    dev.free()
