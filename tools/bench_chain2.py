# ad-hoc: crossover between the single-workgroup chain kernel and the two-launch form
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from symmer_amd import PauliwordOp, kernels, packing
from symmer_amd.kernels import DeviceOp
rng = np.random.default_rng(7)
n, K = 1000, 1000
qs = packing.pack_rows(rng.random((K, 2 * n)) < 0.02)
ks = rng.integers(0, 4, K).astype(np.int32)
for T in [int(x) for x in os.environ.get("CHAIN_TS", "64,96,128,160,192,256,384,100000").split(",")]:
    P = PauliwordOp(rng.random((T, 2 * n)) < 0.3, rng.standard_normal(T) + 0j).cleanup()
    dev = kernels.cleanup_dev(DeviceOp.upload(P.packed, P.coeff_vec))
    kk = K if T < 10000 else 128
    kernels.rotate_clifford_chain_dev(dev, qs[:10], ks[:10]).free(); kernels.sync()
    t0 = time.perf_counter(); out = kernels.rotate_clifford_chain_dev(dev, qs[:kk], ks[:kk]); kernels.sync(); t1 = time.perf_counter() - t0
    out.free()
    print(f'T={T:6d}: {t1/kk*1e6:6.2f} us per rotation', flush=True)
