# ad-hoc timing (not a test): chains of Clifford rotations on small operators (the circuit simulator's case, README claim 1)
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from symmer_amd import PauliwordOp, kernels, packing
from symmer_amd.kernels import DeviceOp
rng = np.random.default_rng(7)
n, K = 1000, 2000
qs = packing.pack_rows(rng.random((K, 2 * n)) < 0.02)
ks = rng.integers(0, 4, K).astype(np.int32)
for T in (1, 64, 1000, 8000):
    P = PauliwordOp(rng.random((T, 2 * n)) < 0.3, rng.standard_normal(T) + 0j).cleanup()
    dev = kernels.cleanup_dev(DeviceOp.upload(P.packed, P.coeff_vec))
    kernels.rotate_clifford_chain_dev(dev, qs[:10], ks[:10]).free(); kernels.sync()
    t0 = time.perf_counter(); out = kernels.rotate_clifford_chain_dev(dev, qs, ks); kernels.sync(); t1 = time.perf_counter() - t0
    out.free()
    # the same chain, one launch set per rotation
    cur = dev; t0 = time.perf_counter()
    for j in range(200):
        res, allc = kernels.rotate_single_dev(cur, qs[j], float(ks[j]) * np.pi / 2)
        if not allc:
            if cur is not dev: cur.free()
            cur = res
    kernels.sync(); t2 = (time.perf_counter() - t0) / 200
    rots = [(PauliwordOp._from_packed(qs[j:j + 1], n, [1]), float(ks[j]) * np.pi / 2) for j in range(K)]
    t0 = time.perf_counter(); R = P.perform_rotations(rots); t3 = time.perf_counter() - t0
    print(f'T={T:5d}: chain kernel {t1*1e3:8.2f} ms for {K} rotations = {t1/K*1e6:6.2f} us each; one by one {t2*1e6:6.1f} us each; '
          f'perform_rotations (Python API, {K} rotations) {t3*1e3:8.1f} ms', flush=True)
