# ad-hoc: host-side profile of perform_rotations on a long Clifford chain of a small observable (README claim 1)
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from symmer_amd import PauliwordOp, packing
rng = np.random.default_rng(7)
n, K, T = 1000, 2000, 64
qs = packing.pack_rows(rng.random((K, 2 * n)) < 0.02)
ks = rng.integers(0, 4, K)
P = PauliwordOp(rng.random((T, 2 * n)) < 0.3, rng.standard_normal(T) + 0j).cleanup()
rots = [(PauliwordOp._from_packed(qs[j:j + 1], n, [1]), float(ks[j]) * np.pi / 2) for j in range(K)]
P.perform_rotations(rots)
t0 = time.perf_counter(); R = P.perform_rotations(rots); t = time.perf_counter() - t0
print(f'perform_rotations: {t*1e3:.1f} ms for {K} rotations ({t/K*1e6:.2f} us each)')
pr = cProfile.Profile(); pr.enable(); P.perform_rotations(rots); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(16)
