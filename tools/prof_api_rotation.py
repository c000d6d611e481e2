"""Where the host time of `P._rotate_by_single_Pword(Q, 0.3)` goes (drop-in API around the 27 us one-launch rotation kernel)."""
import cProfile, pstats, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from symmer_amd import PauliwordOp, kernels
P = bench.host_operator(100000, 1000, 1236)
rng = np.random.default_rng(1)
Q = PauliwordOp(rng.random((1, 2000)) < 0.3, [1])
for _ in range(5):
    R = P._rotate_by_single_Pword(Q, 0.3)
kernels.sync()
N = 200
t0 = time.perf_counter()
for _ in range(N):
    R = P._rotate_by_single_Pword(Q, 0.3)
kernels.sync()
print('per call us (result replaced each time):', (time.perf_counter() - t0) / N * 1e6)
keep = []
t0 = time.perf_counter()
for _ in range(50):
    keep.append(P._rotate_by_single_Pword(Q, 0.3))
kernels.sync()
print('per call us (results kept):', (time.perf_counter() - t0) / 50 * 1e6)
del keep
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    R = P._rotate_by_single_Pword(Q, 0.3)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
