# ad-hoc: PauliwordOp.perform_rotations through the Python API (upload, chain, download)
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from symmer_amd import PauliwordOp
rng = np.random.default_rng(3)
n = 1000
for T in (100000, 10000, 1000):
    P = PauliwordOp(rng.random((T, 2 * n)) < 0.3, rng.standard_normal(T) + 0j)
    Q = PauliwordOp((rng.random(2 * n) < 0.3).reshape(1, -1), [1])
    P.perform_rotations([(Q, 0.3)] * 3)
    for K in (10, 100):
        t0 = time.perf_counter(); R = P.perform_rotations([(Q, 0.3)] * K); t = time.perf_counter() - t0
        print(f'perform_rotations: {T} terms, same Q x {K} (0.3 rad): {t*1e3:.2f} ms total, {t/K*1e6:.1f} us per rotation, {R.n_terms} terms out', flush=True)
    Qs = [PauliwordOp((rng.random(2 * n) < 0.3).reshape(1, -1), [1]) for _ in range(100)]
    t0 = time.perf_counter(); R = P.perform_rotations([(q, np.pi / 2) for q in Qs]); t = time.perf_counter() - t0
    print(f'perform_rotations: {T} terms, 100 Clifford rotations: {t*1e3:.2f} ms total, {t/100*1e6:.1f} us per rotation', flush=True)
