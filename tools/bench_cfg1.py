# ad-hoc: BASELINE cfg1 (100 qubits, 500 terms squared) through the drop-in API, with a host-side profile
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from symmer_amd import PauliwordOp, kernels
rng = np.random.default_rng(1235)
P = PauliwordOp(rng.random((500, 200)) < 0.3, rng.standard_normal(500) + 1j * rng.standard_normal(500))
(P * P)
def run():
    P._packed_cache = None
    return P * P
t0 = time.perf_counter()
for _ in range(20): R = run()
t = (time.perf_counter() - t0) / 20
print(f'cfg1 P * P: {t*1e3:.3f} ms, {R.n_terms} terms', flush=True)
if os.environ.get('PROFILE'):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(20): run()
    pr.disable(); pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
