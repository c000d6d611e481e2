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
import ctypes
from symmer_amd import _lib
def counter(i):
    v = ctypes.c_int64(0); _lib.check(_lib.lib().symgpu_debug_counter(i, ctypes.byref(v))); return v.value
c0 = counter(3); h0 = counter(0)
ts = []
for _ in range(10):
    t0 = time.perf_counter(); run(); ts.append((time.perf_counter() - t0) * 1e3)
print('per call ms:', ' '.join(f'{x:.2f}' for x in ts), '| hipMalloc calls', counter(3) - c0, '| hash reseeds', counter(0) - h0, flush=True)
