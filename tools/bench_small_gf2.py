# ad-hoc: latency of small GF(2) reductions (stabiliser-sized matrices)
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from symmer_amd import kernels, packing
rng = np.random.default_rng(3)
for R, C, dens in ((10, 40, 0.2), (40, 200, 0.05), (64, 2000, 0.01), (64, 8000, 0.3), (30, 60, 0.5)):
    m = packing.pack_bits(rng.random((R, C)) < dens)
    kernels.rref(m); t0 = time.perf_counter()
    for _ in range(20): kernels.rref(m)
    print(f'rref {R}x{C} density {dens}: {(time.perf_counter()-t0)/20*1e3:.3f} ms', flush=True)
