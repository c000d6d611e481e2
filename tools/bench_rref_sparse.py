# ad-hoc: GF(2) reduction of mid-size matrices, dense against sparse (700 x 700 @ density 0.003 within 2x of the dense rate)
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from symmer_amd import kernels, packing
rng = np.random.default_rng(11)
for R, C in ((700, 700), (1500, 3000), (4000, 54000)):
    for dens in (0.5, 0.05, 0.003):
        if R == 4000 and dens != 0.5: continue
        m = rng.random((R, C)) < dens
        packed = packing.pack_bits(m)
        kernels.rref(packed)
        t0 = time.perf_counter(); reps = 5
        for _ in range(reps): red, cnt = kernels.rref(packed)[:2]
        t = (time.perf_counter() - t0) / reps
        print(f'rref {R} x {C} density {dens}: {t*1e3:.3f} ms (incl. upload/download), {cnt} reference row-XORs', flush=True)
