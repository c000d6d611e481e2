# ad-hoc: four non-Clifford rotations with growing term count (1e5 -> 5e5), per step and per switch
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from symmer_amd import kernels, packing
from symmer_amd.kernels import DeviceOp
rng = np.random.default_rng(1236)
P = DeviceOp.random(100000, 1000, 0.3, seed=1236)
qs = [packing.pack_rows((rng.random((1, 2000)) < 0.3))[0] for _ in range(4)]
def chain4():
    ts, cur = [], P
    for q in qs:
        t0 = time.perf_counter()
        res, allc = kernels.rotate_single_dev(cur, q, 0.3)
        kernels.sync(); ts.append((time.perf_counter() - t0) * 1e6)
        if cur is not P: cur.free()
        cur = res
    n = cur.n_terms; cur.free()
    return ts, n
for rep in range(3):
    ts, n = chain4()
    print(f'pass {rep}: ' + ' '.join(f'{t:8.1f}' for t in ts) + f' us  (total {sum(ts):.0f} us, {n} terms)', flush=True)
