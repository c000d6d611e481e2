import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from symmer_amd import parallel, kernels
from symmer_amd.kernels import DeviceOp
A = DeviceOp.random(10000, 1000, 0.3, seed=1237)
rows, coeff = A.download()
G = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cls = parallel.linear_row_classes(rows, max(1, (G - 1).bit_length()))
for rep in range(3):
    st = {}
    kernels.sync(); t0 = time.perf_counter()
    res = parallel.hash_partition_local_dev(A, A, 0, G, True, 1e-15, stats=st, classes=(cls, cls))
    kernels.sync(); t = time.perf_counter() - t0
    res.free()
print(G, t * 1e3, st)
