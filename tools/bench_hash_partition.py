# ad-hoc: one rank's share of the hash-partitioned cfg3 product + cleanup (10^4 terms squared, 1,000 qubits), timed on one GPU:
# device-resident (csrc/partition.hip) and host-staged (parallel.hash_partition_local), G = 8 and 2
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from symmer_amd import parallel, kernels
from symmer_amd.kernels import DeviceOp
A = DeviceOp.random(10000, 1000, 0.3, seed=1237)
rows, coeff = A.download()
for G in (8, 4, 2):
    n_bits = max(1, (G - 1).bit_length())
    cls = parallel.linear_row_classes(rows, n_bits)
    for rep in range(3):
        st = {}
        kernels.sync(); t0 = time.perf_counter()
        res = parallel.hash_partition_local_dev(A, A, 0, G, True, 1e-15, stats=st, classes=(cls, cls))
        kernels.sync(); t = time.perf_counter() - t0
        nt = res.n_terms; res.free()
    print(f'G={G}: rank 0 share {nt} terms of {st["pairs_owned"]} owned pairs: device-resident {t*1e3:.2f} ms', flush=True)
t0 = time.perf_counter()
r, c, g = parallel.hash_partition_local(rows, coeff, rows, coeff, 0, 8, True, 1e-15)
print(f'G=8 host-staged: {(time.perf_counter()-t0)*1e3:.1f} ms')
