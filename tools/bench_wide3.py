import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from symmer_amd import PauliwordOp
nq = 100_000_000
rngw = np.random.default_rng(1240)
bits = lambda: np.unpackbits(rngw.integers(0, 256, 2 * nq // 8, dtype=np.uint8)).astype(bool).reshape(1, -1)
A = PauliwordOp(bits(), [1.0]); B = PauliwordOp(bits(), [1.0])
(A * B)
for _ in range(3):
    t0 = time.perf_counter(); R = A * B; print('wide product, operands packed: %.2f ms' % ((time.perf_counter() - t0) * 1e3), R.n_terms, flush=True)
