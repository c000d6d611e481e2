# ad-hoc: wide single-term operators (1e7 / 1e8 qubits) through the drop-in API: which paths are serial over the words of a row?
import sys, os, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from symmer_amd.operators import PauliwordOp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
rng = np.random.default_rng(4)
A = PauliwordOp(rng.random((1, 2 * n)) < 0.3, [1.0]); B = PauliwordOp(rng.random((1, 2 * n)) < 0.3, [1.0])
def timed(name, fn):
    fn()
    t0 = time.perf_counter(); r = fn(); t = time.perf_counter() - t0
    print(f'n={n} {name}: {t*1e3:.1f} ms', flush=True)
    return r
timed('A * B', lambda: A * B)
timed('A + B', lambda: A + B)
timed('commutes_termwise', lambda: A.commutes_termwise(B))
timed('Y_count', lambda: PauliwordOp(A.symp_matrix, [1.0]).Y_count)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    timed('rotate 0.3', lambda: A._rotate_by_single_Pword(B, 0.3))
    timed('rotate pi/2', lambda: A._rotate_by_single_Pword(B, np.pi / 2))
