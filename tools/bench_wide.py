# ad-hoc: README claim 4 of the reference (README.md:54) — multiply two 100,000,000-qubit Pauli terms — through the drop-in API
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from symmer_amd.operators import PauliwordOp
from symmer_amd import kernels
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
rng = np.random.default_rng(4)
A = PauliwordOp(rng.random((1, 2 * n)) < 0.3, [1.0]); B = PauliwordOp(rng.random((1, 2 * n)) < 0.3, [1.0])
for rep in range(3):
    A._packed_cache = None; B._packed_cache = None
    t0 = time.perf_counter(); C = A * B; t = time.perf_counter() - t0
    print(f'n={n}: A * B (1 x 1 term) {t*1e3:.1f} ms, coefficient {C.coeff_vec[0]}', flush=True)
# oracle check on the host: rows XOR, phase exponent
xa, za, xb, zb = A.X_block[0], A.Z_block[0], B.X_block[0], B.Z_block[0]
rows_ok = np.array_equal(C.X_block[0], xa ^ xb) and np.array_equal(C.Z_block[0], za ^ zb)
e = (3 * (int((xa & za).sum()) + int((xb & zb).sum())) + int(((xa ^ xb) & (za ^ zb)).sum()) + 2 * int((xa & zb).sum())) % 4
print('rows ok', rows_ok, 'coefficient ok', np.isclose(C.coeff_vec[0], 1j ** e))
if os.environ.get('WIDE_PROFILE'):
    import cProfile, pstats
    A._packed_cache = None; B._packed_cache = None
    pr = cProfile.Profile(); pr.enable(); C = A * B; pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
