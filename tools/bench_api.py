# ad-hoc end-to-end timing of the Python API (host buffers in, host results out: packing + PCIe + kernels), not a test
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from symmer_amd.operators import PauliwordOp
from symmer_amd import kernels

def timed(fn, reps=3):
    fn(); t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    return (time.perf_counter() - t0) / reps, r

np.random.seed(0)
for n, T in ((100, 1000), (100, 10000), (1000, 3000)):
    A = PauliwordOp.random(n, T); B = PauliwordOp.random(n, T)
    t_pack, _ = timed(lambda: __import__('symmer_amd').packing.pack_rows(A.symp_matrix))
    def mul():
        A._packed_cache = None; B._packed_cache = None
        return A * B
    t_mul, C = timed(mul)
    t_unpack, _ = timed(lambda: __import__('symmer_amd').packing.unpack_rows(C.packed, n), 1)
    t_com, _ = timed(lambda: A.commutes_termwise(B))
    t_add, _ = timed(lambda: A + B)
    q = PauliwordOp.random(n, 1, complex_coeffs=False); q.coeff_vec[:] = 1
    t_rot, _ = timed(lambda: A.perform_rotations([(q, 0.3)] * 4))
    print(f'n={n} T={T}: pack {t_pack*1e3:.2f} ms | A*B ({T*T:.1e} pairs -> {C.n_terms} terms) {t_mul*1e3:.1f} ms = {T*T/t_mul:.2e} pairs/s | unpack result {t_unpack*1e3:.1f} ms | '
          f'commutes_termwise {t_com*1e3:.1f} ms = {T*T/t_com:.2e} pairs/s | A+B {t_add*1e3:.2f} ms | 4 rotations {t_rot*1e3:.2f} ms', flush=True)
