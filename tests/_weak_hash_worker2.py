# as _weak_hash_worker.py, but the FIRST cleanup of the process is a fused product + cleanup (packed pair keys, squared operator),
# then a rotation on weak hashes: every path that consumes row hashes meets bulk collisions once.
import os, sys, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from symmer_amd import PauliwordOp, _lib
from oracle import oracle_np as onp
from _golden import assert_op_equal
mode = sys.argv[1]
rng = np.random.default_rng(32)
dy = lambda t: (rng.integers(-8, 9, t) + 1j * rng.integers(-8, 9, t)) / 16.0
n = 70
A = PauliwordOp(rng.random((80, 2 * n)) < 0.3, dy(80)); B = PauliwordOp(rng.random((50, 2 * n)) < 0.3, dy(50))
if mode == 'square':
    X, Y = A, A
elif mode == 'pair':
    X, Y = A, B
if mode in ('square', 'pair'):
    R = X * Y
    es, ec = onp.mul(X.symp_matrix, X.coeff_vec, Y.symp_matrix, Y.coeff_vec)
    assert np.array_equal(R.symp_matrix, es) and np.array_equal(R.coeff_vec, ec)
else:                                                                      # rotation first: hash join on weak hashes (seed 1)
    symp, c = onp.cleanup_op(np.vstack([A.symp_matrix, B.symp_matrix]), np.hstack([A.coeff_vec, B.coeff_vec]))
    q = rng.random(2 * n) < 0.3
    half = symp.shape[0] // 2
    symp, c = onp.cleanup_op(np.vstack([symp, symp[:half] ^ q]), np.hstack([c, dy(half)]))
    P = PauliwordOp(symp, c)
    for ang in (0.3, np.pi / 2):
        R = P._rotate_by_single_Pword(PauliwordOp(q.reshape(1, -1), [1]), ang)
        er, ec = onp.rotate_by_single_pword(symp, c, q, ang)
        assert_op_equal(R.symp_matrix, R.coeff_vec, er, ec, exact=ang != 0.3, tol=1e-12)
v = ctypes.c_int64(-1); _lib.check(_lib.lib().symgpu_debug_counter(0, ctypes.addressof(v)))
assert mode == 'rotate' or v.value >= 1, 'the weak first seed must have forced a reseed'   # the hash join itself verifies rows on every tag hit
print('WEAK_HASH_OK', mode, v.value, flush=True)
