# more seeds for the randomised parity sweep of tests/test_gpu_fuzz.py, plus larger operands for product / commutation / cleanup
# (run on the GPU box): python tests/stress_ops.py [first_seed] [n_seeds]
import sys, os, time, inspect
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import test_gpu_fuzz as F
import test_gpu_parity as P
from symmer_amd import kernels, packing
from oracle import oracle_c as oc
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0
t0 = time.time()
fuzzers = [f for name, f in inspect.getmembers(F, inspect.isfunction) if name.startswith('test_fuzz') and list(inspect.signature(f).parameters) == ['seed']]
fuzzers.append(P.test_rotation_chain_with_duplicates_and_tiny_terms)      # perform_rotations: Clifford runs, mixed angles, duplicates, tiny terms
for seed in range(first, first + count):
    for f in fuzzers:
        try:
            f(seed)
        except AssertionError as e:
            bad += 1
            print(f'MISMATCH {f.__name__}({seed}): {str(e)[:200]}', flush=True)
# larger operands: both commutation kernels, the fused row stream, the sort with many tiles
rng = np.random.default_rng(first)
for case in range(max(10, count // 10)):
    n = int(rng.choice([10, 64, 100, 130, 1000, 2000]))
    N, M = int(rng.integers(500, 6000)), int(rng.integers(300, 3000))
    a = packing.pack_rows(rng.random((N, 2 * n)) < 0.3); b = packing.pack_rows(rng.random((M, 2 * n)) < 0.3)
    ca = (rng.integers(-8, 9, N) + 1j * rng.integers(-8, 9, N)) / 16; cb = (rng.integers(-8, 9, M) + 1j * rng.integers(-8, 9, M)) / 16
    ok = np.array_equal(kernels.commutes(a, b), oc.commutes(a, b))
    if N * M <= 3000000:
        r, c = kernels.mul_allpairs(a, ca, b, cb, True); er, ec = oc.mul_allpairs(a, ca, b, cb, True)
        ok = ok and np.array_equal(r, er) and np.array_equal(c, ec)
        r2, c2 = kernels.mul_cleanup(a, ca, b, cb, True, 1e-15); er2, ec2 = oc.cleanup(er, ec, 1e-15)
        ok = ok and np.array_equal(r2, er2) and np.array_equal(c2, ec2)
    if not ok:
        bad += 1
        print(f'MISMATCH large case {case}: n={n} N={N} M={M}', flush=True)
print(f'stress ops: seeds {first}..{first + count - 1} x {len(fuzzers)} fuzzers + large cases, {bad} mismatches, {time.time()-t0:.1f} s')
sys.exit(1 if bad else 0)
