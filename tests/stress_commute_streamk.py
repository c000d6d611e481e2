# Randomised shapes through the stream-K launch of the Four-Russians commutation kernel (csrc/commute_m4r7.hip): every tile height, ragged
# row / column tiles, row lengths that are not multiples of 16, operands with all-zero 7-bit groups — the stream-K table (and the fix-up
# variant) against the one-tile-per-workgroup table byte for byte, and that one against the C oracle on sampled blocks.
# (run on the GPU box): python tests/stress_commute_streamk.py [first_seed] [n_cases]
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from symmer_amd import kernels, packing
from oracle import oracle_c as oc
first = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 24
os.environ['SYMGPU_COMMUTE_M4R'] = '1'
bad = 0
t0 = time.time()
for case in range(count):
    rng = np.random.default_rng(first + case)
    r = int(rng.choice([16, 24, 48]))
    n = int(rng.choice([7, 20, 64, 100, 130, 300, 1000]))
    rt, ct = int(rng.integers(16, 24)), int(rng.integers(12, 20))
    if rt * ct < 256:
        ct = (256 + rt - 1) // rt + 1
    N = 32 * r * rt - int(rng.integers(0, 32 * r - 1))
    M = 2048 * ct - int(rng.integers(0, 2047))
    dens = float(rng.choice([0.3, 0.02]))
    a = packing.pack_rows(rng.random((N, 2 * n)) < dens); b = packing.pack_rows(rng.random((M, 2 * n)) < 0.3)
    if case % 3 == 0:
        a[:, 0] &= np.uint64(0xFFFF)                                   # groups 3.. of the first word all zero on the left: skipped steps
    os.environ['SYMGPU_M4R_R'] = str(r)
    os.environ['SYMGPU_M4R_STREAM'] = '0'
    ref = kernels.commutes(a, b)
    os.environ['SYMGPU_M4R_STREAM'] = '1'                              # forced: short operators take one tile per workgroup by themselves
    ok = True
    for _ in range(12):
        r0, c0 = int(rng.integers(0, max(1, N - 200))), int(rng.integers(0, max(1, M - 200)))
        ok = ok and np.array_equal(ref[r0:r0 + 200, c0:c0 + 200], oc.commutes(a[r0:r0 + 200], b[c0:c0 + 200]))
    ok = ok and np.array_equal(ref[N - 40:], oc.commutes(a[N - 40:], b))
    got = kernels.commutes(a, b)
    ok_s = np.array_equal(got, ref)
    os.environ['SYMGPU_M4R_FIXUP'] = '1'
    got = kernels.commutes(a, b)
    ok_f = np.array_equal(got, ref)
    del os.environ['SYMGPU_M4R_FIXUP']
    if not (ok and ok_s and ok_f):
        bad += 1
        print(f'MISMATCH case {first + case}: R={r} n={n} N={N} M={M} oracle={ok} stream={ok_s} fixup={ok_f}', flush=True)
print(f'stress commute stream-K: {count} cases from seed {first}, {bad} mismatches, {time.time() - t0:.1f} s')
