# randomised cross-check (run on the GPU box) of the flag pass that works from the operand hash tables (csrc/pair_dups.hip, the default
# where it applies) against the sorted flag pass (SYMGPU_CLEANUP_DIRECT=0) and the complete sort of all keys (SYMGPU_CLEANUP_SUSPECTS=0):
# squared operators and general products of 1.6e6 ... 3e7 keys around the path's limits (bucket widths 2^8 ... 2^12, operands that just fit /
# just do not fit a workgroup's LDS), with planted product rows that occur two, three and many times, repeated rows, an operand holding the
# identity, tiny coefficients; the C oracle on a sampled sub-product.
#     python tests/stress_pair_dups.py [seed] [cases]
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from symmer_amd import kernels, packing
from oracle import oracle_c as oc

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(seed)
bad = 0
t0 = time.time()


def dyadic(k):
    c = (rng.integers(-8, 9, k) + 1j * rng.integers(-8, 9, k)) / 8.0
    c[c == 0] = 0.5
    return c


for case in range(n_cases):
    n = int(rng.choice([14, 20, 33, 64, 100, 130]))
    squared = rng.random() < 0.5
    if squared:
        N = int(rng.choice([1800, 2500, 4000, 6000, 8000, 10000, 10100, 11000, 12200, 12400])); M = N   # (from 10,031 terms: 8,192 buckets; 12,300: beyond the path)
    else:
        N, M = [(1300, 1300), (2000, 1700), (4000, 3000), (6000, 900), (9000, 700), (5000, 5000), (12000, 600), (16000, 300)][int(rng.integers(0, 8))]
    A = rng.random((N, 2 * n)) < 0.3
    B = A if squared else rng.random((M, 2 * n)) < 0.3
    flavour = ['plain', 'planted', 'planted', 'triples', 'repeated', 'identity', 'many'][int(rng.integers(0, 7))]
    if flavour in ('planted', 'triples'):                    # rows that are products of other rows: product rows that occur twice / three times
        for _ in range(int(rng.integers(5, 200))):
            i, j, l = rng.integers(0, N, 3)
            A[l] = A[i] ^ A[j]
            if not squared:
                i, j, l = rng.integers(0, M, 3)
                B[l] = B[i] ^ B[j]
        if flavour == 'triples':
            for _ in range(30):
                i, j, k, l = rng.integers(0, N, 4)
                A[l] = A[i] ^ A[j] ^ A[k]
    if flavour == 'repeated':                                # a few rows many times (the hash buckets overflow: the path has to give up)
        A[rng.integers(0, N, N // 4)] = A[rng.integers(0, 5, N // 4)]
    if flavour == 'identity':
        A[int(rng.integers(0, N))] = False
        if not squared:
            B[int(rng.integers(0, M))] = False
    if flavour == 'many':                                    # one product row reached by ~40 pairs
        base = A[0].copy()
        for k in range(1, 20):
            A[2 * k] = A[2 * k - 1] ^ base
    if flavour not in ('repeated',):
        A = np.unique(A, axis=0); N = A.shape[0]
        if squared:
            B = A; M = N
        else:
            B = np.unique(B, axis=0); M = B.shape[0]
    a = dyadic(N); b = a if squared else dyadic(M)
    if rng.random() < 0.3:
        a[rng.random(N) < 0.05] = 1e-17
        if squared:
            b = a
    Ap = packing.pack_rows(A); Bp = Ap if squared else packing.pack_rows(B)
    thr = 1e-15
    outs = {}
    for name, env in (('direct', {}), ('sorted', {'SYMGPU_CLEANUP_DIRECT': '0'}), ('full', {'SYMGPU_CLEANUP_SUSPECTS': '0'})):
        os.environ.update(env)
        outs[name] = kernels.mul_cleanup(Ap, a, Bp, b, True, thr)
        for k in env:
            os.environ.pop(k)
    ok = all(outs['direct'][0].shape == o[0].shape and np.array_equal(outs['direct'][0], o[0]) and np.array_equal(outs['direct'][1], o[1]) for o in (outs['sorted'], outs['full']))
    # the oracle on the sub-product of 300 x 300 terms that holds planted rows (rows and order exact; coefficients exact for dyadic ones)
    sub_ok = True
    if flavour in ('plain', 'planted', 'triples', 'identity') and not np.any(np.abs(a) < 1e-10):
        si = np.sort(rng.choice(N, min(N, 300), replace=False)); so = si if squared else np.sort(rng.choice(M, min(M, 300), replace=False))
        g = kernels.mul_cleanup(Ap[si], a[si], Bp[so], b[so], True, thr)
        r = oc.mul(Ap[si], a[si], Bp[so], b[so], thr)
        sub_ok = g[0].shape == r[0].shape and np.array_equal(g[0], r[0]) and np.array_equal(g[1], r[1])
    if not (ok and sub_ok):
        bad += 1
        print(f'MISMATCH case {case} (seed {seed}): n={n} N={N} M={M} squared={squared} {flavour}: paths agree {ok}, oracle sub-product {sub_ok}; '
              f'rows {[o[0].shape[0] for o in outs.values()]}', flush=True)
print(f'stress pair_dups: {n_cases} cases from seed {seed}, {bad} mismatches, {time.time() - t0:.1f} s')
sys.exit(1 if bad else 0)
