# randomised cross-check of the GF(2) elimination schedules (run on the GPU box): default against SYMGPU_GF2_FULL_PANEL=0,
# SYMGPU_GF2_FUSED_SELECT=0, SYMGPU_GF2_LOOKAHEAD=0, SYMGPU_GF2_SMALL=0 and — for small matrices — the NumPy restatement of the
# reference loop; random, sparse, banded, low-rank, duplicate-row and zero-row matrices
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from symmer_amd import kernels, packing
from oracle import oracle_np as onp
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.default_rng(seed)
bad = 0
t0 = time.time()
ENVS = [{}, {'SYMGPU_GF2_FUSED_SELECT': '0'}, {'SYMGPU_GF2_M4R': '0'}, {'SYMGPU_GF2_SMALL': '0'}, {'SYMGPU_GF2_FUSED_SELECT': '0', 'SYMGPU_GF2_SMALL': '0'}]
for case in range(n_cases):
    R = int(rng.choice([1, 2, 63, 64, 65, 100, 129, 300, 700, 1500, 2500]))
    C = int(rng.choice([1, 5, 64, 65, 200, 700, 3000, 9000, 16384, 16400, 30000]))
    if R * C > 3e7: C = int(3e7 // R)
    kind = int(rng.integers(0, 6))
    dens = float(rng.choice([0.001, 0.003, 0.02, 0.2, 0.5]))
    m = rng.random((R, C)) < dens
    if kind == 1:                                            # banded: every row leads near its index
        m[:] = False
        for r in range(R):
            c0 = min(C - 1, (r * C) // max(1, R)); w = min(C - c0, int(rng.integers(1, 40)))
            m[r, c0:c0 + w] = rng.random(w) < 0.6
    elif kind == 2:                                          # low rank: rows are XORs of a few basis rows
        k = max(1, min(R, 20)); basis = rng.random((k, C)) < max(dens, 0.02)
        m = (rng.integers(0, 2, (R, k)) @ basis.astype(np.int64)) % 2 == 1
    elif kind == 3:                                          # duplicates and zero rows
        m[rng.integers(0, R, max(1, R // 3))] = m[rng.integers(0, R, max(1, R // 3))]
        m[rng.integers(0, R, max(1, R // 5))] = False
    elif kind == 4:                                          # identity-like with noise on the right (the symmetry matrices' shape)
        m[:] = False
        for r in range(min(R, C)): m[r, r] = True
        if C > R: m[:, R:] = rng.random((R, C - R)) < dens
    packed = packing.pack_bits(m)
    outs = []
    for env in ENVS:
        for k in ('SYMGPU_GF2_FUSED_SELECT', 'SYMGPU_GF2_M4R', 'SYMGPU_GF2_SMALL'): os.environ.pop(k, None)
        os.environ.update(env)
        outs.append(kernels.rref(packed, want_pivots=True))
    ok = all(np.array_equal(outs[0][0], o[0]) and outs[0][1] == o[1] and np.array_equal(outs[0][2], o[2]) for o in outs[1:])
    why = '' if ok else 'GPU schedules differ'
    if R * C <= 2e6:
        ered, ecnt = onp.rref_noswap(m, count_xors=True)
        if not (np.array_equal(packing.unpack_bits(outs[0][0], C), ered) and outs[0][1] == ecnt):
            ok = False; why += ' oracle differs'
    if not ok:
        bad += 1
        print(f'MISMATCH case {case}: R={R} C={C} kind={kind} dens={dens}: {why}', flush=True)
print(f'stress gf2: {n_cases} cases, {bad} mismatches, {time.time()-t0:.1f} s')
sys.exit(1 if bad else 0)
