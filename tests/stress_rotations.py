# stress (not a test): random rotations, one-launch kernel against the multi-launch paths, bit for bit; runs of Clifford rotations,
# register chain against the multi-launch forms.  python3 tests/stress_rotations.py [cases] [seed]
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from symmer_amd import kernels, packing
from symmer_amd.kernels import DeviceOp
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
t0 = time.time()
bad = 0
for case in range(cases):
    n = int(rng.choice([1, 7, 40, 64, 65, 100, 128, 200, 300, 512, 1000, 1024, 1500, 2000, 2048]))
    T = int(rng.choice([1, 2, 3, 17, 63, 64, 65, 200, 1000, 1023, 1025, 5000, 20000, 70000]))
    if n * T > 6e7:
        T = max(1, int(6e7 // n))
    raw = DeviceOp.random(T, n, float(rng.choice([0.05, 0.3, 0.5])), seed=int(rng.integers(1, 1 << 30)))
    dev = kernels.cleanup_dev(raw); raw.free()
    # plant partners: append rows P ^ Q for a part of the operator, then clean again
    q = packing.pack_rows((rng.random((1, 2 * n)) < float(rng.choice([0.02, 0.3]))))[0]
    rows, coeff = dev.download()
    k = rows.shape[0] // 3
    if k:
        rows2 = np.vstack([rows, rows[:k] ^ q]); coeff2 = np.hstack([coeff, coeff[:k] * 0.5])
        dev.free()
        up = DeviceOp.upload(rows2, coeff2); dev = kernels.cleanup_dev(up); up.free()
    for ang in (0.3, -2.2, np.pi / 2, np.pi, 3 * np.pi / 2):
        os.environ['SYMGPU_ROT_RESIDENT'] = '0'
        a, ca = kernels.rotate_single_dev(dev, q, ang)
        os.environ.pop('SYMGPU_ROT_RESIDENT')
        b, cb = kernels.rotate_single_dev(dev, q, ang)
        os.environ['SYMGPU_ROT_HBM'] = '2'                         # the one-launch kernel with the rows left in memory (what operators beyond the chip take)
        c, cc = kernels.rotate_single_dev(dev, q, ang)
        os.environ.pop('SYMGPU_ROT_HBM')
        ok = ca == cb == cc
        if ok and not ca:
            ra, rb, rc = a.download(), b.download(), c.download()
            ok = np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1]) and np.array_equal(ra[0], rc[0]) and np.array_equal(ra[1], rc[1])
        for h in (a, b, c):
            if h is not None: h.free()
        if not ok:
            bad += 1; print(f'MISMATCH rotation case {case} n={n} T={T} angle={ang}', flush=True)
    K = int(rng.choice([1, 5, 39, 40, 41, 90]))
    qs = packing.pack_rows(rng.random((K, 2 * n)) < float(rng.choice([0.02, 0.3])))
    ks = rng.integers(0, 4, K).astype(np.int32)
    x = kernels.rotate_clifford_chain_dev(dev, qs, ks)
    os.environ['SYMGPU_CHAIN_REG'] = '0'
    y = kernels.rotate_clifford_chain_dev(dev, qs, ks)
    os.environ.pop('SYMGPU_CHAIN_REG')
    rx, ry = x.download(), y.download()
    if not (np.array_equal(rx[0], ry[0]) and np.array_equal(rx[1], ry[1])):
        bad += 1; print(f'MISMATCH chain case {case} n={n} T={T} K={K}', flush=True)
    for h in (x, y, dev): h.free()
print(f'stress: {cases} cases, {bad} mismatches, {time.time() - t0:.1f} s', flush=True)
sys.exit(1 if bad else 0)
