"""Generate ``tests/golden/rotate_dup.npz`` by RUNNING THE REFERENCE (imported through ``ref_shim``): single-Pauli rotations of
operators that CONTAIN DUPLICATE ROWS, and rotations with a caller-supplied Clifford ``threshold``.

BUILD CONTAINER ONLY (needs /root/reference).  Data only: seeded inputs and the reference's outputs.
Run:  python oracle/tools/gen_golden_rotate_dup.py

Why a family of its own: for an odd multiple of pi/2 the reference forms ``anticom_self * Pword`` through ``__mul__``
(base.py:1143), which merges duplicate product rows and applies the 1e-15 threshold to the SUM, while rows of the commuting part
stay unmerged (base.py:1151-1154); even multiples merge nothing.  Cases: duplicates in both parts, duplicates that cancel
exactly, duplicates whose members are each below the threshold but whose sum is above it, and (``threshold`` field) angles one
ulp off a multiple of pi/2 rotated with a looser Clifford threshold (base.py:1146).
"""
import os, sys, warnings
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: F401
warnings.simplefilter('ignore')
import numpy as np
from symmer.operators import PauliwordOp

OUT = os.path.join(HERE, '..', '..', 'tests', 'golden')
rng = np.random.default_rng(20260)
cases, k = {}, 0


def add(**arrays):
    global k
    for key, val in arrays.items():
        a = np.asarray(val)
        cases[f'{k:04d}/{key}'] = a.astype(np.uint8) if a.dtype == bool else a
    k += 1


def dyadic(t):
    return (rng.integers(-8, 9, t) + 1j * rng.integers(-8, 9, t)) / 16.0


angles = (np.pi / 2, -np.pi / 2, np.pi, 3 * np.pi / 2, 5 * np.pi / 2, 0.0, 0.3, -1.1)
for trial in range(16):
    n = int((2, 5, 33, 70)[trial % 4]); t = int(rng.integers(3, 60))
    base = rng.random((t, 2 * n)) < 0.35
    reps = rng.integers(0, t, size=int(rng.integers(2, t + 2)))          # rows repeated, some of them several times
    symp = np.vstack([base, base[reps]])
    coeff = dyadic(symp.shape[0])
    if trial % 4 == 1:                                                    # a duplicate pair that cancels exactly
        symp = np.vstack([symp, base[0], base[0]]); coeff = np.hstack([coeff, [0.375 - 0.25j, -0.375 + 0.25j]])
    if trial % 4 == 2:                                                    # each member <= 1e-15, the sum above it
        symp = np.vstack([symp, base[1], base[1]]); coeff = np.hstack([coeff, [0.75e-15, 0.75e-15]])
        keep = ~np.all(symp[:-2] == base[1], axis=1)                      # no other copy of that row
        symp = np.vstack([symp[:-2][keep], symp[-2:]]); coeff = np.hstack([coeff[:-2][keep], coeff[-2:]])
    if trial % 4 == 3:                                                    # a lone row below the threshold (dropped for odd k only)
        lone = rng.random(2 * n) < 0.5
        symp = np.vstack([symp, lone]); coeff = np.hstack([coeff, [0.5e-15]])
    order = rng.permutation(symp.shape[0])
    symp, coeff = symp[order], coeff[order]
    q = rng.random(2 * n) < 0.45
    if not q.any():
        q[0] = True
    P = PauliwordOp(symp, coeff); Q = PauliwordOp(q.reshape(1, -1), [1])
    for ang in angles:
        R = P._rotate_by_single_Pword(Q, ang)
        add(in_symp=symp, in_coeff=coeff, q=q, angle=float(ang), threshold=1e-18, out_symp=R.symp_matrix,
            out_coeff=np.asarray(R.coeff_vec, dtype=complex), same_object=np.array(R is P))
# caller-supplied Clifford threshold: angles a few ulp off k*pi/2 are Clifford under 1e-12 and non-Clifford under the default
for trial in range(6):
    n = int((3, 20, 66)[trial % 3]); t = int(rng.integers(4, 40))
    P = PauliwordOp(rng.random((t, 2 * n)) < 0.35, dyadic(t)).cleanup()
    q = rng.random(2 * n) < 0.45
    q[0] = True
    Q = PauliwordOp(q.reshape(1, -1), [1])
    kmul = int((1, 3, 2, 5, 1, 3)[trial])
    ang = float(np.nextafter(kmul * np.pi / 2, 10.0)) * (1 + 2e-16 * trial)
    for thr in (1e-18, 1e-12):
        R = P._rotate_by_single_Pword(Q, ang, thr)
        add(in_symp=P.symp_matrix, in_coeff=np.asarray(P.coeff_vec, dtype=complex), q=q, angle=ang, threshold=thr, out_symp=R.symp_matrix,
            out_coeff=np.asarray(R.coeff_vec, dtype=complex), same_object=np.array(R is P))
cases['n_cases'] = np.array(k)
np.savez_compressed(os.path.join(OUT, 'rotate_dup.npz'), **cases)
print('rotate_dup:', k, 'cases,', os.path.getsize(os.path.join(OUT, 'rotate_dup.npz')), 'bytes')
