"""Generate ``tests/golden/noncontextual.npz`` by RUNNING THE REFERENCE (through ``ref_shim``): ``PauliwordOp.is_noncontextual``
(base.py:1074-1088) and ``check_adjmat_noncontextual`` (utils.py:567-589) on random operators (contextual), clique-structured
operators (noncontextual by construction: universally commuting terms plus cliques that anticommute with each other), such operators
with one term flipped (contextual again), all-commuting operators, operators with duplicated terms and operators of fewer than four
terms (always True).  Stored per case: the symplectic matrix, the reference's answer for the operator and for its adjacency matrix.
Data only.  BUILD CONTAINER ONLY.  Run: python oracle/tools/gen_golden_noncontextual.py
"""
import os, sys, warnings
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: F401
warnings.simplefilter('ignore')
import numpy as np
from symmer.operators import PauliwordOp
from symmer.operators.utils import check_adjmat_noncontextual

OUT = os.path.join(HERE, '..', '..', 'tests', 'golden')
rng = np.random.default_rng(4242)
cases, k = {}, 0


def add(symp, kind):
    global k
    symp = np.asarray(symp, dtype=bool)
    P = PauliwordOp(symp, np.ones(symp.shape[0]))
    cases[f'{k:04d}/symp'] = symp.astype(np.uint8)
    cases[f'{k:04d}/kind'] = np.array(kind)
    cases[f'{k:04d}/is_noncontextual'] = np.array(bool(P.is_noncontextual))
    cases[f'{k:04d}/adjmat_noncontextual'] = np.array(bool(check_adjmat_noncontextual(P.adjacency_matrix)))
    k += 1


def clique_structured(n, n_univ, cliques):
    """Universally commuting Z strings on qubits 1.. plus cliques {A_c * Z-string}: A_c pairwise anticommuting single-qubit-0/1 Paulis, so
    terms of one clique commute with each other (same A_c) and anticommute with every other clique."""
    heads = [(1, 0, 0, 0), (0, 0, 1, 0), (1, 0, 1, 0)]                       # X0, Z0, Y0 on qubit 0: pairwise anticommuting
    rows = []
    for _ in range(n_univ):
        r = np.zeros(2 * n, dtype=bool); r[n + 1:] = rng.random(n - 1) < 0.5
        rows.append(r)
    for c, size in enumerate(cliques):
        x0, x1, z0, z1 = heads[c]
        for _ in range(size):
            r = np.zeros(2 * n, dtype=bool); r[n + 1:] = rng.random(n - 1) < 0.5
            r[0], r[n] = bool(x0), bool(z0)
            rows.append(r)
    rows = np.array(rows)
    return rows[rng.permutation(rows.shape[0])]


# 0: random operators (contextual with overwhelming probability once they have a few dozen terms; small ones go either way)
for n, T in ((2, 4), (2, 6), (3, 5), (3, 9), (4, 12), (6, 30), (10, 60), (40, 300), (70, 200), (130, 150)):
    for _ in range(3):
        add(rng.random((T, 2 * n)) < 0.4, 0)
# 1: clique-structured, noncontextual by construction
for n, nu, cl in ((4, 2, (2, 2)), (6, 5, (4, 3)), (10, 0, (6, 5, 4)), (30, 20, (20, 20)), (70, 1, (40, 30, 10)), (8, 3, (5,)), (5, 6, ())):
    for _ in range(2):
        add(clique_structured(n, nu, cl), 1)
# 2: the same with ONE extra term that commutes with part of a clique only
for n, nu, cl in ((6, 5, (4, 3)), (10, 0, (6, 5, 4)), (30, 20, (20, 20)), (70, 1, (40, 30, 10))):
    s = clique_structured(n, nu, cl)
    extra = np.zeros(2 * n, dtype=bool); extra[1] = True                      # X1: anticommutes with the terms whose string has Z1
    add(np.vstack([s, extra]), 2)
# 3: everything commutes (Z strings; X strings)
for n, T in ((3, 8), (20, 100), (64, 64), (65, 200)):
    z = np.zeros((T, 2 * n), dtype=bool); z[:, n:] = rng.random((T, n)) < 0.4
    add(z, 3)
    x = np.zeros((T, 2 * n), dtype=bool); x[:, :n] = rng.random((T, n)) < 0.4
    add(x, 3)
# 4: duplicated terms (equal rows in the adjacency matrix beyond the clique structure)
for n, T in ((3, 10), (6, 40), (20, 120)):
    base = rng.random((max(2, T // 4), 2 * n)) < 0.4
    add(base[rng.integers(0, base.shape[0], T)], 4)
    s = clique_structured(n, 2, (3, 3))
    add(np.vstack([s, s[:3]]), 4)
# 5: fewer than four terms: always True (base.py:1084-1086)
for n, T in ((1, 1), (2, 2), (3, 3), (5, 3)):
    add(rng.random((T, 2 * n)) < 0.5, 5)
# 6: the reference's own known answers (tests/test_operators/test_base.py:581-595)
noncon = ['IIII', 'IIIZ', 'IIZI', 'IIZZ', 'IZII', 'IZIZ', 'ZIII', 'ZIIZ', 'ZIZI', 'ZIZZ', 'ZZII', 'ZZIZ', 'IXXX', 'IXYY', 'IYXY', 'IYYX']
con = ['IIII', 'IIIZ', 'IIZI', 'IZII', 'ZIII', 'IIXI', 'IXII', 'XIII', 'ZZZZ']
for plist in (noncon, con):
    add(PauliwordOp.from_list(plist).symp_matrix, 6)
cases['n_cases'] = np.array(k)
np.savez_compressed(os.path.join(OUT, 'noncontextual.npz'), **cases)
ans = [bool(cases[f'{i:04d}/is_noncontextual']) for i in range(k)]
print('noncontextual:', k, 'cases,', sum(ans), 'noncontextual,', os.path.getsize(os.path.join(OUT, 'noncontextual.npz')), 'bytes')
