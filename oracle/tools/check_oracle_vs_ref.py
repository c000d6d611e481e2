"""Pin ``oracle/oracle_np.py`` to the reference, imported here through ``ref_shim`` (BUILD CONTAINER
ONLY — needs /root/reference).  Randomised sweep over every restated function; exits non-zero on the
first mismatch.  Run:  python oracle/tools/check_oracle_vs_ref.py
"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import ref_shim  # noqa: F401
warnings.simplefilter('ignore')
import numpy as np
from symmer.operators import PauliwordOp, IndependentOp
from symmer.operators.utils import (_rref_binary, rref_binary, _cref_binary, cref_binary,
                                    symplectic_cleanup, check_independent)
from oracle import oracle_np as orc

rng = np.random.default_rng(20261002)


def dyadic(rng, t):
    return (rng.integers(-8, 9, t) + 1j * rng.integers(-8, 9, t)) / 16.0


def rand_op(n, t, density=0.3, kind='dyadic'):
    symp = rng.random((t, 2 * n)) < density
    c = dyadic(rng, t) if kind == 'dyadic' else rng.standard_normal(t) + 1j * rng.standard_normal(t)
    return symp, c


def same(a_rows, a_c, b_rows, b_c, exact=True):
    assert a_rows.shape == b_rows.shape, (a_rows.shape, b_rows.shape)
    assert np.array_equal(a_rows, b_rows)
    if exact:
        assert np.array_equal(a_c, b_c), np.max(np.abs(a_c - b_c))
    else:
        assert np.allclose(a_c, b_c, rtol=0, atol=1e-12)


n_checked = 0
for n in (1, 2, 3, 5, 63, 64, 65, 100, 130):
    for (N, M) in ((1, 1), (2, 7), (7, 2), (64, 5), (5, 64), (40, 40)):
        A = rand_op(n, N); B = rand_op(n, M)
        PA, PB = PauliwordOp(*A), PauliwordOp(*B)
        ref = PA * PB
        same(ref.symp_matrix, ref.coeff_vec, *orc.mul(*A, *B))
        # commutation
        assert np.array_equal(PA.commutes_termwise(PB), orc.commutes_termwise(A[0], B[0]))
        assert np.array_equal(PA.adjacency_matrix, orc.commutes_termwise(A[0], A[0]))
        # cleanup / add
        S = PA + PB
        same(S.symp_matrix, S.coeff_vec, *orc.cleanup_op(np.vstack([A[0], B[0]]), np.hstack([A[1], B[1]])))
        assert np.array_equal(PA.Y_count, orc.y_count(A[0]))
        # free-function cleanup without threshold
        r0 = symplectic_cleanup(np.vstack([A[0], A[0]]), np.hstack([A[1], -A[1]]))
        same(*r0, *orc.symplectic_cleanup(np.vstack([A[0], A[0]]), np.hstack([A[1], -A[1]])))
        n_checked += 1

# duplicate-heavy and cancelling inputs
for n, t in ((3, 500), (2, 300), (4, 1000)):
    A = rand_op(n, t)
    PA = PauliwordOp(*A)
    ref = PA * PA
    same(ref.symp_matrix, ref.coeff_vec, *orc.mul(*A, *A))
    ref = PA.cleanup()
    same(ref.symp_matrix, ref.coeff_vec, *orc.cleanup_op(*A))
    ref = PA - PA
    r = orc.cleanup_op(np.vstack([A[0], A[0]]), np.hstack([A[1], -A[1]]))
    same(ref.symp_matrix, ref.coeff_vec, *r)
    assert ref.n_terms == 0

# Gaussian coefficients: tolerance rule
for n, N, M in ((10, 50, 30), (70, 30, 50)):
    A = rand_op(n, N, kind='gauss'); B = rand_op(n, M, kind='gauss')
    ref = PauliwordOp(*A) * PauliwordOp(*B)
    same(ref.symp_matrix, ref.coeff_vec, *orc.mul(*A, *B), exact=False)

# rotations
angles = (0.3, -1.1, 2.0, 0.0, np.pi / 2, -np.pi / 2, np.pi, 3 * np.pi / 2, 2 * np.pi, 5 * np.pi / 2, -np.pi)
for trial in range(120):
    n = int(rng.integers(1, 70)); t = int(rng.integers(1, 200))
    symp, c = rand_op(n, t)
    # drop duplicate rows, then append P*Q partners to force merges
    symp, c = orc.cleanup_op(symp, c)
    q = rng.random(2 * n) < 0.4
    if trial % 2 == 0 and symp.shape[0] > 2:
        extra = symp[: symp.shape[0] // 2] ^ q
        symp = np.vstack([symp, extra]); c = np.hstack([c, dyadic(rng, extra.shape[0])])
        symp, c = orc.cleanup_op(symp, c)
    if symp.shape[0] == 0:
        continue
    P = PauliwordOp(symp, c); Q = PauliwordOp(q.reshape(1, -1), [1])
    for ang in angles:
        ref = P._rotate_by_single_Pword(Q, ang)
        exact = abs(round(2 * ang / np.pi) - 2 * ang / np.pi) <= 1e-18
        same(ref.symp_matrix, ref.coeff_vec, *orc.rotate_by_single_pword(symp, c, q, ang), exact=exact)
    rots = [(PauliwordOp((rng.random(2 * n) < 0.4).reshape(1, -1), [1]), float(a)) for a in (np.pi / 2, 0.3, np.pi / 2)]
    ref = P.perform_rotations(rots)
    got = orc.perform_rotations(symp, c, [(r.symp_matrix[0], a) for r, a in rots])
    same(ref.symp_matrix, ref.coeff_vec, *got, exact=False)
    n_checked += 1

# GF(2)
for R in (1, 5, 63, 64, 65, 200):
    for C in (1, 5, 63, 64, 65, 200):
        for dens in (0.5, 0.05):
            m = rng.random((R, C)) < dens
            if R > 3:
                m[R // 2] = False; m[R - 1] = m[0]
            assert np.array_equal(_rref_binary(m), orc.rref_noswap(m))
            assert np.array_equal(_cref_binary(m), orc.cref_noswap(m))
            if m.any():
                assert np.array_equal(rref_binary(m), orc.rref_ordered(m))
                assert np.array_equal(cref_binary(m), orc.cref_ordered(m))
            n_checked += 1

# symmetry generators with planted symmetries + independence + reconstruction
for n, t, k in ((4, 10, 2), (10, 40, 3), (40, 200, 5), (70, 100, 8)):
    symp, c = rand_op(n, t)
    symp[:, :k] = False
    P = PauliwordOp(symp, c)
    S = IndependentOp.symmetry_generators(P, commuting_override=True)
    got = orc.symmetry_generators_symp(symp)
    assert np.array_equal(S.symp_matrix, got), (n, t, k)
    assert got.shape[0] >= k
    assert check_independent(S) == orc.check_independent(got)
    G = P.generators
    assert np.array_equal(G.symp_matrix, orc.generators(symp))
    R_ref, mask_ref = P.generator_reconstruction(G)
    R_got, mask_got = orc.generator_reconstruction(symp, G.symp_matrix)
    assert np.array_equal(R_ref, R_got) and np.array_equal(mask_ref, mask_got)
    assert np.array_equal(P.sort('lex').symp_matrix, symp[orc.lex_order(symp)])
    n_checked += 1

# packing round trip
for n in (1, 63, 64, 65, 130):
    s = rng.random((7, 2 * n)) < 0.5
    assert np.array_equal(orc.unpack_rows(orc.pack_rows(s), n), s)

print(f'oracle_np matches the reference on {n_checked} randomized case groups: OK')
