"""Second randomised sweep (round 2 paths): squared operators with and without duplicate rows (the identity segment, the i >= o
keys), rotations over every row length of the chunk-per-lane kernels, all-pairs products on both kernel paths, wide rows forced
through the word-parallel kernels.  Bit-exact against the NumPy / C oracles (dyadic coefficients; non-Clifford rotations 1e-12)."""
import numpy as np
import pytest
from symmer_amd import PauliwordOp, kernels, packing
from oracle import oracle_c as oc
from oracle import oracle_np as onp
from _golden import assert_op_equal

pytestmark = pytest.mark.gpu
N_QUBITS = [1, 5, 64, 65, 100, 128, 130, 200, 256, 300, 449, 512, 700, 1000, 1024, 1100, 2000, 2048, 4096]


def _dyadic(rng, t):
    return (rng.integers(-8, 9, t) + 1j * rng.integers(-8, 9, t)) / 16


@pytest.mark.parametrize('seed', range(40))
def test_fuzz_squared_operator(seed):
    rng = np.random.default_rng(11000 + seed)
    n = int(rng.choice(N_QUBITS))
    N = int(rng.integers(1, 140))
    symp = rng.random((N, 2 * n)) < rng.choice([0.05, 0.3])
    if seed % 3 == 0 and N > 2:                                        # duplicate rows in P: pairs (i, o), i != o, that multiply to the identity
        symp[rng.integers(0, N, N // 3)] = symp[rng.integers(0, N)]
    if seed % 5 == 0:
        n_small = min(n, 3)                                            # few qubits set: many product rows coincide
        symp[:, n_small:n] = False; symp[:, n + n_small:] = False
    c = _dyadic(rng, N)
    P = PauliwordOp(symp, c)
    R = P * P
    es, ec = onp.mul(symp, c, symp, c)
    assert np.array_equal(R.symp_matrix, es) and np.array_equal(R.coeff_vec, ec)


@pytest.mark.parametrize('seed', range(40))
def test_fuzz_rotations(seed):
    rng = np.random.default_rng(12000 + seed)
    n = int(rng.choice(N_QUBITS))
    T = int(rng.integers(1, 400))
    symp, c = onp.cleanup_op(rng.random((T, 2 * n)) < 0.3, _dyadic(rng, T))
    q = rng.random(2 * n) < 0.3
    half = symp.shape[0] // 2
    if half and seed % 2 == 0:                                         # P / P*Q partners: the join must merge them
        symp, c = onp.cleanup_op(np.vstack([symp, symp[:half] ^ q]), np.hstack([c, _dyadic(rng, half)]))
    P = PauliwordOp(symp, c); Q = PauliwordOp(q.reshape(1, -1), [1])
    for ang in (0.3, np.pi / 2, np.pi, 3 * np.pi / 2, -0.7):
        R = P._rotate_by_single_Pword(Q, ang)
        er, ec = onp.rotate_by_single_pword(symp, c, q, ang)
        clifford = abs(round(2 * ang / np.pi) - 2 * ang / np.pi) <= 1e-18
        assert_op_equal(R.symp_matrix, R.coeff_vec, er, ec, exact=clifford, tol=1e-12)
    rots = [(PauliwordOp((rng.random(2 * n) < 0.3).reshape(1, -1), [1]), a) for a in (np.pi / 2, 0.4, 3 * np.pi / 2, np.pi / 2, -1.3)]
    R = P.perform_rotations(rots)
    er, ec = onp.perform_rotations(symp, c, [(r.symp_matrix[0], a) for r, a in rots])
    assert_op_equal(R.symp_matrix, R.coeff_vec, er, ec, exact=False, tol=1e-12)


@pytest.mark.parametrize('seed', range(30))
def test_fuzz_allpairs_both_paths_and_wide(seed, monkeypatch):
    rng = np.random.default_rng(13000 + seed)
    n = int(rng.choice(N_QUBITS))
    N, M = int(rng.integers(1, 90)), int(rng.integers(1, 90))
    a = packing.pack_rows(rng.random((N, 2 * n)) < 0.3); b = packing.pack_rows(rng.random((M, 2 * n)) < 0.3)
    ca, cb = _dyadic(rng, N), _dyadic(rng, M)
    for env in ({'SYMGPU_PRODUCT_FUSED': '1'}, {'SYMGPU_PRODUCT_FUSED': '0'}, {'SYMGPU_WIDE': '1'}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for left in (True, False):
            r, c = kernels.mul_allpairs(a, ca, b, cb, left)
            er, ec = oc.mul_allpairs(a, ca, b, cb, left)
            assert np.array_equal(r, er) and np.array_equal(c, ec)
        assert np.array_equal(kernels.commutes(a, b), oc.commutes(a, b))
        r2, c2 = kernels.mul_cleanup(a, ca, b, cb, True, 1e-15)          # inner = a, outer = b: rows o * N + i, then first-occurrence cleanup
        er2, ec2 = oc.cleanup(*oc.mul_allpairs(a, ca, b, cb, True), 1e-15)
        assert np.array_equal(r2, er2) and np.array_equal(c2, ec2)
        for k in env:
            monkeypatch.delenv(k)
