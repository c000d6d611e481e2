"""Randomised parity sweep through the C ABI against the oracle: many small shapes, including empty operands, single
qubits, word-boundary qubit counts and duplicate-heavy inputs.  Bit-exact (dyadic coefficients)."""
import numpy as np
import pytest
from symmer_amd import kernels, packing
from oracle import oracle_c as oc
from oracle import oracle_np as onp

pytestmark = pytest.mark.gpu
N_QUBITS = [1, 2, 3, 7, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 200, 257]


def _dyadic(rng, t):
    return (rng.integers(-8, 9, t) + 1j * rng.integers(-8, 9, t)) / 16


def _rows(rng, t, n, pool=None):
    if pool is None:
        return packing.pack_rows(rng.random((t, 2 * n)) < rng.choice([0.05, 0.3, 0.6]))
    base = packing.pack_rows(rng.random((pool, 2 * n)) < 0.4)
    return base[rng.integers(0, pool, t)]


@pytest.mark.parametrize('seed', range(40))
def test_fuzz_product_cleanup_commute(seed):
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.choice(N_QUBITS))
    N, M = int(rng.integers(0, 300)), int(rng.integers(0, 300))
    dup = seed % 3 == 0
    a = _rows(rng, N, n, pool=5 if dup else None); b = _rows(rng, M, n, pool=4 if dup else None)
    ca, cb = _dyadic(rng, N), _dyadic(rng, M)
    # all-pairs product (row order o*Ni + i) and fused product + cleanup, both operand orders
    for left in (True, False):
        r, c = kernels.mul_allpairs(a, ca, b, cb, left)
        er, ec = oc.mul_allpairs(a, ca, b, cb, left)
        assert np.array_equal(r, er) and np.array_equal(c, ec)
        r2, c2 = kernels.mul_cleanup(a, ca, b, cb, left, 1e-15)
        er2, ec2 = oc.cleanup(er, ec, 1e-15) if er.shape[0] else (er, ec)
        assert np.array_equal(r2, er2) and np.array_equal(c2, ec2)
    # plain cleanup with and without threshold
    stacked = np.vstack([a, b]); sc = np.hstack([ca, cb])
    for thr in (1e-15, None):
        r3, c3 = kernels.cleanup(stacked, sc, thr)
        er3, ec3 = oc.cleanup(stacked, sc, thr) if stacked.shape[0] else (stacked, sc)
        assert np.array_equal(r3, er3) and np.array_equal(c3, ec3)
    # commutation table
    assert np.array_equal(kernels.commutes(a, b), oc.commutes(a, b) if N and M else np.ones((N, M), dtype=bool))


@pytest.mark.parametrize('seed', range(30))
def test_fuzz_rref_and_symmetry(seed):
    rng = np.random.default_rng(7000 + seed)
    R, C = int(rng.integers(1, 260)), int(rng.integers(1, 700))
    dens = float(rng.choice([0.02, 0.2, 0.5]))
    m = rng.random((R, C)) < dens
    if seed % 4 == 0 and R > 3:
        m[R // 2] = m[0] ^ m[1]                                   # planted dependency
        m[R - 1] = False                                          # zero row
    red, count, piv = kernels.rref(packing.pack_bits(m), want_pivots=True)
    ered, ecount = onp.rref_noswap(m, count_xors=True)
    assert np.array_equal(packing.unpack_bits(red, C), ered) and count == ecount
    n = int(rng.choice([1, 3, 20, 64, 65, 130])); M = int(rng.integers(1, 200))
    h = rng.random((M, 2 * n)) < 0.3
    k = int(rng.integers(0, n + 1))
    h[:, :k] = False                                              # Z_0..Z_{k-1} commute with everything
    gens, _ = kernels.symmetry_kernel(packing.pack_rows(h), n)
    assert np.array_equal(packing.unpack_rows(gens, n), onp.symmetry_generators_symp(h))


@pytest.mark.parametrize('seed', range(30))
def test_fuzz_rotation(seed):
    rng = np.random.default_rng(5000 + seed)
    n = int(rng.choice(N_QUBITS)); T = int(rng.integers(1, 400))
    symp = np.unique(rng.random((T, 2 * n)) < 0.4, axis=0)        # a cleaned operator has no duplicate rows
    rng.shuffle(symp)
    coeff = _dyadic(rng, symp.shape[0])
    coeff[coeff == 0] = 1
    q = rng.random(2 * n) < 0.5
    if seed % 5 == 0 and symp.shape[0] > 1:
        symp[1] = symp[0] ^ q                                     # forces a merge of P and P*Q
        symp = np.unique(symp, axis=0)
        coeff = coeff[:symp.shape[0]]
    from symmer_amd.kernels import DeviceOp
    op = DeviceOp.upload(packing.pack_rows(symp), coeff)
    for angle in (np.pi / 2, np.pi, -np.pi / 2, 3 * np.pi / 2, 0.0):
        res, allc = kernels.rotate_single_dev(op, packing.pack_rows(q.reshape(1, -1))[0], angle)
        er, ec = onp.rotate_by_single_pword(symp, coeff, q, angle)
        if allc:
            assert np.array_equal(er, symp)
        else:
            r, c = res.download(); res.free()
            assert np.array_equal(packing.unpack_rows(r, n), er) and np.array_equal(c, ec)
    op.free()


@pytest.mark.parametrize('seed', range(24))
def test_fuzz_round2_paths(seed, monkeypatch):
    """Random shapes through the round-2 code paths: the Four-Russians commutation kernel forced on small ragged operands (every
    tile height, bytes and bits epilogues), the squared-operator cleanup against the general pair path, rotations of operators
    WITH duplicate rows, and single-launch Clifford chains — all against the oracle."""
    from symmer_amd import PauliwordOp
    rng = np.random.default_rng(12000 + seed)
    n = int(rng.choice(N_QUBITS))
    # (1) commutation, Four-Russians kernel forced
    N, M = int(rng.integers(1, 700)), int(rng.integers(1, 2600))
    a, b = _rows(rng, N, n), _rows(rng, M, n)
    monkeypatch.setenv('SYMGPU_COMMUTE_M4R', '1')
    monkeypatch.setenv('SYMGPU_M4R_R', str(rng.choice([16, 24, 48])))
    assert np.array_equal(kernels.commutes(a, b), oc.commutes(a, b))
    for var in ('SYMGPU_COMMUTE_M4R', 'SYMGPU_M4R_R'):
        monkeypatch.delenv(var, raising=False)
    # (2) squared operator (duplicate-heavy every third seed): shortcut == general pair path == oracle
    T = int(rng.integers(1, 260))
    s = _rows(rng, T, n, pool=6 if seed % 3 == 0 else None)
    cs = _dyadic(rng, T)
    fast = kernels.mul_cleanup(s, cs, s, cs, True, 1e-15)
    er, ec = oc.mul(s, cs, s, cs)
    assert np.array_equal(fast[0], er) and np.array_equal(fast[1], ec)
    # (3) a rotation of an operator with duplicate rows, Clifford and not
    symp = packing.unpack_rows(s, n)
    q = rng.random(2 * n) < 0.5
    P = PauliwordOp(symp, cs); Q = PauliwordOp(q.reshape(1, -1), [1])
    for ang in (float(rng.integers(-2, 6)) * np.pi / 2, 0.37):
        R = P._rotate_by_single_Pword(Q, ang)
        exp_r, exp_c = onp.rotate_by_single_pword(symp, cs, q, ang)
        clifford = abs(round(2 * ang / np.pi) - 2 * ang / np.pi) <= 1e-18
        assert np.array_equal(R.symp_matrix, exp_r)
        assert np.array_equal(R.coeff_vec, exp_c) if clifford else np.allclose(R.coeff_vec, exp_c, rtol=0, atol=1e-12)
    # (4) a run of Clifford rotations (one launch) on the cleaned operator
    C = P.cleanup()
    rots = [((rng.random(2 * n) < 0.4), float(rng.integers(-2, 6)) * np.pi / 2) for _ in range(int(rng.integers(2, 40)))]
    R = C.perform_rotations([(PauliwordOp(qq.reshape(1, -1), [1]), ang) for qq, ang in rots])
    exp_r, exp_c = onp.perform_rotations(C.symp_matrix, C.coeff_vec, rots)
    assert np.array_equal(R.symp_matrix, exp_r) and np.array_equal(R.coeff_vec, exp_c)
