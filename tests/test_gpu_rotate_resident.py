"""The one-launch, LDS-resident rotation kernel (symmer_amd/csrc/rotate_resident.hip) against the multi-launch paths (bit for bit:
rows, row order, coefficients) and against the NumPy oracle of PauliwordOp._rotate_by_single_Pword (reference base.py:1090-1161).
The kernel only takes operators whose duplicate status and row hashes are known, i.e. from the second rotation of a handle on."""
import ctypes
import os
import subprocess
import sys
import numpy as np
import pytest
from symmer_amd import _lib, kernels, packing
from symmer_amd.kernels import DeviceOp
from oracle import oracle_np as onp
from _golden import assert_op_equal

pytestmark = pytest.mark.gpu
TOL = 1e-12
HERE = os.path.dirname(os.path.abspath(__file__))


def counter(which):
    v = ctypes.c_int64(-1)
    _lib.check(_lib.lib().symgpu_debug_counter(which, ctypes.addressof(v)))
    return v.value


def dyadic(rng, t):
    return (rng.integers(-8, 9, t) + 1j * rng.integers(-8, 9, t)) / 16.0


def operator_with_partners(rng, n, T, tiny=True):
    """A duplicate-free operator in which half of the rows have their P*Q partner present (merges), some coefficients are below
    the 1e-15 threshold (dropped rows of every class) — and the Pauli Q."""
    symp, c = onp.cleanup_op(rng.random((T, 2 * n)) < 0.3, dyadic(rng, T))
    q = rng.random(2 * n) < 0.3
    half = symp.shape[0] // 2
    symp, c = onp.cleanup_op(np.vstack([symp, symp[:half] ^ q]), np.hstack([c, dyadic(rng, half)]))
    if tiny and symp.shape[0] > 8:
        c = c.copy()
        c[rng.choice(symp.shape[0], symp.shape[0] // 8, replace=False)] *= 2.0 ** -60      # |c| ~ 1e-19: below the threshold, still exact
    return symp, c, q



def run_both(op, q_packed, ang, monkeypatch):
    """The same rotation of the same handle on the multi-launch path and on the resident kernel -> ((rows, coeff) or None) x 2."""
    monkeypatch.setenv('SYMGPU_ROT_RESIDENT', '0')
    a, allc_a = kernels.rotate_single_dev(op, q_packed, ang)
    monkeypatch.delenv('SYMGPU_ROT_RESIDENT')
    before = counter(1)
    b, allc_b = kernels.rotate_single_dev(op, q_packed, ang)
    assert counter(1) == before + 1, 'the resident kernel did not take (or did not complete) the rotation'
    assert allc_a == allc_b
    ra = None if allc_a else a.download()
    rb = None if allc_b else b.download()
    for h in (a, b):
        if h is not None:
            h.free()
    return ra, rb


ANGLES = (0.3, np.pi / 2, np.pi, -np.pi / 2, 3 * np.pi / 2, -1.1, 2 * np.pi)


@pytest.mark.parametrize('n,T', [(1000, 20000), (1000, 257), (130, 5000), (40, 3000), (64, 777), (100, 2100), (256, 1500), (300, 800), (449, 900),
                                 (1024, 1300), (2000, 700), (2048, 90), (5, 1), (5, 2), (70, 63), (70, 64), (70, 65), (1, 3), (700, 16500)])
def test_resident_rotation_equals_multilaunch_and_oracle(n, T, monkeypatch):
    rng = np.random.default_rng(9000 + 7 * n + T)
    symp, c, q = operator_with_partners(rng, n, T)
    qp = packing.pack_rows(q.reshape(1, -1))[0]
    up = DeviceOp.upload(packing.pack_rows(symp), c)
    op = kernels.cleanup_dev(up, zero_threshold=None)              # same rows (the input is duplicate free), now KNOWN to be so
    up.free()
    r0, c0 = op.download()
    assert np.array_equal(r0, packing.pack_rows(symp)) and np.array_equal(c0, c)
    for ang in ANGLES:
        ra, rb = run_both(op, qp, ang, monkeypatch)
        er, ec = onp.rotate_by_single_pword(symp, c, q, ang)
        if ra is None:
            assert rb is None and np.array_equal(er, symp)
            continue
        assert np.array_equal(ra[0], rb[0]), 'rows / row order differ between the two device paths'
        assert np.array_equal(ra[1], rb[1]), 'coefficients differ between the two device paths'
        clifford = abs(round(2 * ang / np.pi) - 2 * ang / np.pi) <= 1e-18
        got = packing.pack_rows(er)
        if clifford:
            assert np.array_equal(rb[0], got) and np.array_equal(rb[1], ec)
        else:
            keep_d, keep_o = np.abs(rb[1]) > TOL, np.abs(ec) > TOL
            assert np.array_equal(rb[0][keep_d], got[keep_o]) and np.allclose(rb[1][keep_d], ec[keep_o], rtol=0, atol=TOL)
    op.free()


def test_resident_rotation_all_commute_and_identity_q(monkeypatch):
    rng = np.random.default_rng(5)
    n, T = 100, 500
    symp = np.zeros((T, 2 * n), dtype=bool)
    symp[:, n:] = rng.random((T, n)) < 0.4                          # Z-type terms only
    symp, c = onp.cleanup_op(symp, dyadic(rng, T))
    up = DeviceOp.upload(packing.pack_rows(symp), c)
    op = kernels.cleanup_dev(up, zero_threshold=None)
    up.free()
    qz = np.zeros(2 * n, dtype=bool); qz[n:] = rng.random(n) < 0.5   # a Z string commutes with every term
    for q in (qz, np.zeros(2 * n, dtype=bool)):
        for ang in (0.3, np.pi / 2):
            ra, rb = run_both(op, packing.pack_rows(q.reshape(1, -1))[0], ang, monkeypatch)
            assert ra is None and rb is None
    op.free()


def test_resident_chain_of_rotations_vs_oracle():
    """perform_rotations-like chain on the device handles: every step after the first runs on the resident kernel."""
    rng = np.random.default_rng(77)
    n, T = 1000, 3000
    symp, c, q0 = operator_with_partners(rng, n, T, tiny=False)
    qs = [q0] + [rng.random(2 * n) < 0.3 for _ in range(5)]
    angs = [0.3, np.pi / 2, -0.9, 0.3, 3 * np.pi / 2, 1.7]
    cur = DeviceOp.upload(packing.pack_rows(symp), c)
    es, ec = symp, c
    before = counter(1)
    for q, ang in zip(qs + [q0], angs + [0.3]):
        res, allc = kernels.rotate_single_dev(cur, packing.pack_rows(q.reshape(1, -1))[0], ang)
        es, ec = onp.rotate_by_single_pword(es, ec, q, ang)
        if not allc:
            cur.free(); cur = res
    assert counter(1) >= before + 5
    rows, coeff = cur.download()
    cur.free()
    got = packing.pack_rows(es)
    kd, ko = np.abs(coeff) > TOL, np.abs(ec) > TOL
    assert np.array_equal(rows[kd], got[ko]) and np.allclose(coeff[kd], ec[ko], rtol=0, atol=TOL)


def test_resident_kernel_verification_failure_and_timeout_fall_back():
    """Fresh processes: (1) a deliberately weak row hash makes different rows share a hash — the kernel must report it and the
    multi-launch path must deliver the result; (2) a workgroup that never arrives makes the in-launch all-gather time out — same."""
    for mode in ('weakhash', 'timeout'):
        env = dict(os.environ)
        env.pop('SYMGPU_ROT_RESIDENT', None)
        r = subprocess.run([sys.executable, os.path.join(HERE, '_resident_fail_worker.py'), mode], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0 and 'RESIDENT_FAIL_OK' in r.stdout, (mode, r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.parametrize('n,T,K', [(1000, 20000, 25), (1000, 3000, 41), (1000, 129, 85), (100, 5000, 80), (64, 700, 40), (30, 2000, 39), (200, 600, 81),
                                   (2000, 900, 17), (2048, 300, 44), (1000, 100000, 6)])
def test_register_chain_equals_multilaunch_chain_and_oracle(n, T, K, monkeypatch):
    """A run of Clifford rotations with the rows in registers + one sort of the partition bits per 40 rotations (rotate_chain.hip)
    against the per-rotation multi-launch forms (SYMGPU_CHAIN_REG=0), bit for bit, and against the step-by-step oracle for the
    sizes the oracle walks in seconds.  Runs of more than 40 rotations are cut into segments; every k in 0..3; identity rotations."""
    rng = np.random.default_rng(12000 + n + T + K)
    symp, c = onp.cleanup_op(rng.random((T, 2 * n)) < 0.3, dyadic(rng, T))
    up = DeviceOp.upload(packing.pack_rows(symp), c)
    dev = kernels.cleanup_dev(up)
    up.free()
    q = rng.random((K, 2 * n)) < 0.3
    q[K // 2] = False                                                     # commutes with everything
    if K > 3:
        q[3, :n] = False                                                   # a Z string: commutes with every Z-type term only
    qs = packing.pack_rows(q)
    ks = rng.integers(0, 4, K).astype(np.int32)
    a = kernels.rotate_clifford_chain_dev(dev, qs, ks)
    monkeypatch.setenv('SYMGPU_CHAIN_REG', '0')
    b = kernels.rotate_clifford_chain_dev(dev, qs, ks)
    monkeypatch.delenv('SYMGPU_CHAIN_REG')
    ra, ca = a.download(); rb, cb = b.download()
    assert np.array_equal(ra, rb), 'rows / row order differ between the register chain and the multi-launch chain'
    assert np.array_equal(ca, cb)
    if T * K <= 200000:
        es, ec = symp, c
        for j in range(K):
            es, ec = onp.rotate_by_single_pword(es, ec, q[j], float(ks[j]) * np.pi / 2)
        assert np.array_equal(ra, packing.pack_rows(es)) and np.array_equal(ca, ec)
    for h in (a, b, dev):
        h.free()


@pytest.mark.parametrize('n,T,K', [(1000, 90, 30), (1000, 128, 25), (100, 3000, 20), (64, 9000, 9), (1000, 300000 // 1000 * 10, 12)])
def test_every_chain_form_gives_the_same_run_in_one_process(n, T, K, monkeypatch):
    """The path-selection switches are read on every call: the LDS-resident single-workgroup kernel, the L2-resident
    one, the two-launch and the four-launch forms and the register chain must all return the same rows, order and coefficients."""
    rng = np.random.default_rng(31000 + n + T)
    symp, c = onp.cleanup_op(rng.random((T, 2 * n)) < 0.3, dyadic(rng, T))
    up = DeviceOp.upload(packing.pack_rows(symp), c)
    dev = kernels.cleanup_dev(up)
    up.free()
    qs = packing.pack_rows(rng.random((K, 2 * n)) < 0.3)
    ks = rng.integers(0, 4, K).astype(np.int32)
    forms = [{}, {'SYMGPU_CHAIN_REG': '0'}, {'SYMGPU_CHAIN_REG': '0', 'SYMGPU_CHAIN_LOCAL_T': '0'}, {'SYMGPU_CHAIN_LOCAL_T': '0'}]
    results = []
    for env in forms:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        out = kernels.rotate_clifford_chain_dev(dev, qs, ks)
        results.append(out.download())
        out.free()
        for k in env:
            monkeypatch.delenv(k)
    for r, cc in results[1:]:
        assert np.array_equal(r, results[0][0]) and np.array_equal(cc, results[0][1])
    dev.free()


def test_growing_rotation_chain_does_not_meet_hipmalloc():
    """VERDICT r2 (API edge): a chain of non-Clifford rotations whose term count grows meets a new result size with every step; the
    allocator's arena serves those sizes without a hipMalloc per step (debug counter 3), so the first pass costs what later ones cost."""
    rng = np.random.default_rng(3)
    n = 1000
    P = DeviceOp.random(20000, n, 0.3, seed=77)
    qs = [packing.pack_rows((rng.random((1, 2 * n)) < 0.3))[0] for _ in range(5)]

    def chain():
        cur, sizes = P, []
        for q in qs:
            res, _ = kernels.rotate_single_dev(cur, q, 0.3)
            if cur is not P:
                cur.free()
            cur = res
            sizes.append(cur.n_terms)
        cur.free()
        return sizes
    m0 = counter(3)
    sizes = chain()
    m1 = counter(3)
    assert sizes == sorted(sizes) and sizes[-1] > 5 * sizes[0] // 2
    assert m1 - m0 <= 3, f'{m1 - m0} device allocations went to hipMalloc during the first pass of a 5-step chain'
    chain()
    assert counter(3) == m1
    P.free()
