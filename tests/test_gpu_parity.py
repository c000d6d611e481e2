"""GPU parity tests: the HIP path (through the C-ABI and the drop-in classes) against
(1) the committed golden fixtures produced by running the reference, and
(2) the C / NumPy oracles on the same seeded inputs at sizes the oracle finishes in seconds.
Bar: bit-exact rows, row order, commutation tables and GF(2) matrices; coefficients bit-exact for dyadic inputs
and within 1e-12 for Gaussian inputs (rows with |c| <= 1e-12 discarded on both sides — SURVEY.md §7)."""
import numpy as np
import pytest
from symmer_amd import PauliwordOp, IndependentOp, kernels, packing
from symmer_amd.operators import (symplectic_cleanup, _rref_binary, rref_binary, _cref_binary, cref_binary, matmul_GF2,
                                  mul_symplectic)
from oracle import oracle_np as onp
from oracle import oracle_c as oc
from _golden import family, known, known_single_qubit, as_bool, unpackbits_matrix, assert_op_equal, rotate_empty_cases

pytestmark = pytest.mark.gpu
TOL = 1e-12


def dyadic(rng, t):
    return (rng.integers(-8, 9, t) + 1j * rng.integers(-8, 9, t)) / 16.0


# ---------------------------------------------------------------- known answers (reference's own tests) ----
def test_known_answers():
    k = known()
    P = PauliwordOp(as_bool(k['ycount_symp']), np.ones(4))
    assert np.all(P.Y_count == np.array([0, 0, 3, 0]))                                   # test_base.py:510-515
    P = PauliwordOp.from_list(['XXX', 'YYY', 'XXX', 'YYY'], [1, 1, -1, 1])
    assert P == PauliwordOp.from_list(['YYY'], [2])                                       # :537-544
    c = P.cleanup()
    assert_op_equal(c.symp_matrix, c.coeff_vec, k['cleanup_out_symp'], k['cleanup_out_coeff'])
    Z = PauliwordOp(as_bool(k['pl1_symp']), np.zeros(4))
    assert Z.cleanup().n_terms == 0 and Z.cleanup().symp_matrix.shape == (0, 6)           # :531-535, :124-130
    P1 = PauliwordOp.from_list(['III', 'XXX', 'YYY', 'ZZZ']); P2 = PauliwordOp.from_list(['ZXZ', 'XZX', 'XYZ', 'ZIX'])
    assert np.array_equal(P1.commutes_termwise(P2), as_bool(k['pl1_commutes_pl2']))       # :554-566
    assert np.array_equal(P2.adjacency_matrix, as_bool(k['pl2_adjacency']))               # :568-579
    assert np.array_equal(P1.anticommutes_termwise(P2), ~as_bool(k['pl1_commutes_pl2']))
    op1 = PauliwordOp.from_list(['XYXZ', 'YYII']); op2 = PauliwordOp.from_list(['YYZZ', 'XIXZ', 'XZZI'])
    assert np.array_equal(op1.commutes_termwise(op2), as_bool(k['doc_commutes']))         # base.py:944-951
    A = PauliwordOp(as_bool(k['small_A_symp']), k['small_A_coeff']); B = PauliwordOp(as_bool(k['small_B_symp']), k['small_B_coeff'])
    for nm, R in (('AB', A * B), ('BA', B * A)):
        assert_op_equal(R.symp_matrix, R.coeff_vec, k[f'small_{nm}_symp'], k[f'small_{nm}_coeff'])
    P = PauliwordOp.from_list(['XZ', 'ZX', 'XZ', 'II', 'ZX'], [1, 2, 3, 4, -2]).cleanup()
    assert P.to_dictionary == {'XZ': 4, 'II': 4} and list(P.to_dictionary) == ['XZ', 'II']


def test_single_qubit_multiplication_table():
    for pair, (res, (re, im)) in known_single_qubit().items():                            # test_base.py:596-613
        P1, P2 = PauliwordOp.from_dictionary({pair[0]: 1}), PauliwordOp.from_dictionary({pair[1]: 1})
        assert P1 * P2 == PauliwordOp.from_dictionary({res: complex(re, im)})
        row, c = mul_symplectic(P1.symp_matrix[0], 1, P2.symp_matrix[0], 1)
        assert c == complex(re, im)


def test_add_sub_random():
    np.random.seed(7)
    P = PauliwordOp.random(3, 10)
    assert P + P == P * 2                                                                 # :546-548
    assert (P - P).n_terms == 0                                                           # :550-552
    assert sum([P, P, P]) == P * 3
    E = PauliwordOp(np.zeros((0, 6), dtype=bool), [])
    assert (P * E).symp_matrix.shape == (0, 6) and (E * P).symp_matrix.shape == (0, 6)    # SURVEY §8a'
    assert P.commutes_termwise(E).shape == (10, 0) and E.commutes_termwise(P).shape == (0, 10)
    ee = E + E
    assert ee.n_terms == 1 and ee.coeff_vec[0] == 0 and not ee.symp_matrix.any()
    assert (P ** 2) == P * P and (P ** 0) == PauliwordOp.from_list(['III'], [1])
    assert P.commutes(P) and hash(P) == hash(P.copy())


# ---------------------------------------------------------------- golden families --------------------------
@pytest.mark.parametrize('case', family('mul'))
def test_mul_golden(case):
    A = PauliwordOp(as_bool(case['a_symp']), case['a_coeff']); B = PauliwordOp(as_bool(case['b_symp']), case['b_coeff'])
    R = A * B
    assert_op_equal(R.symp_matrix, R.coeff_vec, case['out_symp'], case['out_coeff'], exact=bool(case['exact']), tol=TOL)
    if A.n_terms and B.n_terms:
        # unfused path (materialised product + separate cleanup) must agree with the fused one
        inner, outer, left = (B, A, False) if A.n_terms < B.n_terms else (A, B, True)
        rows, coeff = kernels.mul_allpairs(inner.packed, inner.coeff_vec, outer.packed, outer.coeff_vec, left)
        rows2, coeff2 = kernels.cleanup(rows, coeff, 1e-15)
        assert_op_equal(packing.unpack_rows(rows2, A.n_qubits), coeff2, case['out_symp'], case['out_coeff'],
                        exact=bool(case['exact']), tol=TOL)


@pytest.mark.parametrize('case', family('cleanup'))
def test_cleanup_golden(case):
    s = as_bool(case['in_symp']); thr = float(case['thr'])
    exact = bool(np.all(np.asarray(case['in_coeff']) * 16 == np.round(np.asarray(case['in_coeff']) * 16)))
    if thr < 0:
        rows, coeff = symplectic_cleanup(s, case['in_coeff'], None)
    else:
        P = PauliwordOp(s, case['in_coeff']).cleanup(thr)
        rows, coeff = P.symp_matrix, P.coeff_vec
    assert_op_equal(rows, coeff, case['out_symp'], case['out_coeff'], exact=exact, tol=TOL)


@pytest.mark.parametrize('case', family('commute'))
def test_commute_golden(case):
    A = PauliwordOp(as_bool(case['a_symp']), np.ones(case['a_symp'].shape[0]))
    B = PauliwordOp(as_bool(case['b_symp']), np.ones(case['b_symp'].shape[0]))
    assert np.array_equal(A.commutes_termwise(B), as_bool(case['out']))
    assert np.array_equal(A.adjacency_matrix, as_bool(case['adj']))


@pytest.mark.parametrize('case', family('rotate'))
def test_rotate_golden(case):
    P = PauliwordOp(as_bool(case['in_symp']), case['in_coeff'])
    n = P.n_qubits
    if int(case['chain']):
        rots = [(PauliwordOp(as_bool(q).reshape(1, -1), [1]), float(a)) for q, a in zip(case['q'], case['angle'])]
        R = P.perform_rotations(rots)
        assert_op_equal(R.symp_matrix, R.coeff_vec, case['out_symp'], case['out_coeff'], exact=False, tol=TOL)
    else:
        ang = float(case['angle'])
        Q = PauliwordOp(as_bool(case['q']).reshape(1, -1), [1])
        R = P._rotate_by_single_Pword(Q, ang)
        clifford = abs(round(2 * ang / np.pi) - 2 * ang / np.pi) <= 1e-18
        assert_op_equal(R.symp_matrix, R.coeff_vec, case['out_symp'], case['out_coeff'], exact=clifford, tol=TOL)
        if np.all(onp.commutes_termwise(P.symp_matrix, Q.symp_matrix)):
            assert R is P                                                                  # base.py:1131-1133
    assert n == case['in_symp'].shape[1] // 2


@pytest.mark.parametrize('general', [False, True])
@pytest.mark.parametrize('case', family('rotate_dup'))
def test_rotate_duplicate_rows_and_threshold_golden(case, general, monkeypatch):
    """Reference outputs for operators WITH duplicate rows (odd k: the rotated anticommuting rows are merged and the threshold
    applies to the sums, the commuting rows stay as they are; base.py:1143-1154) and for caller-supplied Clifford thresholds
    (base.py:1146), on the fast paths (duplicates detected on the device) and with the general path forced."""
    if general:
        monkeypatch.setenv('SYMGPU_ROTATE_GENERAL', '1')
    P = PauliwordOp(as_bool(case['in_symp']), case['in_coeff'])
    ang, thr = float(case['angle']), float(case['threshold'])
    Q = PauliwordOp(as_bool(case['q']).reshape(1, -1), [1])
    R = P._rotate_by_single_Pword(Q, ang, thr)
    clifford = abs(round(2 * ang / np.pi) - 2 * ang / np.pi) <= thr
    assert_op_equal(R.symp_matrix, R.coeff_vec, case['out_symp'], case['out_coeff'], exact=clifford, tol=TOL)
    assert (R is P) == bool(case['same_object'])


def test_rotations_that_lose_every_term_golden():
    """Reference outputs for rotations of operators that lose every term, have none, or are 0 * I: which of the two states a step
    of perform_rotations ends in depends on where the terms were lost (rotation: cleanup() gives 0 * I; the loop's own cleanup:
    no terms; non-Clifford ``commute_self + anticom_part``: base.py:1159-1161) — base.py:631-632, utils.py:275-278.  K = 1 .. 4."""
    for c in rotate_empty_cases():
        P = PauliwordOp(c['in_symp'], c['in_coeff'])
        if c['kind'] == 'single':
            R = P._rotate_by_single_Pword(PauliwordOp(c['q'][:1], [1]), c['angles'][0])
            assert (R is P) == c['same_object'], c
        else:
            R = P.perform_rotations([(PauliwordOp(q.reshape(1, -1), [1]), a) for q, a in zip(c['q'], c['angles'])])
        assert R.symp_matrix.shape == c['out_symp'].shape and np.array_equal(R.symp_matrix, c['out_symp']), c
        assert np.allclose(R.coeff_vec, c['out_coeff'], rtol=0, atol=1e-15), c


def test_rotation_chain_from_operator_with_duplicates():
    """perform_rotations on an operator that still holds duplicate rows (the reference rotates first and cleans up after,
    base.py:1185) — against the step-by-step oracle."""
    rng = np.random.default_rng(91)
    n, t = 40, 60
    base_rows = rng.random((t, 2 * n)) < 0.3
    symp = np.vstack([base_rows, base_rows[rng.integers(0, t, 25)]])
    c = (rng.integers(-8, 9, symp.shape[0]) + 1j * rng.integers(-8, 9, symp.shape[0])) / 16.0
    P = PauliwordOp(symp, c)
    rots = [((rng.random(2 * n) < 0.4), a) for a in (np.pi / 2, 3 * np.pi / 2, 0.3, np.pi / 2, np.pi)]
    R = P.perform_rotations([(PauliwordOp(q.reshape(1, -1), [1]), a) for q, a in rots])
    er, ec = onp.perform_rotations(symp, c, rots)
    assert_op_equal(R.symp_matrix, R.coeff_vec, er, ec, exact=False, tol=TOL)


def test_saturating_nonclifford_chain_vs_oracle():
    """VERDICT r5 item 7 / SURVEY 8d cfg2: a chain of > 100 NON-Clifford rotations that cycles through 8 fixed generators (a Trotter circuit,
    symmer/evolution/exponentiation.py:26-38).  The term set saturates at (seed terms) x 2^8 products — 2,048 here, 10^5 in bench.py — and
    from then on every rotation merges rows instead of adding them.  Step by step on the device against the step-by-step oracle: the term
    count and the ROW ORDER at saturation, the coefficients within 1e-12 after 120 rotations."""
    rng = np.random.default_rng(97)
    n, seed_terms = 40, 8
    symp = rng.random((seed_terms, 2 * n)) < 0.3
    c = rng.standard_normal(seed_terms) + 1j * rng.standard_normal(seed_terms)
    gens = [rng.random(2 * n) < 0.5 for _ in range(8)]
    angles = [0.3, -1.1, 0.7, 2.0, 0.3, 0.45, -0.2, 1.3]
    rots = [(gens[k % 8], angles[k % 8]) for k in range(120)]
    P = PauliwordOp(symp, c)
    cur, counts = P, []
    es, ec = symp, c
    for q, a in rots:
        cur = cur._rotate_by_single_Pword(PauliwordOp(q.reshape(1, -1), [1]), a)
        es, ec = onp.rotate_by_single_pword(es, ec, q, a)
        counts.append(cur.n_terms)
        assert cur.n_terms == es.shape[0]
    assert counts[-1] == counts[-9] == counts[40] and 500 <= counts[-1] <= seed_terms * 256, counts[::8]       # saturated long before the end
    assert_op_equal(cur.symp_matrix, cur.coeff_vec, es, ec, exact=False, tol=TOL)
    chained = P.perform_rotations([(PauliwordOp(q.reshape(1, -1), [1]), a) for q, a in rots])
    er, erc = onp.perform_rotations(symp, c, rots)
    assert_op_equal(chained.symp_matrix, chained.coeff_vec, er, erc, exact=False, tol=TOL)


@pytest.mark.parametrize('n,T,K', [(1000, 1, 300), (70, 37, 200), (5, 60, 120), (130, 1500, 60), (64, 128, 30), (64, 129, 12), (64, 448, 30), (64, 1537, 12), (100, 4096, 9), (100, 4097, 7), (1000, 3000, 11), (2000, 700, 9), (4096, 300, 9), (1, 4, 50),
                                   (4096, 60, 15), (2000, 128, 10), (2048, 127, 10), (1000, 128, 12), (1, 128, 20),       # the LDS-resident kernel at its limits
                                   (200, 20000, 25)])
def test_clifford_chain_single_launch_vs_oracle(n, T, K):
    """perform_rotations with runs of Clifford rotations on a small clean operator = ONE launch per run (symgpu_rotate_clifford_chain_dev):
    rows, row order and coefficients (exact phases: bit-exact) against the step-by-step oracle, every k in -2..5, rotations that
    commute with everything, a non-Clifford rotation in the middle (splits the run), 128 / 129 / 4096 / 4097 / 20000 terms (single-workgroup, two-launch, four-launch form;
    launch / back-to-back multi-workgroup launches)."""
    rng = np.random.default_rng(5000 + n + T)
    symp = rng.random((T, 2 * n)) < (0.3 if n > 1 else 0.5)
    P = PauliwordOp(symp, dyadic(rng, T)).cleanup()
    rots = []
    for j in range(K):
        q = rng.random(2 * n) < 0.4
        if j % 17 == 5:
            q[:] = False                                           # identity: commutes with everything
        rots.append((q, float(rng.integers(-2, 6)) * np.pi / 2))
    if K > 100 and P.n_terms < 500:
        rots[K // 2] = (rots[K // 2][0], 0.3)                      # one non-Clifford step: the run is split around it
    R = P.perform_rotations([(PauliwordOp(q.reshape(1, -1), [1]), a) for q, a in rots])
    er, ec = onp.perform_rotations(P.symp_matrix, P.coeff_vec, rots)
    exact = not any(abs(a - 0.3) < 1e-12 for _, a in rots)
    assert_op_equal(R.symp_matrix, R.coeff_vec, er, ec, exact=exact, tol=TOL)


@pytest.mark.parametrize('local', ['8192', '0'])
def test_clifford_chain_kernel_at_its_row_limit(local, monkeypatch):
    """8192 rows: the single-workgroup chain kernel at its limit (SYMGPU_CHAIN_LOCAL_T=8192; by default it is used up to 128 rows)
    and the back-to-back multi-workgroup chain (=0) against the one-by-one path."""
    from symmer_amd.kernels import DeviceOp
    monkeypatch.setenv('SYMGPU_CHAIN_LOCAL_T', local)
    rng = np.random.default_rng(8)
    n, T, K = 100, kernels.CLIFFORD_CHAIN_KERNEL_LIMIT, 9
    P = PauliwordOp(rng.random((T, 2 * n)) < 0.3, dyadic(rng, T)).cleanup()
    dev = kernels.cleanup_dev(DeviceOp.upload(P.packed, P.coeff_vec))
    assert dev.n_terms <= T
    qs = packing.pack_rows(rng.random((K, 2 * n)) < 0.3)
    ks = np.array([1, 3, 2, 0, 1, 1, 3, 2, 1], dtype=np.int32)
    out = kernels.rotate_clifford_chain_dev(dev, qs, ks)
    cur = dev
    for j in range(K):
        res, allc = kernels.rotate_single_dev(cur, qs[j], float(ks[j]) * np.pi / 2)
        if not allc:
            if cur is not dev:
                cur.free()
            cur = res
    r0, c0 = out.download(); r1, c1 = cur.download()
    assert np.array_equal(r0, r1) and np.array_equal(c0, c1)
    for h in {id(dev): dev, id(out): out, id(cur): cur}.values():
        h.free()


def test_clifford_chain_abi_refuses_unclean_or_large_operators():
    from symmer_amd import _lib
    from symmer_amd.kernels import DeviceOp
    rng = np.random.default_rng(3)
    rows = packing.pack_rows(rng.random((10, 20)) < 0.5)
    raw = DeviceOp.upload(rows, np.ones(10, dtype=complex))          # not from a cleanup: duplicate status unknown
    q = packing.pack_rows(rng.random((2, 20)) < 0.5)
    with pytest.raises(_lib.SymgpuError):
        kernels.rotate_clifford_chain_dev(raw, q, [1, 3])
    clean = kernels.cleanup_dev(raw)
    with pytest.raises(_lib.SymgpuError):
        kernels.rotate_clifford_chain_dev(clean, q, [1, 7])           # k outside 0..3
    out = kernels.rotate_clifford_chain_dev(clean, q[:0], np.zeros(0, dtype=np.int32))    # empty chain: a copy
    r0, c0 = clean.download(); r1, c1 = out.download()
    assert np.array_equal(r0, r1) and np.array_equal(c0, c1)
    for h in (raw, clean, out):
        h.free()


def test_large_angle_warning_only_when_the_rotation_acts():
    """base.py:1156-1157: warned on the non-Clifford branch only, i.e. not when every term commutes with the rotation axis."""
    import warnings as w
    P = PauliwordOp.from_list(['ZZI', 'IZZ'], [1, 2])
    with w.catch_warnings(record=True) as rec:
        w.simplefilter('always')
        assert P._rotate_by_single_Pword(PauliwordOp.from_list(['ZII']), 1e7 + 0.3) is P
    assert not any('Large angle' in str(x.message) for x in rec)
    with w.catch_warnings(record=True) as rec:
        w.simplefilter('always')
        P._rotate_by_single_Pword(PauliwordOp.from_list(['XII']), 1e7 + 0.3)
    assert any('Large angle' in str(x.message) for x in rec)


@pytest.mark.parametrize('case', family('sector'))
def test_update_sector_with_quantum_state_golden(case):
    """IndependentOp.update_sector with a QuantumState reference (basis states, dominated and balanced superpositions):
    the reference's assignments incl. the zeros (independent_op.py:275-301, :364-383); an unnormalised state is refused."""
    from symmer_amd.operators import QuantumState
    import warnings as w
    G = IndependentOp(as_bool(case['stab_symp']), np.ones(case['stab_symp'].shape[0], dtype=int))
    psi = QuantumState(case['state_matrix'].astype(int), case['state_coeff'])
    with w.catch_warnings(record=True) as rec:
        w.simplefilter('always')
        G.update_sector(psi)
    assert np.array_equal(G.coeff_vec, case['sector'])
    assert any('assigned zero values' in str(x.message) for x in rec) == bool(np.any(case['sector'] == 0))
    if psi.n_terms == 1:                                          # a basis state: the bit-array form gives the same sector
        G2 = IndependentOp(as_bool(case['stab_symp']), np.ones(case['stab_symp'].shape[0], dtype=int))
        with w.catch_warnings():
            w.simplefilter('ignore')
            G2.update_sector(case['state_matrix'][0])
        assert np.array_equal(G2.coeff_vec, case['sector'])
    with pytest.raises(AssertionError):
        G.update_sector(QuantumState(case['state_matrix'].astype(int), 2 * case['state_coeff']))


@pytest.mark.parametrize('case', family('gf2'))
def test_gf2_golden(case):
    m = unpackbits_matrix(case['m'], case['shape'])
    R, C = m.shape
    assert np.array_equal(_rref_binary(m), unpackbits_matrix(case['rref_noswap'], (R, C)))
    assert np.array_equal(rref_binary(m), unpackbits_matrix(case['rref'], (R, C)))
    assert np.array_equal(_cref_binary(m), unpackbits_matrix(case['cref_noswap'], (R, C)))
    assert np.array_equal(cref_binary(m), unpackbits_matrix(case['cref'], (R, C)))
    _, n_xor = kernels.rref(packing.pack_bits(m))
    assert n_xor == onp.rref_noswap(m, count_xors=True)[1]                                 # reference-order XOR count


@pytest.mark.parametrize('case', family('symgen'))
def test_symgen_golden(case):
    H = PauliwordOp(as_bool(case['h_symp']), np.ones(case['h_symp'].shape[0]))
    S = IndependentOp.symmetry_generators(H, commuting_override=True)
    assert np.array_equal(S.symp_matrix, as_bool(case['symgen'])) and S.coeff_vec.dtype.kind == 'i'
    G = H.generators
    assert np.array_equal(G.symp_matrix, as_bool(case['gens']))
    Rm, mask = H.generator_reconstruction(G)
    assert np.array_equal(Rm, case['recon']) and np.array_equal(mask, as_bool(case['recon_mask']))


def test_independent_op_known():
    k = known()
    H2 = PauliwordOp(as_bool(k['H2_symp']), k['H2_coeff'])
    G1 = IndependentOp.symmetry_generators(H2)                                             # test_independent_op.py:97-104
    assert np.array_equal(G1.symp_matrix, as_bool(k['H2_symgen']))
    G2 = IndependentOp.from_list(['ZIZI', 'IZIZ', 'IIZZ'])
    assert np.all(G1.generator_reconstruction(G2)[1]) and np.all(G2.generator_reconstruction(G1)[1])
    with pytest.warns(UserWarning):
        assert IndependentOp.symmetry_generators(PauliwordOp.from_list(['X', 'Y', 'Z'])).n_terms == 0   # :50-53
    op = PauliwordOp.from_list(['IZZ', 'ZZI', 'IXX', 'XXI', 'IYY', 'YYI'])
    with_override = IndependentOp.symmetry_generators(op, commuting_override=True)
    without = IndependentOp.symmetry_generators(op, commuting_override=False)                 # :56-66
    assert with_override == IndependentOp.from_list(['XXX', 'ZZZ']) and with_override != without
    assert without in [IndependentOp.from_list(['XXX']), IndependentOp.from_list(['ZZZ'])]
    with pytest.warns():
        IndependentOp.symmetry_generators(PauliwordOp.from_list(['Z' * 20]))               # :68-71 (greedy clique cover)
    with pytest.raises(ValueError):
        IndependentOp.from_list(['X', 'Y', 'Z'])                                           # :73-75
    with pytest.raises(ValueError):
        IndependentOp([[0, 1]], [1], target_sqp='x')                                       # :27-29
    with pytest.raises(ValueError):
        IndependentOp.from_list(['XZ'], [2])                                               # :117-119
    assert IndependentOp.from_list(['X', 'Z']) == IndependentOp([[0, 1], [1, 0]], [1, 1])
    G = IndependentOp.from_list(['IZ', 'ZI', 'XX'])
    assert G[2] == IndependentOp.from_list(['XX'])


# ---------------------------------------------------------------- oracle parity on seeded inputs -----------
@pytest.mark.parametrize('n,N,M', [(1000, 3000, 2500), (2000, 700, 1100), (100, 513, 257), (1, 70, 300), (64, 1, 1000), (4097, 40, 33)])
def test_commutes_vs_oracle(n, N, M):
    rng = np.random.default_rng(100 + n)
    a = packing.pack_rows(rng.random((N, 2 * n)) < 0.3); b = packing.pack_rows(rng.random((M, 2 * n)) < 0.3)
    assert np.array_equal(kernels.commutes(a, b), oc.commutes(a, b))
    assert np.array_equal(kernels.commutes(a, a), oc.commutes(a, a))


def test_matmul_gf2_vs_oracle():
    rng = np.random.default_rng(5)
    A = rng.random((70, 130)) < 0.5; B = rng.random((130, 45)) < 0.5
    assert np.array_equal(matmul_GF2(A, B), onp.matmul_gf2(A, B))
    from symmer_amd.operators import numba_binary_matmal_GF2, numba_dot_matmal_GF2
    assert np.array_equal(numba_binary_matmal_GF2(A, B), onp.matmul_gf2(A, B)) and np.array_equal(numba_dot_matmal_GF2(A, B), onp.matmul_gf2(A, B))


@pytest.mark.parametrize('fused', ['1', '0'])
@pytest.mark.parametrize('n,Ni,No,left', [(1000, 700, 300, True), (1000, 300, 700, False), (100, 500, 500, True), (130, 1, 77, True),
                                           (2000, 257, 9, False), (63, 1000, 3, True),
                                           # every row length of the phase-byte row stream (16-byte chunks per row 1, 2, 4, 8, 16, 32, 64), both
                                           # orientations, term counts that do not fill the last block / wave / row group
                                           (1, 1001, 5, True), (64, 257, 33, False), (65, 129, 7, False), (128, 1, 1, True), (200, 333, 17, True),
                                           (256, 65, 31, False), (449, 99, 20, True), (512, 31, 64, False), (961, 17, 5, False), (1024, 1023, 3, True),
                                           (1985, 9, 40, True), (2048, 130, 2, False), (4033, 5, 9, False), (4096, 67, 4, True),
                                           (300, 100, 7, True), (3000, 11, 13, False)])       # 5 and 47 chunks per row: the two-kernel path
def test_mul_allpairs_vs_oracle(n, Ni, No, left, fused, monkeypatch):
    """symgpu_mul_allpairs on both product paths: SYMGPU_PRODUCT_FUSED=1 (default: row stream + phase bytes + expansion where the
    row length is a power-of-two number of chunks) and =0 (word-major coefficient kernel + plain row stream)."""
    monkeypatch.setenv('SYMGPU_PRODUCT_FUSED', fused)
    rng = np.random.default_rng(200 + n + Ni)
    a = packing.pack_rows(rng.random((Ni, 2 * n)) < 0.3); b = packing.pack_rows(rng.random((No, 2 * n)) < 0.3)
    ca, cb = dyadic(rng, Ni), dyadic(rng, No)
    rows, coeff = kernels.mul_allpairs(a, ca, b, cb, left)
    erows, ecoeff = oc.mul_allpairs(a, ca, b, cb, left)
    assert np.array_equal(rows, erows) and np.array_equal(coeff, ecoeff)
    cg, cbg = rng.standard_normal(Ni) + 1j * rng.standard_normal(Ni), rng.standard_normal(No) + 1j * rng.standard_normal(No)
    _, coeff = kernels.mul_allpairs(a, cg, b, cbg, left)
    _, ecoeff = oc.mul_allpairs(a, cg, b, cbg, left)
    assert np.array_equal(coeff, ecoeff)        # both sides are the un-fused IEEE expression: bit-exact even for Gaussian input


@pytest.mark.parametrize('order', ['0', '1'])
def test_fused_output_stage_block_orders(order, monkeypatch):
    """The fused output stage of the cleanup writes its rows in one of two block orders (every XCD a contiguous eighth of the output, or
    the plain interleaved order) and keeps whichever it measured faster on the buffer at hand (cleanup.hip emit_order_begin);
    SYMGPU_EMIT_ORDER pins one.  1,500 terms squared is above the 2^20-key gate of the lazy path that ends in that stage."""
    monkeypatch.setenv('SYMGPU_EMIT_ORDER', order)
    rng = np.random.default_rng(611)
    n, N = 100, 1500
    A = PauliwordOp(rng.random((N, 2 * n)) < 0.3, dyadic(rng, N))
    R = A * A
    erows, ecoeff = oc.mul(A.packed, A.coeff_vec, A.packed, A.coeff_vec)
    assert np.array_equal(R.packed, erows) and np.array_equal(R.coeff_vec, ecoeff)


@pytest.mark.parametrize('fused', ['1', '0'])
@pytest.mark.parametrize('n,Ni,No,left', [(1000, 5000, 9, True), (2000, 3001, 5, False), (3000, 2100, 4, True), (100, 70000, 3, False), (20, 200001, 2, True)])
def test_mul_allpairs_inner_operand_in_tiles(n, Ni, No, left, fused, monkeypatch):
    """An inner operand whose eighth does not fit an XCD's L2 is streamed tile by tile (product.hip inner_tile_chunks: 10^5 terms of 2,000
    qubits ran at 0.47 of the HBM peak untiled).  SYMGPU_PRODUCT_TILE_MB forces small tiles here: several tiles, a ragged last one, both
    row streams (phase-byte stream and plain stream + word-major coefficients), rows whose chunk count is not a power of two."""
    monkeypatch.setenv('SYMGPU_PRODUCT_FUSED', fused)
    monkeypatch.setenv('SYMGPU_PRODUCT_TILE_MB', '0.3')
    rng = np.random.default_rng(210 + n + Ni)
    a = packing.pack_rows(rng.random((Ni, 2 * n)) < 0.3); b = packing.pack_rows(rng.random((No, 2 * n)) < 0.3)
    ca, cb = dyadic(rng, Ni), dyadic(rng, No)
    rows, coeff = kernels.mul_allpairs(a, ca, b, cb, left)
    erows, ecoeff = oc.mul_allpairs(a, ca, b, cb, left)
    assert np.array_equal(rows, erows) and np.array_equal(coeff, ecoeff)


@pytest.mark.parametrize('n,N,M', [(100, 500, 500), (1000, 300, 200), (3, 700, 700), (65, 64, 900), (5000, 90, 60), (10, 70000, 3),
                                   (40, 1, 5000), (40, 5000, 1), (2, 9, 9)])
def test_mul_cleanup_vs_oracle(n, N, M):
    """cfg 1 (100 qubits, 500 terms squared: 250k pairs -> 124,751 unique) plus wider/narrower cases, bit-exact."""
    rng = np.random.default_rng(300 + n)
    A = PauliwordOp(rng.random((N, 2 * n)) < 0.3, dyadic(rng, N))
    B = A if N == M else PauliwordOp(rng.random((M, 2 * n)) < 0.3, dyadic(rng, M))
    R = A * B
    erows, ecoeff = oc.mul(A.packed, A.coeff_vec, B.packed, B.coeff_vec)
    assert np.array_equal(R.packed, erows) and np.array_equal(R.coeff_vec, ecoeff)
    if n == 100:
        rows_nothr, _ = kernels.mul_cleanup(A.packed, A.coeff_vec, A.packed, A.coeff_vec, True, None)
        assert rows_nothr.shape[0] == 1 + N * (N - 1) // 2                                # 124,751


@pytest.mark.parametrize('n,N', [(100, 500), (3, 700), (1, 50), (70, 1), (1000, 257), (12, 3000)])
def test_squared_operator_path_equals_general_pair_path(n, N, monkeypatch):
    """P * P with one device operand sorts only the pairs with i >= o (half of the pairs; the twin (o, i) has the same coefficient if
    the terms commute and the opposite one if they anticommute, so the pair is weighted 2 or 0 — a zero-weight pair still fixes the
    first-occurrence position of its row, which small n exercises: n = 1, 3 have 4 / 64 distinct rows) — same rows, same order as
    the general pair path (SYMGPU_CLEANUP_NOSQUARE=1) and as the oracle; coefficients bit-exact for dyadic inputs; Gaussian
    coefficients within the tolerance rule (sums associate twin-first)."""
    rng = np.random.default_rng(900 + n + N)
    A = PauliwordOp(rng.random((N, 2 * n)) < 0.3, dyadic(rng, N))
    for thr in (1e-15, 0.0, None):                                # None keeps zero sums: the library must not take the shortcut there
        fast = kernels.mul_cleanup(A.packed, A.coeff_vec, A.packed, A.coeff_vec, True, thr)
        monkeypatch.setenv('SYMGPU_CLEANUP_NOSQUARE', '1')
        slow = kernels.mul_cleanup(A.packed, A.coeff_vec, A.packed, A.coeff_vec, True, thr)
        monkeypatch.delenv('SYMGPU_CLEANUP_NOSQUARE')
        assert np.array_equal(fast[0], slow[0]) and np.array_equal(fast[1], slow[1])
    erows, ecoeff = oc.mul(A.packed, A.coeff_vec, A.packed, A.coeff_vec)
    R = A * A
    assert np.array_equal(R.packed, erows) and np.array_equal(R.coeff_vec, ecoeff)
    G = PauliwordOp(A.symp_matrix, rng.standard_normal(N) + 1j * rng.standard_normal(N))
    fast = G * G
    monkeypatch.setenv('SYMGPU_CLEANUP_NOSQUARE', '1')
    slow = G * G
    monkeypatch.delenv('SYMGPU_CLEANUP_NOSQUARE')
    # a row of the n = 1, 3 cases is the sum of thousands of O(1) products (N^2 / 4^n pairs per row): the two association orders
    # differ by the rounding of such a sum, ~ terms x 1.1e-16 x partial sums, so the 1e-12 bar is scaled by the terms per row
    per_row = max(1.0, N * N / 4.0 ** min(n, 30))
    assert_op_equal(fast.symp_matrix, fast.coeff_vec, slow.symp_matrix, slow.coeff_vec, exact=False, tol=TOL * per_row)


def test_read_backs_through_mapped_memory_equal_plain_copies(monkeypatch):
    """Counts and status words that a call needs on the host in the middle of its work travel through mapped host memory (a one-wavefront
    kernel + a polled sequence number, context.hip); SYMGPU_READBACK_PLAIN=1 makes every one of them a copy + stream synchronisation.
    Product + cleanup, a rotation with duplicate rows (multi-launch path), GF(2) elimination, a projection: same results both ways."""
    rng = np.random.default_rng(77)
    n = 40
    A = PauliwordOp(rng.random((700, 2 * n)) < 0.3, dyadic(rng, 700))
    D = PauliwordOp(np.vstack([A.symp_matrix[:300], A.symp_matrix[:300]]), dyadic(rng, 600))           # duplicate rows
    Q = PauliwordOp(rng.random((1, 2 * n)) < 0.5, [1])
    M = rng.random((90, 300)) < 0.3

    def run():
        R = A * A
        S = D._rotate_by_single_Pword(Q, 0.37)
        return R.packed, R.coeff_vec, S.packed, S.coeff_vec, rref_binary(M)

    fast = run()
    monkeypatch.setenv('SYMGPU_READBACK_PLAIN', '1')
    plain = run()
    monkeypatch.delenv('SYMGPU_READBACK_PLAIN')
    for x, y in zip(fast, plain):
        assert np.array_equal(x, y)
    from symmer_amd import _lib
    assert all('read-back' not in d for d in _lib.degraded())


@pytest.mark.parametrize('planted', [False, True])
def test_singles_kept_without_looking_equals_looking(planted, monkeypatch):
    """k_mark_singles skips the coefficient arithmetic when the operands' smallest coefficients prove that every pair of non-zero weight
    clears the threshold (k_coeff_floor); SYMGPU_CLEANUP_NOFLOOR=1 makes it look at every coefficient.  Same bits either way — with O(1)
    coefficients (shortcut taken) and with planted tiny ones whose products fall under the threshold (shortcut refused by itself)."""
    rng = np.random.default_rng(4242 + planted)
    n, N, M = 64, 900, 700
    ca = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    cb = rng.standard_normal(M) + 1j * rng.standard_normal(M)
    if planted:
        ca[rng.integers(0, N, 40)] *= 1e-9
        cb[rng.integers(0, M, 40)] *= 1e-9
    A = PauliwordOp(rng.random((N, 2 * n)) < 0.3, ca)
    B = PauliwordOp(rng.random((M, 2 * n)) < 0.3, cb)
    monkeypatch.setenv('SYMGPU_CLEANUP_LAZY', '1')                      # the flow with k_mark_singles at this size
    for X, Y in ((A, A), (A, B)):
        for thr in (1e-15, 1e-12, None):
            fast = kernels.mul_cleanup(X.packed, X.coeff_vec, Y.packed, Y.coeff_vec, True, thr)
            monkeypatch.setenv('SYMGPU_CLEANUP_NOFLOOR', '1')
            slow = kernels.mul_cleanup(X.packed, X.coeff_vec, Y.packed, Y.coeff_vec, True, thr)
            monkeypatch.delenv('SYMGPU_CLEANUP_NOFLOOR')
            assert np.array_equal(fast[0], slow[0]) and np.array_equal(fast[1].view(np.uint64), slow[1].view(np.uint64))
        erows, ecoeff = onp.mul(X.symp_matrix, X.coeff_vec, Y.symp_matrix, Y.coeff_vec)
        R = X * Y
        assert_op_equal(R.symp_matrix, R.coeff_vec, erows, ecoeff, exact=False)


def test_mul_cleanup_gaussian_tolerance():
    rng = np.random.default_rng(11)
    n, N = 100, 300
    A = PauliwordOp(rng.random((N, 2 * n)) < 0.3, rng.standard_normal(N) + 1j * rng.standard_normal(N))
    R = A * A
    erows, ecoeff = onp.mul(A.symp_matrix, A.coeff_vec, A.symp_matrix, A.coeff_vec)
    assert_op_equal(R.symp_matrix, R.coeff_vec, erows, ecoeff, exact=False, tol=TOL)


@pytest.mark.parametrize('T,n,dup', [(200000, 1000, 0.5), (50000, 100, 0.9), (100000, 3, 1.0), (4097, 2000, 0.2), (3000, 5000, 0.6), (1, 70, 0.0),
                                     (65, 64, 0.5),
                                     (9_000_000, 40, 0.45)])           # > 2 x 4M input indices: three batches of the output stage
def test_cleanup_vs_oracle(T, n, dup):
    rng = np.random.default_rng(400 + n)
    base = packing.pack_rows(rng.random((max(1, int(T * (1 - dup)) + 1), 2 * n)) < 0.3)
    rows = base[rng.integers(0, base.shape[0], T)]
    coeff = dyadic(rng, T)
    for thr in (1e-15, None):
        r, c = kernels.cleanup(rows, coeff, thr)
        er, ec = oc.cleanup(rows, coeff, thr)
        assert np.array_equal(r, er) and np.array_equal(c, ec)


@pytest.mark.parametrize('R,C,dens', [(1000, 3000, 0.5), (300, 20000, 0.01), (2000, 500, 0.5), (33, 64, 0.5), (700, 700, 0.003)])
def test_rref_vs_oracle(R, C, dens):
    rng = np.random.default_rng(500 + R)
    m = rng.random((R, C)) < dens
    m[R // 3] = False
    m[R - 1] = m[0]
    p = packing.pack_bits(m)
    red, n_xor, piv = kernels.rref(p, want_pivots=True)
    ered, en_xor, epiv = oc.rref(p, want_pivots=True)
    assert np.array_equal(red, ered) and n_xor == en_xor and np.array_equal(piv, epiv)


def test_rref_recovers_from_an_in_launch_time_out(monkeypatch):
    """The fused GF(2) schedule waits inside a launch for flags of other workgroups; should that wait give up (workgroups not co-resident)
    the half-updated matrix is restored from the copy taken at entry and reduced with separate launches (robustness).  The
    time-out is injected after the first attempt: reduced matrix, pivots and reference row-XOR count must still be the oracle's."""
    rng = np.random.default_rng(321)
    m = rng.random((900, 5000)) < 0.4
    p = packing.pack_bits(m)
    monkeypatch.setenv('SYMGPU_GF2_FUSED_SELECT', '2')                     # 2: the first attempt is treated as timed out
    red, n_xor, piv = kernels.rref(p, want_pivots=True)
    ered, en_xor, epiv = oc.rref(p, want_pivots=True)
    assert np.array_equal(red, ered) and n_xor == en_xor and np.array_equal(piv, epiv)
    monkeypatch.delenv('SYMGPU_GF2_FUSED_SELECT')
    red2, n2 = kernels.rref(p)
    assert np.array_equal(red2, ered) and n2 == en_xor


def test_rref_wide_rows_global_panel():
    """rows wider than the LDS budget take the global-memory panel path"""
    rng = np.random.default_rng(9)
    m = rng.random((40, 64 * 20000)) < 0.001
    p = packing.pack_bits(m)
    red, n_xor = kernels.rref(p)
    ered, en_xor = oc.rref(p)
    assert np.array_equal(red, ered) and n_xor == en_xor


@pytest.mark.parametrize('n,M,k', [(300, 2000, 12), (1000, 5000, 20), (64, 64, 3)])
def test_symmetry_kernel_vs_oracle(n, M, k):
    rng = np.random.default_rng(600 + n)
    symp = rng.random((M, 2 * n)) < 0.3
    symp[:, :k] = False                                   # planted: Z_0..Z_{k-1} commute with every term
    H = PauliwordOp(symp, np.ones(M))
    for _ in range(4):                                    # scramble with Clifford rotations (symmer/utils.py:141-149)
        H = H._rotate_by_single_Pword(PauliwordOp((rng.random(2 * n) < 0.3).reshape(1, -1), [1]), np.pi / 2)
    rows, n_xor = kernels.symmetry_kernel(H.packed, n)
    erows, en_xor = oc.symmetry_generators(H.packed, n)
    assert np.array_equal(rows, erows) and n_xor == en_xor
    assert rows.shape[0] == (k if M >= 4 * n else max(k, 2 * n - M))     # generic rank: 2n - (M terms' rank)
    S = IndependentOp.symmetry_generators(H, commuting_override=True)
    assert np.all(S.commutes_termwise(H))


@pytest.mark.parametrize('n,T', [(1000, 20000), (130, 5000),
                                 # every row length of the chunk-per-lane analysis kernel (1, 2, 4, 8, 16, 32, 64 chunks per row) and two others
                                 (40, 3000), (64, 777), (100, 2100), (256, 1500), (449, 900), (1024, 1300), (2000, 700), (4096, 300), (300, 800), (3000, 300)])
def test_rotation_vs_oracle(n, T):
    rng = np.random.default_rng(700 + n)
    symp, c = onp.cleanup_op(rng.random((T, 2 * n)) < 0.3, dyadic(rng, T))
    q = rng.random(2 * n) < 0.3
    half = symp.shape[0] // 2
    symp, c = onp.cleanup_op(np.vstack([symp, symp[:half] ^ q]), np.hstack([c, dyadic(rng, half)]))   # force P / P*Q merges
    P = PauliwordOp(symp, c); Q = PauliwordOp(q.reshape(1, -1), [1])
    for ang in (0.3, np.pi / 2, np.pi, -np.pi / 2, 3 * np.pi / 2):
        R = P._rotate_by_single_Pword(Q, ang)
        er, ec = onp.rotate_by_single_pword(symp, c, q, ang)
        clifford = abs(round(2 * ang / np.pi) - 2 * ang / np.pi) <= 1e-18
        assert_op_equal(R.symp_matrix, R.coeff_vec, er, ec, exact=clifford, tol=TOL)
    rots = [(PauliwordOp((rng.random(2 * n) < 0.3).reshape(1, -1), [1]), a) for a in (np.pi / 2, 0.4, np.pi / 2, -1.3)]
    R = P.perform_rotations(rots)
    er, ec = onp.perform_rotations(symp, c, [(r.symp_matrix[0], a) for r, a in rots])
    assert_op_equal(R.symp_matrix, R.coeff_vec, er, ec, exact=False, tol=TOL)


def test_rccl_allgather_single_rank():
    """The RCCL data plane with ONE rank (the GPU box has one device): unique id, communicator, all-gather of packed
    rows + coefficients into a larger operator, barrier, destroy — everything but the peer traffic."""
    import ctypes
    from symmer_amd import _lib
    from symmer_amd.kernels import DeviceOp
    lib = _lib.lib()
    ident = (ctypes.c_uint8 * 128)()
    _lib.check(lib.symgpu_comm_unique_id(ctypes.addressof(ident)))
    _lib.check(lib.symgpu_comm_init(ctypes.addressof(ident), 0, 1))
    try:
        rng = np.random.default_rng(3)
        rows = packing.pack_rows(rng.random((1000, 260)) < 0.3); coeff = dyadic(rng, 1000)
        shard = DeviceOp.upload(rows, coeff)
        shard.set_rows(777)                                    # tail of the shard is padding: must come out zeroed
        full = DeviceOp.alloc(1000, rows.shape[1] // 2, with_coeff=True)
        _lib.check(lib.symgpu_comm_allgather_op(shard.handle, full.handle))
        assert full.n_terms == 1000
        r, c = full.download()
        assert np.array_equal(r[:777], rows[:777]) and np.array_equal(c[:777], coeff[:777])
        assert not r[777:].any() and not c[777:].any()
        _lib.check(lib.symgpu_comm_barrier())
    finally:
        _lib.check(lib.symgpu_comm_destroy())


def test_rotation_input_with_duplicate_rows_takes_general_path():
    """The hash-join fast path assumes distinct rows; duplicates are detected on the device and routed to the sort-based path.
    Result must still equal the reference's step-by-step evaluation (dyadic coefficients, cos/sin irrational -> 1e-12)."""
    rng = np.random.default_rng(77)
    n, T = 70, 300
    symp = rng.random((T, 2 * n)) < 0.3
    q = rng.random(2 * n) < 0.4
    symp = np.vstack([symp, symp[:50], symp[:30] ^ q, symp[10:20]])          # duplicates and P*Q partners
    c = dyadic(rng, symp.shape[0])
    P = PauliwordOp(symp, c); Q = PauliwordOp(q.reshape(1, -1), [1])
    for ang in (0.3, -2.2):
        R = P._rotate_by_single_Pword(Q, ang)
        er, ec = onp.rotate_by_single_pword(symp, c, q, ang)
        assert_op_equal(R.symp_matrix, R.coeff_vec, er, ec, exact=False, tol=TOL)


# ---------------------------------------------------------------- tapering workflow ------
def test_is_noncontextual_known():
    import json, os
    from _golden import GOLDEN
    with open(os.path.join(GOLDEN, 'known_noncontextual.json')) as f:
        for plist, expect in json.load(f):                                                 # test_base.py:581-595
            assert PauliwordOp.from_list(plist).is_noncontextual == expect


@pytest.mark.parametrize('case', family('taper'))
def test_qubit_tapering_golden(case):
    """QubitTapering on the reference's molecular Hamiltonian fixtures (tests/hamiltonian_data/*.json): same symmetry
    generators, sector, Clifford rotations, rotated stabilisers, free qubits and tapered operator as the reference."""
    from symmer_amd.projection import QubitTapering
    H = PauliwordOp(as_bool(case['h_symp']), case['h_coeff'])
    T = QubitTapering(H, target_sqp=chr(int(case['target'])))
    assert np.array_equal(T.symmetry_generators.symp_matrix, as_bool(case['stab_symp']))
    if case['hf'].shape[0]:
        out = T.taper_it(ref_state=case['hf'])
    else:
        out = T.taper_it(sector=case['stab_coeff'])
    assert np.array_equal(T.stabilizers.coeff_vec, case['stab_coeff'])
    rots = T.stabilizers.stabilizer_rotations
    got_rots = np.vstack([r.symp_matrix for r, _ in rots]) if rots else np.zeros((0, 2 * H.n_qubits), dtype=bool)
    assert np.array_equal(got_rots, as_bool(case['rotations']))
    assert np.array_equal(T.rotated_stabilizers.symp_matrix, as_bool(case['rotated_stab']))
    assert np.array_equal(T.rotated_stabilizers.coeff_vec, case['rotated_stab_coeff'])
    assert np.array_equal(T.free_qubit_indices, case['free'])
    assert_op_equal(out.symp_matrix, out.coeff_vec, case['out_symp'], case['out_coeff'], exact=False, tol=TOL)
    assert out.n_qubits == H.n_qubits - T.n_taper


def test_f3_f4_device_paths_vs_host_restatements():
    """SURVEY 8f rows f3 / f4 on the device (csrc/project.hip) against NumPy restatements of the reference lines, random inputs:
    projection (projection/base.py:60-84), noncontextuality test (utils.py:567-589 on the oracle's adjacency matrix), bra * ket
    (base.py:1808-1815) — and none of them builds the one-byte-per-bit matrix of its operand."""
    from symmer_amd.projection.base import S3Projection
    from symmer_amd import QuantumState
    rng = np.random.default_rng(808)
    # ---- projection: random operator, stabilisers = single-qubit Z / X on chosen qubits with random sector
    for n, T, stab_q in ((70, 400, [3, 64, 69]), (10, 50, [0, 1, 2, 3, 4, 5, 6, 7, 8, 9]), (130, 300, [5]), (6, 40, [1, 4])):
        symp = rng.random((T, 2 * n)) < 0.3
        symp = np.vstack([symp, symp[: T // 4]])                                        # duplicates that only differ on stabilised qubits appear too
        symp[T:, stab_q[0]] ^= True
        coeff = dyadic(rng, symp.shape[0])
        stab = np.zeros((len(stab_q), 2 * n), dtype=bool)
        kinds = rng.integers(0, 2, len(stab_q))
        for s, (q, kd) in enumerate(zip(stab_q, kinds)):
            stab[s, q + (n if kd else 0)] = True                                       # kd 1: Z_q, 0: X_q
        eig = rng.choice([-1, 1], len(stab_q))
        op = PauliwordOp._from_packed(packing.pack_rows(symp), n, coeff)
        proj = S3Projection(IndependentOp(stab, eig))
        proj.rotated_stabilizers = PauliwordOp(stab, eig)
        proj.free_qubit_indices = np.setdiff1d(np.arange(n), stab_q)
        proj.rotated_flag = True
        out = proj._perform_projection(op)
        assert op._symp is None, 'the projection expanded its operand'
        # the reference's lines on bool matrices
        commutes = onp.commutes_termwise(symp, stab)
        keep_rows = np.all(commutes, axis=1)
        cols = np.nonzero(stab)[1]
        ev = symp[keep_rows][:, cols] * eig
        ev[ev == 0] = 1
        w = coeff[keep_rows] * np.prod(ev, axis=1)
        free = proj.free_qubit_indices
        if free.size:
            er, ec = onp.cleanup_op(symp[keep_rows][:, np.hstack([free, free + n])], w) if keep_rows.any() else (np.zeros((1, 2 * free.size), dtype=bool), np.zeros(1, dtype=complex))
            assert_op_equal(out.symp_matrix, out.coeff_vec, er, ec)
        else:
            assert out.n_qubits == 0 and out.coeff_vec[0] == np.sum(w)
    # ---- noncontextuality: random (contextual), clique-structured (noncontextual), all commuting; the expected answers come from the ORACLE's
    # restatement, which tests/test_oracle_golden.py pins to the reference's outputs (tests/golden/noncontextual.npz) — the product's own host
    # function is not the judge here (the reference-generated cases themselves: test_noncontextual_golden below)
    def host_answer(symp):
        return onp.check_adjmat_noncontextual(onp.commutes_termwise(symp, symp))
    cases = [rng.random((300, 2 * 40)) < 0.3, rng.random((5, 2 * 3)) < 0.5]
    zs = np.zeros((200, 2 * 50), dtype=bool); zs[:, 50:] = rng.random((200, 50)) < 0.4
    cases.append(zs)                                                                     # Z strings only: everything commutes
    # two anticommuting cliques on disjoint... noncontextual set: universally commuting Z's plus {X0 A_i} and {Z0 B_i} style cliques
    nc = np.zeros((60, 2 * 30), dtype=bool)
    nc[:20, 30 + 1:] = rng.random((20, 29)) < 0.5                                        # Z on qubits 1..29: commute with all below
    nc[20:40, 30 + 1:] = rng.random((20, 29)) < 0.5; nc[20:40, 0] = True                 # X0 * Z-string: clique 1
    nc[40:, 30 + 1:] = rng.random((20, 29)) < 0.5; nc[40:, 30] = True                    # Z0 * Z-string: clique 2 (anticommutes with clique 1)
    cases.append(nc)
    for symp in cases:
        op = PauliwordOp._from_packed(packing.pack_rows(symp), symp.shape[1] // 2, np.ones(symp.shape[0]))
        assert op.is_noncontextual == host_answer(symp)
        assert op._symp is None and 'adjacency_matrix' not in op.__dict__, 'the noncontextuality test built a host matrix'
    assert host_answer(nc) and not host_answer(cases[0])
    # ---- bra * ket: states with shared and unshared basis strings, duplicates inside a state, complex amplitudes
    for nq, Na, Nb in ((70, 300, 500), (5, 20, 30), (130, 1000, 40)):
        base = rng.integers(0, 2, (Na + Nb, nq))
        a_m = np.vstack([base[:Na], base[: Na // 3]]); b_m = np.vstack([base[Na // 2:], base[Na // 2: Na // 2 + 7]])
        a_c = dyadic(rng, a_m.shape[0]); b_c = dyadic(rng, b_m.shape[0])
        bra = QuantumState(a_m, a_c, vec_type='bra'); ket = QuantumState(b_m, b_c)
        got = bra * ket
        left, right = (bra, ket) if bra.state_op.n_terms < ket.n_terms else (ket, bra)
        rd = right.to_dictionary
        expect = 0
        for bstring, lc in left.to_dictionary.items():
            expect += lc * rd.get(bstring, 0)
        assert got == expect                                                             # dyadic amplitudes: bit for bit, same order of additions


def test_f3_f4_entry_point_edge_cases():
    """C-ABI edge cases of csrc/project.hip and of the indexed cleanups: empty operands, no stabilisers, nothing survives, no shared basis
    string, threshold on the indexed product, wrong Wq."""
    from symmer_amd import _lib
    rng = np.random.default_rng(99)
    n = 70
    symp = rng.random((50, 2 * n)) < 0.3
    op = kernels.DeviceOp.upload(packing.pack_rows(symp), dyadic(rng, 50))
    # no stabilisers: everything survives, qubits 0..9 kept
    res, n_s = kernels.project_dev(op, np.zeros((0, 4), dtype='<u8'), np.zeros(0, dtype=int), np.arange(10), n)
    r, c = res.download(); res.free()
    er, ec = onp.cleanup_op(symp[:, np.hstack([np.arange(10), n + np.arange(10)])], op.download()[1])
    assert n_s == 50 and np.array_equal(packing.unpack_rows(r, 10), er) and np.array_equal(c, ec)
    # a stabiliser nothing commutes with... X_0 and Z_0 together remove every term that touches qubit 0 in either block; force all to touch it
    symp2 = symp.copy(); symp2[:, 0] = True; symp2[:, n] = False                      # X on qubit 0: anticommutes with Z_0
    op2 = kernels.DeviceOp.upload(packing.pack_rows(symp2), dyadic(rng, 50))
    z0 = np.zeros((1, 2 * n), dtype=bool); z0[0, n] = True
    res, n_s = kernels.project_dev(op2, packing.pack_rows(z0), np.array([1]), np.arange(1, n), n)
    assert n_s == 0 and res.n_terms == 0
    res.free(); op2.free()
    # empty operator
    empty = kernels.DeviceOp.upload(np.zeros((0, 2), dtype='<u8'), np.zeros(0, dtype=complex))          # 2 qubits, no terms
    res, n_s = kernels.project_dev(empty, packing.pack_rows(np.array([[False, False, True, False]])), np.array([1]), np.arange(1, 2), 2)
    assert n_s == 0 and res.n_terms == 0
    res.free()
    assert kernels.noncontextual_dev(empty) is True
    empty.free()
    with pytest.raises(_lib.SymgpuError):
        kernels.project_dev(op, np.zeros((0, 4), dtype='<u8'), np.zeros(0, dtype=int), np.arange(10), 200)       # n_qubits does not match Wq
    one = kernels.DeviceOp.upload(packing.pack_rows(symp[:1]))
    assert kernels.noncontextual_dev(one) is True
    one.free(); op.free()
    # states without a common basis string, and a state against itself
    a = kernels.cleanup_dev(kernels.DeviceOp.upload(packing.pack_rows(np.hstack([np.eye(8, dtype=bool), np.zeros((8, 8), dtype=bool)])), dyadic(rng, 8)))
    b_rows = np.hstack([~np.eye(8, dtype=bool), np.zeros((8, 8), dtype=bool)])
    b = kernels.cleanup_dev(kernels.DeviceOp.upload(packing.pack_rows(b_rows), dyadic(rng, 8)))
    assert kernels.state_inner_dev(a, b) == 0
    ra, ca = a.download()
    assert kernels.state_inner_dev(a, a) == sum((x * x for x in ca), 0)
    raw = kernels.DeviceOp.upload(ra, ca)
    with pytest.raises(_lib.SymgpuError):
        kernels.state_inner_dev(raw, a)                                                # not from a cleanup: refused
    for h in (a, b, raw):
        h.free()
    # indexed product with a threshold: kept terms only, indices still those of the first pairs
    A = onp.pack_rows(rng.random((40, 2 * 5)) < 0.4); ca = dyadic(rng, 40)
    r, c, i_f, o_f = kernels.mul_cleanup_indexed(A, ca, A, ca, True, 1e-15)
    er, ec = oc.mul(A, ca, A, ca)
    assert np.array_equal(r, er) and np.array_equal(c, ec) and np.array_equal(A[i_f] ^ A[o_f], r) and np.all(np.diff(o_f * 40 + i_f) > 0)


def test_independent_op_rotations_and_sector():
    """tests/test_operators/test_independent_op.py:77-109 (rotation onto single-qubit Z / X, sector assignment)."""
    k = known()
    op = PauliwordOp.from_list(['Z' * 20])
    for target in ('Z', 'X'):
        G = IndependentOp.symmetry_generators(op)
        G.target_sqp = target
        rotated = G.rotate_onto_single_qubit_paulis()
        blk, other = (rotated.Z_block, rotated.X_block) if target == 'Z' else (rotated.X_block, rotated.Z_block)
        assert np.all(np.sum(blk, axis=1) <= 1) and np.all(~other)
    H2 = PauliwordOp(as_bool(k['H2_symp']), k['H2_coeff'])
    G = IndependentOp.symmetry_generators(H2)
    ref_state = np.array([1, 1, 0, 0])
    G.update_sector(ref_state=ref_state)
    assert np.all(G.coeff_vec == (-1) ** np.sum(np.bitwise_and(G.Z_block, ref_state.astype(bool)), axis=1))
    with pytest.warns(UserWarning):
        X = IndependentOp.from_list(['XX', 'ZZ'])
        X.update_sector([0, 0])
    assert X.coeff_vec.tolist() == [0, 1]


@pytest.mark.parametrize('case', family('circuit'))
def test_circuit_symmerlator_golden(case):
    """random Clifford(+rotation) circuits through CircuitSymmerlator, rotated observable and <0|..|0>
    expectation value equal to the reference's (generated by oracle/tools/gen_golden.py)."""
    from symmer_amd.evolution import CircuitSymmerlator
    cs = CircuitSymmerlator(int(case['n']))
    for g, a, b, t in zip(case['gates'], case['qa'], case['qb'], case['angle']):
        g = str(g)
        if g in ('rx', 'ry', 'rz'):
            cs.gate_map[g](int(a), float(t))
        elif int(b) >= 0:
            cs.gate_map[g](int(a), int(b))
        else:
            cs.gate_map[g](int(a))
    O = PauliwordOp(as_bool(case['in_symp']), case['in_coeff'])
    R = cs.apply_sequence(O)
    assert_op_equal(R.symp_matrix, R.coeff_vec, case['out_symp'], case['out_coeff'], exact=False, tol=TOL)
    assert abs(complex(cs.evaluate(O)) - complex(case['expval'])) < 1e-10


@pytest.mark.parametrize('case', family('state'))
def test_quantum_state_golden(case):
    """operator x ket, bra x operator, inner products and expectation values against the reference."""
    from symmer_amd import QuantumState
    from symmer_amd.operators import single_term_expval
    P = PauliwordOp(as_bool(case['p_symp']), case['p_coeff'])
    psi = QuantumState(case['psi_m'], case['psi_c']); phi = QuantumState(case['phi_m'], case['phi_c'])
    assert psi._is_normalized()
    out = P * psi
    assert np.array_equal(out.state_matrix, case['out_m']) and np.allclose(out.state_op.coeff_vec, case['out_c'], rtol=0, atol=TOL)
    bra = psi.dagger * P
    assert bra.vec_type == 'bra'
    assert np.array_equal(bra.state_matrix, case['bra_m']) and np.allclose(bra.state_op.coeff_vec, case['bra_c'], rtol=0, atol=TOL)
    assert abs(complex(psi.dagger * phi) - complex(case['inner'])) < 1e-12
    assert abs(P.expval(psi) - complex(case['expval']).real) < 1e-10
    got = np.array([single_term_expval(Pk, psi) for Pk in P])
    assert np.allclose(got, case['term_expvals'], rtol=0, atol=1e-10)
    assert psi + psi == psi * 2 and (psi - psi).n_terms == 0


def test_indexed_cleanups_and_hash_partitioned_shares():
    """The cleanups that return first-occurrence indices (symgpu_*_indexed_dev) against the oracle, and the hash-partitioned multi-GPU cleanup
    (parallel.hash_partition_local) with the DEVICE kernels: the shares of G = 1, 2, 3, 4, 8 ranks, computed one after the other on this GPU
    and merged by first pair index, equal the single-process oracle result — squared operators (twins on one rank), general products in both
    operand orders, repeated rows, through and below the size gates of the lazy cleanup flow."""
    from symmer_amd import parallel
    rng = np.random.default_rng(606)
    for n, Ni, No in ((100, 700, 400), (1000, 2100, 2100), (40, 300, 5)):
        A = onp.pack_rows(rng.random((Ni, 2 * n)) < 0.3); B = onp.pack_rows(rng.random((No, 2 * n)) < 0.3)
        A[Ni // 2: Ni // 2 + 20] = A[:20]; B[No // 2:] = B[: No - No // 2]            # repeated rows in both operands
        a = dyadic(rng, Ni); b = dyadic(rng, No)
        # indexed product + cleanup: rows / sums as the plain call, indices = first occurrence in pair order
        r, c, i_f, o_f = kernels.mul_cleanup_indexed(A, a, B, b, True, None)
        pr, pc = oc.mul_allpairs(A, a, B, b, True)
        er, ec = oc.cleanup(pr, pc, None)
        assert np.array_equal(r, er) and np.array_equal(c, ec)
        assert np.array_equal(A[i_f] ^ B[o_f], r)
        g = o_f * Ni + i_f
        assert np.all(np.diff(g) > 0), 'first occurrences must come in pair order'
        first_np, _ = onp.first_occurrence_unique(np.ascontiguousarray(pr).view(np.uint8).reshape(pr.shape[0], -1))
        assert np.array_equal(g, first_np)
        # indexed plain cleanup
        stack = np.vstack([A, A[::3], B]); sc = np.hstack([a, a[::3], b])
        r2, c2, f2 = kernels.cleanup_indexed(stack, sc, 1e-15)
        e2r, e2c = oc.cleanup(stack, sc, 1e-15)
        assert np.array_equal(r2, e2r) and np.array_equal(c2, e2c) and np.array_equal(stack[f2], r2) and np.all(np.diff(f2) > 0)
        for X, x, Y, y, left in ((A, a, B, b, True), (A, a, A, a, True), (B, b, A, a, False)):
            if X.shape[0] * Y.shape[0] > 3_000_000:
                continue
            pr, pc = oc.mul_allpairs(X, x, Y, y, left)
            er, ec = oc.cleanup(pr, pc, 1e-15)
            for G in (1, 2, 3, 4, 8):
                shares, owned = [], 0
                for rank in range(G):
                    st = {}
                    shares.append(parallel.hash_partition_local(X, x, Y, y, rank, G, left, 1e-15, stats=st))
                    owned += st['pairs_owned']
                    assert st['keys_exchanged'] == 0
                assert owned == X.shape[0] * Y.shape[0], 'every pair has exactly one owner'
                gg = np.concatenate([s_[2] for s_ in shares])
                assert np.unique(gg).size == gg.size, 'a term came out of two ranks'
                order = np.argsort(gg, kind='stable')
                rows = np.concatenate([s_[0] for s_ in shares], axis=0)[order]; coeff = np.concatenate([s_[1] for s_ in shares])[order]
                assert np.array_equal(rows, er) and np.array_equal(coeff, ec), (n, G, left)


def test_hash_partitioned_shares_device_resident():
    """The device-resident share computation (csrc/partition.hip: gather of sub-operands, indexed fused product + cleanup, pair indices, merge by
    index) for G = 1, 2, 3, 8: the shares merged by pair index on the device (symgpu_merge_indexed_dev without cleanup) are the oracle's result,
    also above the 2^22-key gate of the lazy cleanup flow (squared 3,000-term operator) and with Gaussian coefficients within 1e-12."""
    from symmer_amd import parallel
    from symmer_amd.kernels import DeviceOp
    rng = np.random.default_rng(707)
    for n, Ni, No, gauss in ((100, 700, 400, False), (1000, 900, 900, False), (100, 3000, 3000, False), (70, 500, 300, True)):
        A = onp.pack_rows(rng.random((Ni, 2 * n)) < 0.3); B = onp.pack_rows(rng.random((No, 2 * n)) < 0.3)
        if Ni < 2000:
            A[Ni // 2: Ni // 2 + 20] = A[:20]; B[No // 2:] = B[: No - No // 2]
        a = (rng.standard_normal(Ni) + 1j * rng.standard_normal(Ni)) if gauss else dyadic(rng, Ni)
        b = (rng.standard_normal(No) + 1j * rng.standard_normal(No)) if gauss else dyadic(rng, No)
        dA, dB = DeviceOp.upload(A, a), DeviceOp.upload(B, b)
        for X, Y, ex, left in ((dA, dB, (A, a, B, b), True), (dA, dA, (A, a, A, a), True), (dB, dA, (B, b, A, a), False)):
            if Ni >= 3000 and X is not Y:
                continue
            pr, pc = oc.mul_allpairs(*ex, left)
            er, ec = oc.cleanup(pr, pc, 1e-15)
            for G in ((1, 2, 3, 8) if Ni < 3000 else (2,)):
                # G = 1 and 2 of the first case also with the sub-products CUT along their outer index (the limit of one indexed product call is
                # 2^32 pairs — at the north star's 1e5 x 1e5 terms and two classes a sub-product has 2.5e9 .. 5e9): forced here with a tiny limit
                cut = 9000 if (G <= 2 and Ni == 700) else None
                shares = [parallel.hash_partition_local_dev(X, Y, rank, G, left, 1e-15, max_pairs=cut) for rank in range(G)]
                key_bits = int(ex[0].shape[0] * ex[2].shape[0] - 1).bit_length()
                merged = kernels.merge_indexed_dev(shares, key_bits, False)
                rows, coeff = merged.download()
                g = kernels.op_first_index(merged)
                assert np.all(np.diff(g.astype(np.int64)) > 0)
                if gauss:
                    assert_op_equal(onp.unpack_rows(rows, n), coeff, onp.unpack_rows(er, n), ec, exact=False, tol=TOL)
                else:
                    assert np.array_equal(rows, er) and np.array_equal(coeff, ec), (n, G, left)
                for h in shares + [merged]:
                    h.free()
        dA.free(); dB.free()


def test_mul_cleanup_tiled_over_outer_operand():
    """Products beyond the 32-bit pair-index limit are tiled over the outer operand; forced here with a tiny tile."""
    rng = np.random.default_rng(8)
    n, N, M = 100, 300, 290
    A = PauliwordOp(rng.random((N, 2 * n)) < 0.3, dyadic(rng, N))
    rows, coeff = kernels.mul_cleanup(A.packed, A.coeff_vec, A.packed[:M], A.coeff_vec[:M], True, 1e-15, max_pairs=7000)
    erows, ecoeff = oc.mul(A.packed, A.coeff_vec, A.packed[:M], A.coeff_vec[:M])
    assert np.array_equal(rows, erows) and np.array_equal(coeff, ecoeff)


@pytest.mark.parametrize('T,n', [(3_000_000, 2), (2_500_001, 1), (1_000_000, 40)])
def test_cleanup_long_segments_sum_in_input_order(T, n):
    """Segments of 1e5+ equal rows span many 64-position chunks and the chunk ranges of several wavefronts; with Gaussian
    coefficients the result only matches bit for bit if every segment is summed sequentially in ascending input order."""
    rng = np.random.default_rng(77 + n)
    base = packing.pack_rows(rng.random((16, 2 * n)) < 0.5)
    rows = base[rng.integers(0, 16, T)]
    coeff = rng.standard_normal(T) + 1j * rng.standard_normal(T)
    r, c = kernels.cleanup(rows, coeff, 1e-15)
    er, ec = oc.cleanup(rows, coeff, 1e-15)
    assert np.array_equal(r, er) and np.array_equal(c, ec)


@pytest.mark.parametrize('n,N,M', [(3, 300, 200), (100, 400, 300), (1, 60, 50), (1000, 150, 90), (12, 2500, 40)])
def test_cleanup_lazy_and_fused_switches(n, N, M, monkeypatch):
    """Round 3: terms that merge with nothing are decided in index order before the sort (k_mark_singles) and the output stage is one
    fused launch (k_emit_fused).  SYMGPU_CLEANUP_LAZY=0 files every term from the sorted order, SYMGPU_EMIT_FUSED=0 takes the
    batched list + stream stage: all four combinations give the oracle's rows, order and sums — on a general product, a squared
    operator and a plain cleanup, duplicate-heavy (n = 1, 3: a handful of distinct rows) and duplicate-free alike."""
    rng = np.random.default_rng(4100 + n + N)
    A = PauliwordOp(rng.random((N, 2 * n)) < 0.3, dyadic(rng, N))
    B = PauliwordOp(rng.random((M, 2 * n)) < 0.3, dyadic(rng, M))
    stacked_rows = np.concatenate([A.packed, B.packed, A.packed[: N // 2]])
    stacked_coeff = np.concatenate([A.coeff_vec, B.coeff_vec, -A.coeff_vec[: N // 2]])     # exact cancellations among them
    want_ab = oc.mul(A.packed, A.coeff_vec, B.packed, B.coeff_vec)
    want_aa = oc.mul(A.packed, A.coeff_vec, A.packed, A.coeff_vec)
    want_c = oc.cleanup(stacked_rows, stacked_coeff, 1e-15)
    for lazy in ('1', '0'):
        for fused in ('1', '0'):
            monkeypatch.setenv('SYMGPU_CLEANUP_LAZY', lazy)
            monkeypatch.setenv('SYMGPU_EMIT_FUSED', fused)
            for thr in (1e-15, None):
                got = kernels.mul_cleanup(A.packed, A.coeff_vec, B.packed, B.coeff_vec, True, thr)
                ref = want_ab if thr is not None else oc.mul(A.packed, A.coeff_vec, B.packed, B.coeff_vec, None)
                assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]), (lazy, fused, thr)
            got = kernels.mul_cleanup(A.packed, A.coeff_vec, A.packed, A.coeff_vec, True, 1e-15)
            assert np.array_equal(got[0], want_aa[0]) and np.array_equal(got[1], want_aa[1]), (lazy, fused, 'squared')
            got = kernels.cleanup(stacked_rows, stacked_coeff, 1e-15)
            assert np.array_equal(got[0], want_c[0]) and np.array_equal(got[1], want_c[1]), (lazy, fused, 'cleanup')
    monkeypatch.delenv('SYMGPU_CLEANUP_LAZY')
    monkeypatch.delenv('SYMGPU_EMIT_FUSED')
    # the prefix sums behind the output positions: two launches up to 2^23 elements, the recursive form above it (forced here)
    monkeypatch.setenv('SYMGPU_SCAN_RECURSIVE', '1')
    got = kernels.mul_cleanup(A.packed, A.coeff_vec, B.packed, B.coeff_vec, True, 1e-15)
    assert np.array_equal(got[0], want_ab[0]) and np.array_equal(got[1], want_ab[1]), 'recursive scan'
    got = kernels.cleanup(stacked_rows, stacked_coeff, 1e-15)
    assert np.array_equal(got[0], want_c[0]) and np.array_equal(got[1], want_c[1]), 'recursive scan, cleanup'
    monkeypatch.delenv('SYMGPU_SCAN_RECURSIVE')


def test_mul_cleanup_unpacked_fallback_path(monkeypatch):
    """The fused product + cleanup normally sorts packed (hash | e | o | i) keys; operands whose index fields need more than
    32 bits, or a long mixed prefix run, use separate 64-bit keys + index values with materialised pair coefficients."""
    import os
    rng = np.random.default_rng(9)
    n, N, M = 70, 400, 150
    A = PauliwordOp(rng.random((N, 2 * n)) < 0.15, dyadic(rng, N))
    B = PauliwordOp(rng.random((M, 2 * n)) < 0.15, dyadic(rng, M))
    erows, ecoeff = oc.mul(A.packed, A.coeff_vec, B.packed, B.coeff_vec)
    monkeypatch.setenv('SYMGPU_CLEANUP_UNPACKED', '1')
    rows, coeff = kernels.mul_cleanup(A.packed, A.coeff_vec, B.packed, B.coeff_vec, True, 1e-15)
    assert np.array_equal(rows, erows) and np.array_equal(coeff, ecoeff)
    rows2, coeff2 = kernels.mul_cleanup(A.packed, A.coeff_vec, A.packed, A.coeff_vec, True, 1e-15)
    monkeypatch.delenv('SYMGPU_CLEANUP_UNPACKED')
    rows3, coeff3 = kernels.mul_cleanup(A.packed, A.coeff_vec, A.packed, A.coeff_vec, True, 1e-15)
    assert np.array_equal(rows2, rows3) and np.array_equal(coeff2, coeff3)


@pytest.mark.parametrize('n,N,M', [(100, 37, 1000), (1000, 129, 517), (64, 5, 64)])
def test_commutes_bit_packed_output(n, N, M):
    """symgpu_commutes_bits_dev: bit j of row i (little-endian u64 words, zero padding up to the word end) == C[i, j]."""
    import ctypes
    from symmer_amd import _lib
    from symmer_amd.kernels import DeviceOp
    rng = np.random.default_rng(200 + n)
    a = packing.pack_rows(rng.random((N, 2 * n)) < 0.3); b = packing.pack_rows(rng.random((M, 2 * n)) < 0.3)
    A, B = DeviceOp.upload(a), DeviceOp.upload(b)
    words = (M + 63) // 64
    lib = _lib.lib()
    bits = ctypes.c_void_p()
    _lib.check(lib.symgpu_dev_alloc(N * words * 8, ctypes.byref(bits)))
    junk = np.full(N * words, 0xFFFFFFFFFFFFFFFF, dtype='<u8')              # the kernel must overwrite every word completely
    _lib.check(lib.symgpu_dev_upload(bits, junk.ctypes.data, junk.nbytes))
    _lib.check(lib.symgpu_commutes_bits_dev(A.handle, 0, N, B.handle, bits))
    out = np.empty((N, words), dtype='<u8')
    _lib.check(lib.symgpu_dev_download(bits, out.ctypes.data, out.nbytes))
    expect = packing.pack_bits(oc.commutes(a, b), words)
    assert np.array_equal(out, expect)
    _lib.check(lib.symgpu_dev_free(bits)); A.free(); B.free()


@pytest.mark.parametrize('force', ['1', '0'])
@pytest.mark.parametrize('n,N,M,dens', [(1000, 3000, 2500, 0.3), (2000, 700, 4100, 0.3), (100, 513, 257, 0.3), (1, 70, 300, 0.5),
                                        (64, 1, 1000, 0.3), (4097, 40, 33, 0.3), (130, 1300, 2049, 0.01), (65, 2600, 64, 0.3),
                                        (300, 1, 1, 0.3), (257, 1281, 2048, 0.3)])
def test_commutes_both_kernels(n, N, M, dens, force, monkeypatch):
    """The Four-Russians kernel (commute_m4r.hip: LDS tables, SYMGPU_COMMUTE_M4R=1) and the register-tile kernel (=0) on
    the same ragged shapes — row counts around the 512/768/1280-row workgroup tiles, column counts around the 2048-column
    tile and the 64-bit word, a sparse operand (most k-blocks skipped), N = 1 and M = 1 — bytes and bit-packed output."""
    import ctypes
    from symmer_amd import _lib
    from symmer_amd.kernels import DeviceOp
    monkeypatch.setenv('SYMGPU_COMMUTE_M4R', force)
    rng = np.random.default_rng(300 + n + N)
    a = packing.pack_rows(rng.random((N, 2 * n)) < dens); b = packing.pack_rows(rng.random((M, 2 * n)) < dens)
    expect = oc.commutes(a, b)
    assert np.array_equal(kernels.commutes(a, b), expect)
    assert np.array_equal(kernels.commutes(a, a), oc.commutes(a, a))
    # row range of a device operand + bit-packed output
    A, B = DeviceOp.upload(a), DeviceOp.upload(b)
    lo, hi = N // 3, N
    words = (M + 63) // 64
    lib = _lib.lib()
    bits = ctypes.c_void_p()
    _lib.check(lib.symgpu_dev_alloc(max(1, (hi - lo) * words * 8), ctypes.byref(bits)))
    junk = np.full((hi - lo) * words, 0xFFFFFFFFFFFFFFFF, dtype='<u8')
    _lib.check(lib.symgpu_dev_upload(bits, junk.ctypes.data, junk.nbytes))
    _lib.check(lib.symgpu_commutes_bits_dev(A.handle, lo, hi, B.handle, bits))
    out = np.empty((hi - lo, words), dtype='<u8')
    _lib.check(lib.symgpu_dev_download(bits, out.ctypes.data, out.nbytes))
    assert np.array_equal(out, packing.pack_bits(expect[lo:hi], words))
    _lib.check(lib.symgpu_dev_free(bits)); A.free(); B.free()


@pytest.mark.parametrize('M', [4208, 4200])
@pytest.mark.parametrize('r', ['16', '24', '48'])
def test_commutes_m4r_tile_heights(r, M, monkeypatch):
    """Every instantiation of the Four-Russians kernel (rows per 16-lane slot; two 7-bit tables per step, csrc/commute_m4r7.hip) on a shape that
    leaves partial row and column tiles (fewer tiles than compute units: one tile per workgroup); M = 4208 takes the fused byte-expanding
    epilogue (16-byte stores), M = 4200 the bit-packed rows + separate expansion."""
    monkeypatch.setenv('SYMGPU_COMMUTE_M4R', '1'); monkeypatch.setenv('SYMGPU_M4R_R', r)
    rng = np.random.default_rng(77)
    n, N = 200, 1500
    a = packing.pack_rows(rng.random((N, 2 * n)) < 0.3); b = packing.pack_rows(rng.random((M, 2 * n)) < 0.3)
    expect = oc.commutes(a, b)
    assert np.array_equal(kernels.commutes(a, b), expect)


@pytest.mark.parametrize('force', ['1', '0'])
@pytest.mark.parametrize('N,M,off', [(700, 2500, 3), (513, 1001, 8), (300, 15, 13), (64, 7, 1), (1100, 4099, 0), (5, 3, 5)])
def test_commutes_any_row_length_any_alignment(N, M, off, force, monkeypatch):
    """np.bool_ tables whose rows are not multiples of 16 (8) bytes, or whose base is not aligned, are computed as bit-packed rows and
    written by the flat 16-byte expansion (commute_m4r.hip k_bits_to_bytes_flat: chunks that run over a row end, rows shorter than a
    chunk, unaligned head and tail) — both commutation kernels, output at `off` bytes into a device buffer whose other bytes must
    stay untouched."""
    import ctypes
    from symmer_amd import _lib
    from symmer_amd.kernels import DeviceOp
    monkeypatch.setenv('SYMGPU_COMMUTE_M4R', force)
    rng = np.random.default_rng(400 + N + M)
    n = 130
    a = packing.pack_rows(rng.random((N, 2 * n)) < 0.3); b = packing.pack_rows(rng.random((M, 2 * n)) < 0.3)
    A, B = DeviceOp.upload(a), DeviceOp.upload(b)
    lib = _lib.lib()
    total = N * M + 64
    buf = ctypes.c_void_p()
    _lib.check(lib.symgpu_dev_alloc(total, ctypes.byref(buf)))
    junk = np.full(total, 0xEE, dtype=np.uint8)
    _lib.check(lib.symgpu_dev_upload(buf, junk.ctypes.data, total))
    _lib.check(lib.symgpu_commutes_dev(A.handle, 0, N, B.handle, ctypes.c_void_p(buf.value + off)))
    got = np.empty(total, dtype=np.uint8)
    _lib.check(lib.symgpu_dev_download(buf, got.ctypes.data, total))
    assert np.array_equal(got[off:off + N * M].reshape(N, M).astype(bool), oc.commutes(a, b))
    assert np.all(got[:off] == 0xEE) and np.all(got[off + N * M:] == 0xEE), 'bytes outside the table were written'
    _lib.check(lib.symgpu_dev_free(buf)); A.free(); B.free()


_STREAMK_REF = {}


@pytest.mark.parametrize('mode', ['stream', 'fixup'])
@pytest.mark.parametrize('M', [32752, 32750])
@pytest.mark.parametrize('r', ['16', '24', '48'])
def test_commutes_m4r_stream_k(r, M, mode, monkeypatch):
    """The stream-K launch of the Four-Russians kernel (csrc/commute_m4r7.hip): 17 row tiles x 16 column tiles = 272 tiles for the 256
    persistent workgroups, so every workgroup's range starts and ends inside a tile; ragged last row and column tiles; M = 32752 rows of
    whole 16-byte chunks, M = 32750 a byte-wise row end and unaligned row starts.  `stream`: a split tile is finished by the owner of its first steps from
    the neighbour's published part; `fixup` (SYMGPU_M4R_FIXUP=1): both parts go to scratch and k_m7_fixup writes the tile — the path
    of a neighbour that has not run yet.  Both must equal, byte for byte, the table of the one-tile-per-workgroup launch
    (SYMGPU_M4R_STREAM=0: what fewer tiles than CUs take), which is checked against the C oracle on 48 random 256 x 256 blocks and on
    the ragged last rows and columns (the whole 0.85 GB table through the oracle takes a minute per shape)."""
    monkeypatch.setenv('SYMGPU_COMMUTE_M4R', '1'); monkeypatch.setenv('SYMGPU_M4R_R', r)
    n = 100
    N = 32 * int(r) * 17 - 37
    key = (N, M)
    if key not in _STREAMK_REF:
        _STREAMK_REF.clear()                                          # one table at a time: up to 0.85 GB each
        rng = np.random.default_rng(78)
        a = packing.pack_rows(rng.random((N, 2 * n)) < 0.3); b = packing.pack_rows(rng.random((M, 2 * n)) < 0.3)
        a[5] = 0                                                      # an identity row
        monkeypatch.setenv('SYMGPU_M4R_STREAM', '0')
        ref = kernels.commutes(a, b)
        for _ in range(48):
            r0, c0 = int(rng.integers(0, N - 256)), int(rng.integers(0, M - 256))
            assert np.array_equal(ref[r0:r0 + 256, c0:c0 + 256], oc.commutes(a[r0:r0 + 256], b[c0:c0 + 256]))
        assert np.array_equal(ref[N - 300:], oc.commutes(a[N - 300:], b)) and np.array_equal(ref[:, M - 300:], oc.commutes(a, b[M - 300:]))
        assert ref[5].all()
        _STREAMK_REF[key] = (a, b, ref)
    a, b, ref = _STREAMK_REF[key]
    monkeypatch.setenv('SYMGPU_M4R_STREAM', '1')                      # (operators this short take one tile per workgroup by themselves)
    if mode == 'fixup':
        monkeypatch.setenv('SYMGPU_M4R_FIXUP', '1')
    got = kernels.commutes(a, b)
    assert got.shape == ref.shape and np.array_equal(got, ref)


def test_commutes_m4r_all_identity_left_operand(monkeypatch):
    """An all-identity left operand has no non-zero 7-bit group: the kernel runs one step on the zero group and everything commutes."""
    monkeypatch.setenv('SYMGPU_COMMUTE_M4R', '1'); monkeypatch.setenv('SYMGPU_M4R_STREAM', '1')
    rng = np.random.default_rng(79)
    a = np.zeros((8704, 4), dtype='<u8'); b = packing.pack_rows(rng.random((32768, 200)) < 0.3)
    assert kernels.commutes(a, b).all()
    monkeypatch.setenv('SYMGPU_M4R_R', '16')
    assert kernels.commutes(a[:600], b[:4096]).all()


@pytest.mark.parametrize('case', family('jordan'))
def test_jordan_and_reindex_golden(case):
    """check_jordan_independent / reindex / jordan_generator_reconstruction against outputs of the reference."""
    from symmer_amd.operators import check_jordan_independent
    kind = int(case['kind'])
    if kind == 0:
        P = PauliwordOp(case['symp'].astype(bool), np.ones(case['symp'].shape[0]))
        assert bool(check_jordan_independent(P)) == bool(case['expect'])
    elif kind == 1:
        P = PauliwordOp(case['symp'].astype(bool), case['coeff'])
        vals = [int(v) for v in case['vals']]
        Q = P.reindex(vals) if int(case['as_list']) else P.reindex({int(a): int(b) for a, b in zip(case['keys'], vals)})
        assert np.array_equal(Q.symp_matrix, case['out'].astype(bool)) and np.array_equal(Q.coeff_vec, P.coeff_vec)
    else:
        G = PauliwordOp(case['g_symp'].astype(bool), np.ones(case['g_symp'].shape[0]))
        H = PauliwordOp(case['h_symp'].astype(bool), np.ones(case['h_symp'].shape[0]))
        R, mask = H.jordan_generator_reconstruction(G)
        assert np.array_equal(mask, case['mask'].astype(bool))
        assert np.array_equal(np.asarray(R)[mask], case['R'][mask])


@pytest.mark.parametrize('case', family('rotate')[::3])
def test_rotate_golden_rows_left_in_memory(case, monkeypatch):
    """SYMGPU_ROT_HBM=2 makes the one-launch rotation kernel leave the rows in memory (the form operators beyond the chip's LDS take,
    csrc/rotate_resident.hip) whatever the operator's size: the reference-generated cases, every row length among them."""
    monkeypatch.setenv('SYMGPU_ROT_HBM', '2')
    test_rotate_golden(case)


@pytest.mark.parametrize('case', family('rotate')[::3])
def test_rotate_golden_general_path(case, monkeypatch):
    """SYMGPU_ROTATE_GENERAL=1 disables the hash-join (non-Clifford) and fused (Clifford) fast paths: the stacked-operator +
    cleanup path that serves inputs with duplicate rows and very large operators must give the same answers."""
    monkeypatch.setenv('SYMGPU_ROTATE_GENERAL', '1')
    test_rotate_golden(case)


@pytest.mark.parametrize('env', [{'SYMGPU_GF2_M4R': '0'}, {'SYMGPU_GF2_FUSED_SELECT': '0'},
                                 {'SYMGPU_GF2_FUSED_SELECT': '0', 'SYMGPU_GF2_SMALL': '0'}, {'SYMGPU_GF2_SMALL': '0'}])
@pytest.mark.parametrize('case', family('gf2')[::4])
def test_gf2_golden_other_sweep_paths(case, env, monkeypatch):
    """The GF(2) elimination has these schedules: lookahead + Four-Russians sweep with the selector launch fused into phase 0
    (default: two launches per block), the same with the selector launch on its own (three: what a time-out falls back to), and the
    flag-per-block-row sweep (LDS attribute refused); all must reproduce the reference's matrices (small matrices too: SYMGPU_GF2_SMALL=0)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    test_gf2_golden(case)


@pytest.mark.parametrize('seed', range(6))
def test_rotation_chain_with_duplicates_and_tiny_terms(seed):
    """perform_rotations runs the per-rotation cleanup only until the operator is known to be duplicate-free and above the
    threshold; the result must equal the reference's cleanup-after-every-rotation on inputs that need the first one."""
    rng = np.random.default_rng(600 + seed)
    n, T = int(rng.choice([3, 20, 70])), 60
    base = rng.random((12, 2 * n)) < 0.4
    symp = base[rng.integers(0, 12, T)]                                  # heavy duplication
    coeff = dyadic(rng, T)
    coeff[::7] = 1e-17                                                   # below the cleanup threshold
    rots = [(rng.random(2 * n) < 0.5, float(a)) for a in rng.choice([np.pi / 2, np.pi, -np.pi / 2, 3 * np.pi / 2], 9)]
    P = PauliwordOp(symp, coeff)
    R = P.perform_rotations([(PauliwordOp(q.reshape(1, -1), [1]), a) for q, a in rots])
    er, ec = onp.perform_rotations(symp, coeff, rots)
    assert_op_equal(R.symp_matrix, R.coeff_vec, er, ec, exact=False, tol=TOL)
    mixed = rots[:3] + [(rots[3][0], 0.37), (rots[4][0], -1.1)] + rots[5:]
    R2 = P.perform_rotations([(PauliwordOp(q.reshape(1, -1), [1]), a) for q, a in mixed])
    er2, ec2 = onp.perform_rotations(symp, coeff, mixed)
    assert_op_equal(R2.symp_matrix, R2.coeff_vec, er2, ec2, exact=False, tol=TOL)


@pytest.mark.parametrize('R,C,dens', [(700, 700, 0.5), (700, 700, 0.003), (1500, 5000, 0.3), (300, 9000, 0.05), (129, 64, 0.5), (1500, 3000, 0.003),
                                      (900, 16384, 0.001), (257, 16400, 0.002), (640, 2000, 0.01), (70, 130, 0.02)])
def test_rref_fused_selector_launch_equals_separate(R, C, dens, monkeypatch):
    """Round 3: the selectors of a block are computed inside phase 0's launch (the tile workgroups wait for the 64 selectors of their
    rows, published by the first four workgroups of the grid).  Same reduced matrix, pivots and reference row-XOR count as with the
    selector launch on its own, and as the oracle — dense, sparse (blocks of a few rows), wide and tall."""
    rng = np.random.default_rng(R * 31 + C)
    m = rng.random((R, C)) < dens
    packed = packing.pack_bits(m)
    red1, cnt1, piv1 = kernels.rref(packed, want_pivots=True)
    monkeypatch.setenv('SYMGPU_GF2_FUSED_SELECT', '0')
    red2, cnt2, piv2 = kernels.rref(packed, want_pivots=True)
    monkeypatch.delenv('SYMGPU_GF2_FUSED_SELECT')
    ered, ecnt = onp.rref_noswap(m, count_xors=True)
    assert np.array_equal(red1, red2) and cnt1 == cnt2 == ecnt and np.array_equal(piv1, piv2)
    assert np.array_equal(packing.unpack_bits(red1, C), ered)


@pytest.mark.parametrize('R,C,dens', [(64, 4096, 0.4), (33, 64, 0.5), (64, 3000, 0.01), (10, 40, 0.2)])
def test_rref_small_and_blocked_paths_agree(R, C, dens, monkeypatch):
    """Matrices with <= 64 rows and <= 64 words take the one-workgroup path; SYMGPU_GF2_SMALL=0 sends them through the blocked
    path: same reduced matrix, pivots and reference row-XOR count, both equal to the oracle."""
    rng = np.random.default_rng(R * 1000 + C)
    m = rng.random((R, C)) < dens
    packed = packing.pack_bits(m)
    red1, cnt1, piv1 = kernels.rref(packed, want_pivots=True)
    monkeypatch.setenv('SYMGPU_GF2_SMALL', '0')
    red2, cnt2, piv2 = kernels.rref(packed, want_pivots=True)
    ered, ecnt = onp.rref_noswap(m, count_xors=True)
    assert np.array_equal(red1, red2) and cnt1 == cnt2 == ecnt and np.array_equal(piv1, piv2)
    assert np.array_equal(packing.unpack_bits(red1, C), ered)


# ---------------------------------------------------------------- few pairs of very long rows (wide.hip) ---
@pytest.mark.parametrize('force', ['1', '0', None])
@pytest.mark.parametrize('n,N,M', [(70000, 1, 1), (20000, 3, 5), (16321, 7, 2), (200003, 2, 2), (130, 9, 11), (1000, 40, 33)])
def test_wide_rows_word_parallel_path(n, N, M, force, monkeypatch):
    """The word-parallel kernels for few pairs of very long rows (SYMGPU_WIDE=1 forces them at any width, =0 forbids them, unset:
    rows of >= 256 words per block and <= 65536 pairs) against the C oracle: commutation bytes and bits, all-pairs product
    (rows + coefficients, both orientations), fused product + cleanup (keys path; squared operator and general pair)."""
    if force is None:
        monkeypatch.delenv('SYMGPU_WIDE', raising=False)
    else:
        monkeypatch.setenv('SYMGPU_WIDE', force)
    rng = np.random.default_rng(900 + n + N)
    A = rng.random((N, 2 * n)) < 0.3; B = rng.random((M, 2 * n)) < 0.3
    if N > 1:
        A[1] = A[0]                                                    # duplicate rows: the cleanup has something to merge
    a, b = packing.pack_rows(A), packing.pack_rows(B)
    assert np.array_equal(kernels.commutes(a, b), oc.commutes(a, b))
    assert np.array_equal(PauliwordOp(A, np.ones(N)).commutes_termwise(PauliwordOp(B, np.ones(M))), oc.commutes(a, b).astype(bool))
    import ctypes
    from symmer_amd import _lib
    da, db = kernels.DeviceOp.upload(a), kernels.DeviceOp.upload(b)
    words = (M + 63) // 64
    bits = ctypes.c_void_p()
    _lib.check(_lib.lib().symgpu_dev_alloc(N * words * 8, ctypes.byref(bits)))
    _lib.check(_lib.lib().symgpu_commutes_bits_dev(da.handle, 0, N, db.handle, bits))
    got = np.empty((N, words), dtype='<u8')
    _lib.check(_lib.lib().symgpu_dev_download(bits, got.ctypes.data, got.nbytes))
    _lib.check(_lib.lib().symgpu_dev_free(bits))
    assert np.array_equal(np.unpackbits(got.view(np.uint8), axis=1, bitorder='little')[:, :M], oc.commutes(a, b))
    da.free(); db.free()
    ca, cb = dyadic(rng, N), dyadic(rng, M)
    for left in (True, False):
        rows, coeff = kernels.mul_allpairs(a, ca, b, cb, left)
        erows, ecoeff = oc.mul_allpairs(a, ca, b, cb, left)
        assert np.array_equal(rows, erows) and np.array_equal(coeff, ecoeff)
    PA, PB = PauliwordOp(A, ca), PauliwordOp(B, cb)
    for X, Y in ((PA, PB), (PB, PA), (PA, PA)):
        R = X * Y
        es, ec = onp.mul(X.symp_matrix, X.coeff_vec, Y.symp_matrix, Y.coeff_vec)
        assert np.array_equal(R.symp_matrix, es) and np.array_equal(R.coeff_vec, ec)


@pytest.mark.parametrize('n,T', [(524288 + 77, 6), (600000, 3)])
def test_very_long_rows_segment_parallel_hash(n, T):
    """Rows of >= 8192 words take the segment-parallel row hash (k_hash_rows_long); it must be the SAME function as the Horner
    scheme of k_hash_rows, the host's host_row_hash (a rotation's Q row) and rotate.hip's in-kernel hash: cleanup merges duplicates,
    and a rotation merges P with (P Q) Q — rows hashed by different implementations — exactly like the oracle."""
    rng = np.random.default_rng(1300 + T)
    S = rng.random((T, 2 * n)) < 0.3
    c = dyadic(rng, T)
    q = rng.random(2 * n) < 0.3
    symp = np.vstack([S, S[: T // 2] ^ q, S[:2]])                     # P, P*Q partners, plain duplicates
    coeff = np.hstack([c, dyadic(rng, T // 2), dyadic(rng, 2)])
    P = PauliwordOp(symp, coeff)
    C = P.cleanup()
    es, ec = onp.cleanup_op(symp, coeff)
    assert np.array_equal(C.symp_matrix, es) and np.array_equal(C.coeff_vec, ec) and C.n_terms == T + T // 2
    Q = PauliwordOp(q.reshape(1, -1), [1])
    for ang in (0.3, np.pi / 2):
        R = C._rotate_by_single_Pword(Q, ang)
        er, ec2 = onp.rotate_by_single_pword(es, ec, q, ang)
        assert_op_equal(R.symp_matrix, R.coeff_vec, er, ec2, exact=ang != 0.3, tol=TOL)
    assert np.array_equal(P.Y_count, (P.X_block & P.Z_block).sum(axis=1))     # one block per row (k_ycount_long)
    R2 = P * P                                                          # fused product + cleanup on operand hashes from the long kernel
    es2, ec3 = onp.mul(symp, coeff, symp, coeff)
    assert np.array_equal(R2.symp_matrix, es2) and np.array_equal(R2.coeff_vec, ec3)


@pytest.mark.parametrize('n,T', [(1, 300), (63, 1000), (1000, 5000), (32768, 40), (40000, 7)])
def test_y_count_vs_numpy(n, T):
    """PauliwordOp.Y_count (base.py:604-615): per-row popcount of X & Z, thread per row and block per row (>= 512 words)."""
    rng = np.random.default_rng(1400 + n)
    P = PauliwordOp(rng.random((T, 2 * n)) < 0.4, np.ones(T))
    assert np.array_equal(P.Y_count, (P.X_block & P.Z_block).sum(axis=1))


# ---------------------------------------------------------------- round 5: reference-generated fixtures for the two soft spots of round 4 ----
@pytest.mark.parametrize('case', family('noncontextual'))
def test_noncontextual_golden(case):
    """PauliwordOp.is_noncontextual (device: csrc/project.hip) and check_adjmat_noncontextual on the device-computed adjacency matrix against the
    reference's answers (tests/golden/noncontextual.npz, oracle/tools/gen_golden_noncontextual.py; base.py:1074-1088, utils.py:567-589)."""
    from symmer_amd.operators.utils import check_adjmat_noncontextual
    symp = as_bool(case['symp'])
    P = PauliwordOp(symp, np.ones(symp.shape[0]))
    assert P.is_noncontextual == bool(case['is_noncontextual'])
    assert check_adjmat_noncontextual(P.adjacency_matrix) == bool(case['adjmat_noncontextual'])
    packed_only = PauliwordOp._from_packed(packing.pack_rows(symp), symp.shape[1] // 2, np.ones(symp.shape[0]))
    assert packed_only.is_noncontextual == bool(case['is_noncontextual']) and packed_only._symp is None


def _glue_cases(kind):
    return [c for c in family('api_glue') if int(c['kind']) == kind]


@pytest.mark.parametrize('resident', [False, True])
@pytest.mark.parametrize('case', _glue_cases(0))
def test_api_glue_sort_getitem_dagger_dictionary_golden(case, resident):
    """sort by every criterion in both orders, __getitem__ (int, negative, slices, lists, arrays, masks), dagger, multiply_by_constant and
    to_dictionary against outputs of the reference (tests/golden/api_glue.npz; base.py:455-492, 894-927, 1366-1376, 750-762, 1403-1416) — for an
    operator built from host arrays and for the SAME operator living on the device only (indexing and scaling then run there)."""
    symp, coeff = as_bool(case['in_symp']), case['in_coeff']
    T = symp.shape[0]

    def fresh():
        if not resident:
            return PauliwordOp(symp, coeff.copy())
        dev = kernels.DeviceOp.upload(packing.pack_rows(symp), coeff)
        return PauliwordOp._from_device(dev, symp.shape[1] // 2)
    for by in ('magnitude', 'lex', 'weight', 'support', 'Z', 'X', 'Y'):
        for key in ('decreasing', 'increasing'):
            got = fresh().sort(by=by, key=key)
            assert_op_equal(got.symp_matrix, got.coeff_vec, case[f'sort_{by}_{key}_symp'], case[f'sort_{by}_{key}_coeff'])
    picks = {'int0': 0, 'intlast': T - 1, 'neg1': -1, 'negT': -T, 'slice_all': slice(None), 'slice_mid': slice(1, T - 1), 'slice_step': slice(0, T, 2),
             'slice_open_end': slice(T // 2, None), 'slice_open_start': slice(None, T // 2), 'list': [int(v) for v in case['get_list_idx']],
             'array': case['get_array_idx'], 'mask': case['get_mask'].astype(bool)}
    for name, key in picks.items():
        got = fresh()[key]
        assert_op_equal(got.symp_matrix, got.coeff_vec, case[f'get_{name}_symp'], case[f'get_{name}_coeff'])
    P = fresh()
    for got, name in ((P.dagger, 'dagger'), (P.multiply_by_constant(0.5 - 0.25j), 'times_const'), (P * 3, 'times_real')):
        assert np.array_equal(got.symp_matrix, as_bool(case[f'{name}_symp']))
        assert np.allclose(got.coeff_vec, case[f'{name}_coeff'], rtol=0, atol=1e-13)
    d = fresh().to_dictionary
    assert list(d.keys()) == [str(s_) for s_ in case['dict_keys']]
    assert np.allclose(np.array(list(d.values())), case['dict_vals'], rtol=0, atol=1e-12)


@pytest.mark.parametrize('case', _glue_cases(1))
def test_api_glue_tensor_pow_sub_golden(case):
    """tensor, __pow__ (0..3), __sub__ and __add__ against outputs of the reference (tests/golden/api_glue.npz; base.py:1188-1204, 875-892, 742-748)."""
    L = PauliwordOp(as_bool(case['left_symp']), case['left_coeff']); R = PauliwordOp(as_bool(case['right_symp']), case['right_coeff'])
    S = PauliwordOp(as_bool(case['other_symp']), case['other_coeff'])
    got = L.tensor(R)
    assert_op_equal(got.symp_matrix, got.coeff_vec, case['tensor_symp'], case['tensor_coeff'])
    for e in (0, 1, 2, 3):
        got = L ** e
        assert_op_equal(got.symp_matrix, got.coeff_vec, case[f'pow{e}_symp'], case[f'pow{e}_coeff'])
    for got, name in ((L - S, 'sub'), (L + S, 'add')):
        assert_op_equal(got.symp_matrix, got.coeff_vec, case[f'{name}_symp'], case[f'{name}_coeff'])


# ---------------------------------------------------------------- round 5: f2 on packed rows, device side (csrc/genrec.hip) ----
@pytest.mark.parametrize('n,T,g,dep', [(1, 3, 1, False), (3, 10, 3, False), (5, 40, 6, False), (20, 200, 15, False), (63, 100, 40, False), (64, 300, 64, False),
                                       (65, 70, 33, False), (100, 500, 90, False), (130, 64, 128, False), (40, 100, 12, True), (6, 30, 12, False)])
def test_generators_rank_and_reconstruction_vs_oracle(n, T, g, dep):
    """PauliwordOp.generators, check_independent and generator_reconstruction (base.py:1436-1456, utils.py:504-519, base.py:523-560) on packed
    rows on the device against the NumPy oracle: terms that are products of the generators mixed with terms outside their span, generator
    sets that are dependent (override_independence_check) and sets that fill all 2n dimensions; no bool matrix is built for the operands."""
    from symmer_amd.operators import check_independent
    rng = np.random.default_rng(n * 1000 + T)
    gens = rng.random((g, 2 * n)) < 0.4
    while onp.check_independent(gens) is False and not dep:
        gens = rng.random((g, 2 * n)) < 0.5
    if dep:
        gens[g // 2] = gens[0] ^ gens[1]                                       # dependent on purpose
    combos = rng.random((T, g)) < 0.3
    inside = (combos.astype(int) @ gens.astype(int)) % 2 == 1
    symp = np.where((np.arange(T) % 3 == 0)[:, None], rng.random((T, 2 * n)) < 0.4, inside)
    G = PauliwordOp._from_packed(packing.pack_rows(gens), n, np.ones(g))
    M = PauliwordOp._from_packed(packing.pack_rows(symp), n, np.ones(T))
    assert check_independent(G) == onp.check_independent(gens) == (not dep)
    R, mask = M.generator_reconstruction(G, override_independence_check=dep)
    eR, emask = onp.generator_reconstruction(symp, gens)
    assert R.dtype == eR.dtype and R.shape == eR.shape and np.array_equal(R, eR)
    assert mask.dtype == np.bool_ and np.array_equal(mask, emask)
    if not dep:
        assert np.array_equal((R[mask] @ gens.astype(int)) % 2 == 1, symp[mask])        # M = R B on the reconstructed terms
    assert G._symp is None and M._symp is None, 'an operand was expanded to one byte per bit'
    gen_op = M.generators
    egen = onp.generators(symp)
    assert np.array_equal(gen_op.symp_matrix, egen) and np.all(gen_op.coeff_vec == 1)
    assert M._symp is None


def test_generator_reconstruction_edge_shapes():
    """No generators / no terms / more generators than 2n keep the host glue of the reference's lines (device rref inside)."""
    rng = np.random.default_rng(5)
    n = 4
    M = PauliwordOp(rng.random((6, 2 * n)) < 0.5, np.ones(6))
    E = PauliwordOp(np.zeros((0, 2 * n), dtype=bool), [])
    R, mask = E.generator_reconstruction(M, override_independence_check=True)
    assert R.shape == (0, 6) and mask.shape == (0,)
    many = PauliwordOp(rng.random((11, 2 * n)) < 0.5, np.ones(11))
    R, mask = M.generator_reconstruction(many, override_independence_check=True)
    eR, emask = onp.generator_reconstruction(M.symp_matrix, many.symp_matrix)
    assert np.array_equal(R, eR) and np.array_equal(mask, emask)
