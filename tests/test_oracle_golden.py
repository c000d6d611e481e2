"""CPU: pins BOTH oracles (oracle/oracle_np.py, oracle/oracle_c.c) to the golden fixtures that were
produced by running the reference (oracle/tools/gen_golden.py)."""
import numpy as np
import pytest
from oracle import oracle_np as onp
from oracle import oracle_c as oc
from _golden import family, known, known_single_qubit, as_bool, unpackbits_matrix, assert_op_equal, rotate_empty_cases


def _n(symp):
    return symp.shape[1] // 2


def test_known_answers():
    k = known()
    assert np.array_equal(onp.y_count(as_bool(k['ycount_symp'])), [0, 0, 3, 0])            # test_base.py:510-515
    assert np.array_equal(oc.ycount(onp.pack_rows(as_bool(k['ycount_symp']))), [0, 0, 3, 0])
    r, c = onp.cleanup_op(as_bool(k['cleanup_in_symp']), k['cleanup_in_coeff'])            # :537-544
    assert_op_equal(r, c, k['cleanup_out_symp'], k['cleanup_out_coeff'])
    assert c.tolist() == [2]
    assert np.array_equal(onp.commutes_termwise(as_bool(k['pl1_symp']), as_bool(k['pl2_symp'])),
                          as_bool(k['pl1_commutes_pl2']))                                   # :554-566
    assert np.array_equal(onp.commutes_termwise(as_bool(k['pl2_symp']), as_bool(k['pl2_symp'])),
                          as_bool(k['pl2_adjacency']))                                      # :568-579
    assert np.array_equal(onp.commutes_termwise(as_bool(k['doc_a']), as_bool(k['doc_b'])),
                          np.array([[1, 1, 1], [1, 0, 1]], dtype=bool))                     # base.py:944-951
    for nm, (l, r_) in (('AB', ('A', 'B')), ('BA', ('B', 'A'))):
        got = onp.mul(as_bool(k[f'small_{l}_symp']), k[f'small_{l}_coeff'], as_bool(k[f'small_{r_}_symp']), k[f'small_{r_}_coeff'])
        assert_op_equal(*got, k[f'small_{nm}_symp'], k[f'small_{nm}_coeff'])
    for key in ('H2', 'override', 'nosym'):
        got = onp.symmetry_generators_symp(as_bool(k[f'{key}_symp']))
        assert np.array_equal(got, as_bool(k[f'{key}_symgen']))
        h = as_bool(k[f'{key}_symp'])
        gc, _ = oc.symmetry_generators(onp.pack_rows(h), _n(h))
        assert np.array_equal(onp.unpack_rows(gc, _n(h)), as_bool(k[f'{key}_symgen']))
    # H2 generators are ZIIZ, IZIZ, IIZZ in that order
    assert np.array_equal(as_bool(k['H2_symgen'])[:, 4:], np.array([[1, 0, 0, 1], [0, 1, 0, 1], [0, 0, 1, 1]], dtype=bool))


def test_single_qubit_table():
    enc = {'I': (0, 0), 'X': (1, 0), 'Z': (0, 1), 'Y': (1, 1)}
    for pair, (res, (re, im)) in known_single_qubit().items():                              # test_base.py:596-613
        a = np.array([enc[pair[0]]], dtype=bool); b = np.array([enc[pair[1]]], dtype=bool)
        rows, c = onp.mul(a, [1], b, [1])
        assert rows.tolist() == [list(map(bool, enc[res]))] and c[0] == complex(re, im)
        rows, c = oc.mul(onp.pack_rows(a), [1], onp.pack_rows(b), [1])
        assert np.array_equal(onp.unpack_rows(rows, 1), [list(map(bool, enc[res]))]) and c[0] == complex(re, im)


@pytest.mark.parametrize('case', family('mul'))
def test_mul(case):
    a, b = as_bool(case['a_symp']), as_bool(case['b_symp'])
    exact = bool(case['exact'])
    assert_op_equal(*onp.mul(a, case['a_coeff'], b, case['b_coeff']), case['out_symp'], case['out_coeff'], exact)
    n = _n(a)
    rows, c = oc.mul(onp.pack_rows(a), case['a_coeff'], onp.pack_rows(b), case['b_coeff'])
    assert_op_equal(onp.unpack_rows(rows, n), c, case['out_symp'], case['out_coeff'], exact)


@pytest.mark.parametrize('case', family('cleanup'))
def test_cleanup(case):
    s = as_bool(case['in_symp']); thr = float(case['thr'])
    if thr < 0:
        got = onp.symplectic_cleanup(s, case['in_coeff'], None)
    else:
        got = onp.cleanup_op(s, case['in_coeff'], thr)
    exact = np.all(np.asarray(case['in_coeff']) * 16 == np.round(np.asarray(case['in_coeff']) * 16))
    assert_op_equal(*got, case['out_symp'], case['out_coeff'], exact)
    if s.shape[0]:
        rows, c = oc.cleanup(onp.pack_rows(s), case['in_coeff'], None if thr < 0 else thr)
        assert_op_equal(onp.unpack_rows(rows, _n(s)), c, case['out_symp'], case['out_coeff'], exact)


@pytest.mark.parametrize('case', family('commute'))
def test_commute(case):
    a, b = as_bool(case['a_symp']), as_bool(case['b_symp'])
    assert np.array_equal(onp.commutes_termwise(a, b), as_bool(case['out']))
    assert np.array_equal(onp.commutes_termwise(a, a), as_bool(case['adj']))
    assert np.array_equal(oc.commutes(onp.pack_rows(a), onp.pack_rows(b)), as_bool(case['out']))


@pytest.mark.parametrize('case', family('rotate'))
def test_rotate(case):
    s = as_bool(case['in_symp'])
    if int(case['chain']):
        rots = [(as_bool(q), float(a)) for q, a in zip(case['q'], case['angle'])]
        got = onp.perform_rotations(s, case['in_coeff'], rots)
        assert_op_equal(*got, case['out_symp'], case['out_coeff'], exact=False)
    else:
        ang = float(case['angle'])
        got = onp.rotate_by_single_pword(s, case['in_coeff'], as_bool(case['q']), ang)
        clifford = abs(round(2 * ang / np.pi) - 2 * ang / np.pi) <= 1e-18
        assert_op_equal(*got, case['out_symp'], case['out_coeff'], exact=clifford)


@pytest.mark.parametrize('case', family('rotate_dup'))
def test_rotate_with_duplicate_rows_and_threshold(case):
    """Operators that contain duplicate rows (odd multiples of pi/2 merge and threshold the product rows only, base.py:1143) and
    caller-supplied Clifford thresholds (base.py:1146) — fixtures from oracle/tools/gen_golden_rotate_dup.py."""
    ang, thr = float(case['angle']), float(case['threshold'])
    got = onp.rotate_by_single_pword(as_bool(case['in_symp']), case['in_coeff'], as_bool(case['q']), ang, thr)
    clifford = abs(round(2 * ang / np.pi) - 2 * ang / np.pi) <= thr
    assert_op_equal(*got, case['out_symp'], case['out_coeff'], exact=clifford)


@pytest.mark.parametrize('case', family('gf2'))
def test_gf2(case):
    m = unpackbits_matrix(case['m'], case['shape'])
    R, C = m.shape
    assert np.array_equal(onp.rref_noswap(m), unpackbits_matrix(case['rref_noswap'], (R, C)))
    assert np.array_equal(onp.rref_ordered(m), unpackbits_matrix(case['rref'], (R, C)))
    assert np.array_equal(onp.cref_noswap(m), unpackbits_matrix(case['cref_noswap'], (R, C)))
    assert np.array_equal(onp.cref_ordered(m), unpackbits_matrix(case['cref'], (R, C)))
    # C oracle on 64-bit packed rows (little-endian bit order of the ABI)
    wc = (C + 63) // 64
    bits = np.zeros((R, wc * 64), dtype=np.uint8); bits[:, :C] = m
    packed = np.packbits(bits, axis=1, bitorder='little').view('<u8')
    red, n_xor = oc.rref(packed)
    got = np.unpackbits(red.view(np.uint8), axis=1, bitorder='little')[:, :C].astype(bool)
    assert np.array_equal(got, unpackbits_matrix(case['rref_noswap'], (R, C)))
    assert n_xor == onp.rref_noswap(m, count_xors=True)[1]


@pytest.mark.parametrize('case', family('symgen'))
def test_symgen(case):
    h = as_bool(case['h_symp'])
    assert np.array_equal(onp.symmetry_generators_symp(h), as_bool(case['symgen']))
    if int(case['planted']) >= 0:
        assert as_bool(case['symgen']).shape[0] >= int(case['planted'])
    gc, _ = oc.symmetry_generators(onp.pack_rows(h), _n(h))
    assert np.array_equal(onp.unpack_rows(gc, _n(h)), as_bool(case['symgen']))
    assert np.array_equal(onp.generators(h), as_bool(case['gens']))
    r, mask = onp.generator_reconstruction(h, as_bool(case['gens']))
    assert np.array_equal(r, case['recon']) and np.array_equal(mask, as_bool(case['recon_mask']))


@pytest.mark.parametrize('case', [c for c in family('jordan') if int(c['kind']) in (0, 1)])
def test_jordan_independence_and_reindex_golden(case):
    if int(case['kind']) == 0:
        assert onp.check_jordan_independent(case['symp'].astype(bool)) == bool(case['expect'])
    else:
        keys, vals = case['keys'].tolist(), case['vals'].tolist()
        old, new = (sorted(vals), vals) if int(case['as_list']) else (keys, vals)
        assert np.array_equal(onp.reindex(case['symp'].astype(bool), old, new), case['out'].astype(bool))


def test_rotations_that_lose_every_term():
    """The 0 * I / no-terms alternation of the reference's cleanup() under rotations (base.py:631-632, utils.py:275-278, :1159-1161)."""
    cases = rotate_empty_cases()
    assert len(cases) > 1000
    for c in cases:
        if c['kind'] == 'single':
            rows, coeff = onp.rotate_by_single_pword(c['in_symp'], c['in_coeff'], c['q'][0], c['angles'][0])
        else:
            rows, coeff = onp.perform_rotations(c['in_symp'], c['in_coeff'], list(zip(c['q'], c['angles'])))
        assert rows.shape == c['out_symp'].shape and np.array_equal(rows, c['out_symp']), c
        assert np.allclose(coeff, c['out_coeff'], rtol=0, atol=1e-15), c


def test_squared_builder_equals_oracle_mul():
    """tests/_expected.py (the expected P * P of the full-size cfg3 GPU test, assembled from the oracle's pair coefficients) is the
    oracle's own product + cleanup: rows, row order and coefficients, dyadic and Gaussian."""
    from _expected import squared_expected
    rng = np.random.default_rng(404)
    for n, N, gauss in ((1000, 900, False), (1000, 700, True), (130, 1200, False)):
        A = onp.pack_rows(rng.random((N, 2 * n)) < 0.3)
        c = (rng.standard_normal(N) + 1j * rng.standard_normal(N)) if gauss else (rng.integers(-8, 9, N) + 1j * rng.integers(-8, 9, N)) / 16.0
        er, ec = oc.mul(A, c, A, c)
        o, i, cc = squared_expected(A, c)
        assert np.array_equal(A[i] ^ A[o], er) and np.array_equal(cc, ec)


@pytest.mark.parametrize('case', family('noncontextual'))
def test_noncontextual_golden(case):
    """tests/golden/noncontextual.npz (oracle/tools/gen_golden_noncontextual.py): the reference's is_noncontextual (base.py:1074-1088)
    and check_adjmat_noncontextual (utils.py:567-589) on random, clique-structured, all-commuting, duplicate-row and tiny operators."""
    symp = as_bool(case['symp'])
    assert onp.is_noncontextual(symp) == bool(case['is_noncontextual'])
    assert onp.check_adjmat_noncontextual(onp.commutes_termwise(symp, symp)) == bool(case['adjmat_noncontextual'])


@pytest.mark.parametrize('case', [c for c in family('api_glue') if int(c['kind']) == 0])
def test_sort_orders_golden(case):
    """tests/golden/api_glue.npz: the oracle's restatement of sort (base.py:455-492) reproduces the reference's term orders."""
    symp, coeff = as_bool(case['in_symp']), case['in_coeff']
    for by in ('magnitude', 'lex', 'weight', 'support', 'Z', 'X', 'Y'):
        for key in ('decreasing', 'increasing'):
            order = onp.sort_order(symp, coeff, by, key)
            assert np.array_equal(symp[order], as_bool(case[f'sort_{by}_{key}_symp'])) and np.array_equal(coeff[order], case[f'sort_{by}_{key}_coeff']), (by, key)
