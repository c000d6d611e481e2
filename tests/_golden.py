"""Loader for the committed golden fixtures (tests/golden/*.npz; produced by running the reference,
see oracle/tools/gen_golden.py).  Data only."""
import os, json
import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def family(name):
    """Yield one dict per case of a fixture family."""
    z = np.load(os.path.join(GOLDEN, f'{name}.npz'))
    n = int(z['n_cases'])
    cases = [dict() for _ in range(n)]
    for key in z.files:
        if '/' not in key:
            continue
        idx, field = key.split('/', 1)
        cases[int(idx)][field] = z[key]
    return cases


def known():
    z = np.load(os.path.join(GOLDEN, 'known.npz'))
    return {k: z[k] for k in z.files}


def known_single_qubit():
    with open(os.path.join(GOLDEN, 'known_single_qubit.json')) as f:
        return json.load(f)


def rotate_empty_cases():
    """tests/golden/rotate_empty.json (oracle/tools/gen_golden_rotate_empty.py): rotations that lose every term / start empty."""
    with open(os.path.join(GOLDEN, 'rotate_empty.json')) as f:
        doc = json.load(f)
    n2 = 2 * int(doc['n_qubits'])
    mat = lambda rows: np.array([[ch == '1' for ch in r] for r in rows], dtype=bool).reshape(-1, n2)
    vec = lambda v: np.array([complex(re, im) for re, im in v], dtype=complex)
    return [dict(kind=c['kind'], in_symp=mat(c['in_symp']), in_coeff=vec(c['in_coeff']), q=mat(c['q']), angles=c['angles'],
                 out_symp=mat(c['out_symp']), out_coeff=vec(c['out_coeff']), same_object=c['same_object']) for c in doc['cases']]


def as_bool(a):
    return np.asarray(a).astype(bool)


def unpackbits_matrix(packed, shape):
    R, C = int(shape[0]), int(shape[1])
    return np.unpackbits(packed, axis=1)[:, :C].astype(bool).reshape(R, C)


def assert_op_equal(rows, coeff, exp_rows, exp_coeff, exact=True, tol=1e-12):
    """Bit-exact rows + row order; coefficients bit-exact (dyadic) or within tol, after dropping on both
    sides rows with |c| <= tol (the Gaussian 'ghost row' rule of SURVEY §7)."""
    rows = as_bool(rows); exp_rows = as_bool(exp_rows)
    coeff = np.asarray(coeff, dtype=complex); exp_coeff = np.asarray(exp_coeff, dtype=complex)
    if not exact:
        k1 = np.abs(coeff) > tol; k2 = np.abs(exp_coeff) > tol
        rows, coeff, exp_rows, exp_coeff = rows[k1], coeff[k1], exp_rows[k2], exp_coeff[k2]
    assert rows.shape == exp_rows.shape, (rows.shape, exp_rows.shape)
    assert np.array_equal(rows, exp_rows)
    if exact:
        assert np.array_equal(coeff, exp_coeff)
    else:
        assert np.allclose(coeff, exp_coeff, rtol=0, atol=tol)
