"""Generate ``tests/golden/api_glue.npz`` by RUNNING THE REFERENCE (through ``ref_shim``): the host glue around the hot path that users and
``__eq__`` rely on — ``sort`` by every criterion in both orders (base.py:455-492), ``__getitem__`` with int / negative int / slices /
index lists / boolean masks (:894-927), ``tensor`` (:1188-1204), ``dagger`` (:1366-1376), ``__pow__`` (:875-892), ``to_dictionary``
(:1403-1416), ``multiply_by_constant`` (:750-762), ``__sub__`` (:742-748).  Inputs and the reference's outputs; data only.
BUILD CONTAINER ONLY.  Run: python oracle/tools/gen_golden_api_glue.py
"""
import os, sys, warnings
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: F401
warnings.simplefilter('ignore')
import numpy as np
from symmer.operators import PauliwordOp

OUT = os.path.join(HERE, '..', '..', 'tests', 'golden')
rng = np.random.default_rng(90210)
cases, k = {}, 0
SORTS = ['magnitude', 'lex', 'weight', 'support', 'Z', 'X', 'Y']


def put(prefix, op):
    cases[f'{prefix}_symp'] = np.asarray(op.symp_matrix, dtype=bool).astype(np.uint8)
    cases[f'{prefix}_coeff'] = np.asarray(op.coeff_vec, dtype=complex)


def dyadic(t):
    return (rng.integers(-8, 9, t) + 1j * rng.integers(-8, 9, t)) / 16.0


def distinct_magnitudes(t):
    return (rng.permutation(t) + 1.0) * np.exp(1j * rng.random(t) * 6.28)


# ---- kind 0: sort + __getitem__ + dagger + multiply_by_constant + to_dictionary on one operator
for n, T, coeffs in ((1, 3, 'dy'), (3, 12, 'mag'), (5, 40, 'dy'), (20, 64, 'mag'), (64, 50, 'mag'), (65, 33, 'dy'), (100, 120, 'mag'), (130, 17, 'dy')):
    symp = rng.random((T, 2 * n)) < 0.35
    if T > 8:
        symp[T // 2] = symp[0]                                              # a duplicate row: to_dictionary merges it
    c = dyadic(T) if coeffs == 'dy' else distinct_magnitudes(T)
    P = PauliwordOp(symp, c)
    pre = f'{k:04d}/'
    cases[pre + 'kind'] = np.array(0)
    put(pre + 'in', P)
    for by in SORTS:
        for key in ('decreasing', 'increasing'):
            put(pre + f'sort_{by}_{key}', P.sort(by=by, key=key))
    picks = {'int0': 0, 'intlast': T - 1, 'neg1': -1, 'negT': -T, 'slice_all': slice(None), 'slice_mid': slice(1, T - 1), 'slice_step': slice(0, T, 2),
             'slice_open_end': slice(T // 2, None), 'slice_open_start': slice(None, T // 2), 'list': [T - 1, 0, 0], 'array': np.array([0, T // 2]),
             'mask': (np.arange(T) % 3 == 0)}
    for name, key in picks.items():
        put(pre + f'get_{name}', P[key])
    cases[pre + 'get_list_idx'] = np.array(picks['list'])
    cases[pre + 'get_array_idx'] = picks['array']
    cases[pre + 'get_mask'] = picks['mask']
    put(pre + 'dagger', P.dagger)
    put(pre + 'times_const', P.multiply_by_constant(0.5 - 0.25j))
    put(pre + 'times_real', P * 3)
    d = P.to_dictionary
    cases[pre + 'dict_keys'] = np.array(list(d.keys()))
    cases[pre + 'dict_vals'] = np.array(list(d.values()), dtype=complex)
    k += 1

# ---- kind 1: tensor, __pow__, __sub__ on pairs
for nl, Tl, nr, Tr in ((1, 2, 1, 3), (2, 5, 3, 4), (3, 7, 64, 6), (40, 10, 30, 12), (64, 8, 1, 2), (100, 9, 31, 5)):
    L = PauliwordOp(rng.random((Tl, 2 * nl)) < 0.4, dyadic(Tl))
    R = PauliwordOp(rng.random((Tr, 2 * nr)) < 0.4, dyadic(Tr))
    pre = f'{k:04d}/'
    cases[pre + 'kind'] = np.array(1)
    put(pre + 'left', L); put(pre + 'right', R)
    put(pre + 'tensor', L.tensor(R))
    for e in (0, 1, 2, 3):
        put(pre + f'pow{e}', L ** e)
    S = PauliwordOp(np.vstack([L.symp_matrix[: Tl // 2 + 1], rng.random((3, 2 * nl)) < 0.4]), dyadic(Tl // 2 + 4))
    put(pre + 'other', S)
    put(pre + 'sub', L - S)
    put(pre + 'add', L + S)
    k += 1
cases['n_cases'] = np.array(k)
np.savez_compressed(os.path.join(OUT, 'api_glue.npz'), **cases)
print('api_glue:', k, 'cases,', len(cases), 'arrays,', os.path.getsize(os.path.join(OUT, 'api_glue.npz')), 'bytes')
