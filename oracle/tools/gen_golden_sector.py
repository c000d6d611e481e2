"""Generate ``tests/golden/sector.npz`` by RUNNING THE REFERENCE (through ``ref_shim``): ``IndependentOp.update_sector`` with
``QuantumState`` reference states (independent_op.py:275-301, assign_value :364-383) — basis states, superpositions dominated
by one sector, and superpositions too balanced to fix a stabiliser (assignment 0 + warning).  Data only.
BUILD CONTAINER ONLY.  Run: python oracle/tools/gen_golden_sector.py
"""
import os, sys, warnings
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: F401
warnings.simplefilter('ignore')
import numpy as np
from symmer.operators import PauliwordOp, IndependentOp, QuantumState

OUT = os.path.join(HERE, '..', '..', 'tests', 'golden')
rng = np.random.default_rng(777)
cases, k = {}, 0


def add(**arrays):
    global k
    for key, val in arrays.items():
        a = np.asarray(val)
        cases[f'{k:04d}/{key}'] = a.astype(np.uint8) if a.dtype == bool else a
    k += 1


stab_sets = [['ZIZI', 'IZIZ', 'IIZZ'], ['ZZII', 'IZZI', 'XXXX'], ['ZIIIII', 'IZZIII', 'IIIZZI', 'IIIIIZ'], ['ZZ'], ['XX', 'ZZ'],
             ['ZIZIZ', 'IZIZI', 'YYIII']]
for labels in stab_sets:
    n = len(labels[0])
    for trial in range(5):
        G = IndependentOp.from_list(labels)
        m = int((1, 1, 2, 3, 6)[trial])
        basis = np.unique(rng.integers(0, 2, size=(m, n)), axis=0)
        amp = rng.standard_normal(basis.shape[0]) + 1j * rng.standard_normal(basis.shape[0])
        if trial == 2:
            amp = np.array([0.98, 0.2][:basis.shape[0]], dtype=complex)          # one dominant basis state
        if trial == 3:
            amp = np.ones(basis.shape[0], dtype=complex)                           # balanced: some stabilisers undecidable
        psi = QuantumState(basis, amp).normalize
        G.update_sector(psi)
        add(stab_symp=G.symp_matrix, state_matrix=psi.state_matrix, state_coeff=np.asarray(psi.state_op.coeff_vec, dtype=complex),
            sector=np.asarray(G.coeff_vec, dtype=np.int64))
cases['n_cases'] = np.array(k)
np.savez_compressed(os.path.join(OUT, 'sector.npz'), **cases)
print('sector:', k, 'cases,', os.path.getsize(os.path.join(OUT, 'sector.npz')), 'bytes')
