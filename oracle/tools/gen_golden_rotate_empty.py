"""Generate ``tests/golden/rotate_empty.json`` by RUNNING THE REFERENCE (imported through ``ref_shim``): rotations of operators
that lose every term on the way, start without terms, or start as 0 * I.

BUILD CONTAINER ONLY (needs /root/reference).  Data only: inputs and the reference's outputs.
Run:  python oracle/tools/gen_golden_rotate_empty.py

Why a family of its own: the reference's ``cleanup()`` turns an operator WITHOUT terms into 0 * I (base.py:631-632)
and 0 * I back into an operator without terms (utils.py:275-278), so under ``perform_rotations`` (a cleanup after every rotation,
base.py:1185) the two states alternate, and WHICH one a step ends in depends on how the terms were lost: in the rotation
(odd multiple of pi/2: ``anticom_self * Q`` merges to nothing, then cleanup() -> 0 * I), in ``commute_self + anticom_part`` of
the non-Clifford branch (both parts without rows -> 0 * I, then cleanup() -> no terms; cancelled commuting rows -> no terms, then
cleanup() -> 0 * I), or in the loop's own cleanup (even multiple: rows kept, cleanup() -> no terms).  ``single/*`` cases hold
one ``_rotate_by_single_Pword`` call, ``chain`` cases a ``perform_rotations`` call with K = 1 .. 4.  JSON (rows as '0101' strings,
coefficients as [re, im]) because 1,300 tiny cases cost 2 MB of zip headers as an npz.
"""
import os, sys, json, warnings, itertools
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: F401
warnings.simplefilter('ignore')
import numpy as np
from symmer.operators import PauliwordOp

OUT = os.path.join(HERE, '..', '..', 'tests', 'golden')
cases = []


def bits(m):
    return [''.join('1' if b else '0' for b in row) for row in np.asarray(m, dtype=bool).reshape(-1, 4)]


def cplx(v):
    return [[float(np.real(c)), float(np.imag(c))] for c in np.asarray(v, dtype=complex).ravel()]


def add(kind, in_symp, in_coeff, n_qubits, q, angles, out_symp, out_coeff, same_object):
    cases.append({'kind': kind, 'in_symp': bits(in_symp), 'in_coeff': cplx(in_coeff), 'q': bits(q), 'angles': [float(a) for a in angles],
                  'out_symp': bits(out_symp), 'out_coeff': cplx(out_coeff), 'same_object': bool(same_object)})


def op(strings, coeffs):
    return PauliwordOp.from_list(strings, coeffs)


starts = {
    'X-X': op(['XI', 'XI'], [1, -1]),                                   # anticommuting pair that cancels
    'X-X,Z-Z': op(['XI', 'ZI', 'XI', 'ZI'], [1, 0.5, -1, -0.5]),        # commuting rows cancel as well
    'X-X,Z': op(['XI', 'XI', 'ZI'], [1, -1, 0.5]),                      # a survivor
    'zeroI': op(['II'], [0]),
    'empty': PauliwordOp(np.zeros((0, 4), dtype=bool), []),
    'X,Y cancel under rotation': op(['XI', 'YI'], [np.cos(0.3), -np.sin(0.3)]),   # P cos + PQ(-i sin) terms meet
    'tiny': op(['XI', 'IZ'], [0.5e-15, 0.25e-15]),                      # every coefficient below the threshold
}
qs = {'Z0': op(['ZI'], [1]), 'X1': op(['IX'], [1]), 'Y0': op(['YI'], [1])}
angle_pool = (np.pi, 0.3, np.pi / 2, 3 * np.pi / 2, 0.0)
for sname, P in starts.items():
    for qname, Q in qs.items():
        for ang in angle_pool:
            R = P._rotate_by_single_Pword(Q, ang)
            add(kind='single', in_symp=P.symp_matrix, in_coeff=np.asarray(P.coeff_vec, dtype=complex), n_qubits=2,
                q=Q.symp_matrix, angles=np.array([ang]), out_symp=R.symp_matrix.reshape(-1, 4),
                out_coeff=np.asarray(R.coeff_vec, dtype=complex), same_object=np.array(R is P))
        for K in (1, 2, 3, 4):
            pool = itertools.product((np.pi, 0.3, np.pi / 2), repeat=K) if K < 4 else [(a,) * 4 for a in (np.pi, 0.3, np.pi / 2)] + [(np.pi / 2, 0.3) * 2]
            for angs in pool:
                R = P.perform_rotations([(Q, a) for a in angs])
                add(kind='chain', in_symp=P.symp_matrix, in_coeff=np.asarray(P.coeff_vec, dtype=complex), n_qubits=2,
                    q=np.vstack([Q.symp_matrix] * K), angles=np.array(angs), out_symp=R.symp_matrix.reshape(-1, 4),
                    out_coeff=np.asarray(R.coeff_vec, dtype=complex), same_object=np.array(False))
    # rotations by different generators in one call
    for angs in itertools.product((np.pi, 0.3, np.pi / 2), repeat=3):
        gens = [qs['Z0'], qs['X1'], qs['Y0']]
        R = P.perform_rotations(list(zip(gens, angs)))
        add(kind='chain', in_symp=P.symp_matrix, in_coeff=np.asarray(P.coeff_vec, dtype=complex), n_qubits=2,
            q=np.vstack([g.symp_matrix for g in gens]), angles=np.array(angs), out_symp=R.symp_matrix.reshape(-1, 4),
            out_coeff=np.asarray(R.coeff_vec, dtype=complex), same_object=np.array(False))
with open(os.path.join(OUT, 'rotate_empty.json'), 'w') as f:
    json.dump({'n_qubits': 2, 'cases': cases}, f, separators=(',', ':'))
print('rotate_empty:', len(cases), 'cases,', os.path.getsize(os.path.join(OUT, 'rotate_empty.json')), 'bytes')
