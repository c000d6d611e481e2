# ad-hoc timing (not a test): the drop-in API on operators of the size the reference's users have (4-30 qubits, 10^2-10^4 terms; SURVEY §8d note,
# /root/reference/tests/hamiltonian_data): per-call medians, operands resident from the second call on
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from symmer_amd import PauliwordOp, IndependentOp, kernels

def med(fn, reps=30):
    fn(); fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); kernels.sync(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e6

rng = np.random.default_rng(1)
for n, T in ((4, 15), (12, 631), (20, 2951), (30, 20000), (30, 1001)):
    P = PauliwordOp.random(n, T).cleanup()
    Q = PauliwordOp.random(n, min(T, 200)).cleanup()
    R1 = PauliwordOp.from_list(['X' + 'I' * (n - 1)])
    out = {
        'P*Q': med(lambda: P * Q), 'P+Q': med(lambda: P + Q), 'P*2': med(lambda: P * 2.0), 'cleanup': med(lambda: P.cleanup()),
        'commutes(P,Q)': med(lambda: P.commutes_termwise(Q)), 'rot(0.3)': med(lambda: P._rotate_by_single_Pword(R1, 0.3)),
        'rot(pi/2)': med(lambda: P._rotate_by_single_Pword(R1, np.pi / 2)), 'P==P': med(lambda: P == P, 5),
        'symmetry_generators': med(lambda: IndependentOp.symmetry_generators(P, commuting_override=True), 5),
        'construct+P*Q+arrays': med(lambda: (lambda R: (R.symp_matrix, R.coeff_vec))(PauliwordOp(P.symp_matrix, P.coeff_vec.copy()) * Q), 10),
    }
    print(f'n={n} T={P.n_terms}: ' + '  '.join(f'{k} {v:.0f}us' for k, v in out.items()), flush=True)
