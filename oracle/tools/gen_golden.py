"""Generate ``tests/golden/*.npz`` by RUNNING THE REFERENCE (imported through ``ref_shim``).

BUILD CONTAINER ONLY (needs /root/reference).  The committed fixtures hold data only — seeded inputs
and the outputs the reference produced for them — so the GPU box (which has no reference) can check
both the oracle and the HIP path against them.

Run:  python oracle/tools/gen_golden.py        (deterministic; rewrites tests/golden/)

Families:
  known      — known answers the reference's own tests assert (tests/test_operators/test_base.py:510-515,
               537-552, 554-579, 596-613; base.py:944-951; test_independent_op.py:5-23,50-66,97-104)
  mul        — A*B incl. operand swap, dyadic coefficients (bit-exact) and Gaussian (1e-12)
  cleanup    — cleanup/add/sub incl. duplicate-heavy, all-cancelling and empty inputs
  commute    — commutes_termwise / adjacency_matrix
  rotate     — _rotate_by_single_Pword (Clifford multiples -2..5, non-Clifford) and perform_rotations
  gf2        — _rref_binary / rref_binary / _cref_binary / cref_binary
  symgen     — IndependentOp.symmetry_generators (H2, planted symmetries, H3+ JW Hamiltonian),
               PauliwordOp.generators, generator_reconstruction
"""
import os, sys, json, warnings
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: F401
warnings.simplefilter('ignore')
import numpy as np
from symmer.operators import PauliwordOp, IndependentOp
from symmer.operators.utils import _rref_binary, rref_binary, _cref_binary, cref_binary, symplectic_cleanup

OUT = os.path.join(HERE, '..', '..', 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)
rng = np.random.default_rng(1234)


def dyadic(t):
    return (rng.integers(-8, 9, t) + 1j * rng.integers(-8, 9, t)) / 16.0


def gauss(t):
    return rng.standard_normal(t) + 1j * rng.standard_normal(t)


def rand_symp(n, t, density=0.3):
    return rng.random((t, 2 * n)) < density


class Family:
    def __init__(self, name):
        self.name, self.d, self.k = name, {}, 0

    def add(self, **arrays):
        for key, val in arrays.items():
            a = np.asarray(val)
            if a.dtype == bool:
                a = a.astype(np.uint8)
            self.d[f'{self.k:04d}/{key}'] = a
        self.k += 1

    def save(self):
        self.d['n_cases'] = np.array(self.k)
        path = os.path.join(OUT, f'{self.name}.npz')
        np.savez_compressed(path, **self.d)
        print(f'{self.name}: {self.k} cases, {os.path.getsize(path) / 1024:.1f} kB')


def op_arrays(P, prefix):
    return {f'{prefix}_symp': P.symp_matrix, f'{prefix}_coeff': np.asarray(P.coeff_vec, dtype=complex)}


# ---------------------------------------------------------------- known answers ----------------
known = {}
# fixtures pauli_list_1 / pauli_list_2 of test_base.py:211-216 (data), evaluated through the reference API
pl1 = ['III', 'XXX', 'YYY', 'ZZZ']; pl2 = ['ZXZ', 'XZX', 'XYZ', 'ZIX']
P = PauliwordOp.from_list(pl1)
known['ycount_symp'] = P.symp_matrix.astype(np.uint8); known['ycount'] = np.asarray(P.Y_count)   # == [0,0,3,0], :510-515
P1 = PauliwordOp.from_list(pl1); P2 = PauliwordOp.from_list(pl2)
known['pl1_symp'] = P1.symp_matrix.astype(np.uint8); known['pl2_symp'] = P2.symp_matrix.astype(np.uint8)
known['pl1_commutes_pl2'] = P1.commutes_termwise(P2).astype(np.uint8)                              # table at :554-566
known['pl2_adjacency'] = P2.adjacency_matrix.astype(np.uint8)                                      # table at :568-579
assert np.array_equal(known['pl1_commutes_pl2'], np.array([[1,1,1,1],[1,0,1,0],[0,0,1,1],[0,1,1,0]]))
assert np.array_equal(known['pl2_adjacency'], np.array([[1,0,1,0],[0,1,1,0],[1,1,1,1],[0,0,1,1]]))
P = PauliwordOp.from_list(['XXX', 'YYY', 'XXX', 'YYY'], [1, 1, -1, 1]).cleanup()      # :537-544
known['cleanup_in_symp'] = PauliwordOp.from_list(['XXX', 'YYY', 'XXX', 'YYY']).symp_matrix.astype(np.uint8)
known['cleanup_in_coeff'] = np.array([1, 1, -1, 1], dtype=complex)
known['cleanup_out_symp'] = P.symp_matrix.astype(np.uint8); known['cleanup_out_coeff'] = P.coeff_vec
op1 = PauliwordOp.from_list(['XYXZ', 'YYII']); op2 = PauliwordOp.from_list(['YYZZ', 'XIXZ', 'XZZI'])   # base.py:944-951
known['doc_a'] = op1.symp_matrix.astype(np.uint8); known['doc_b'] = op2.symp_matrix.astype(np.uint8)
known['doc_commutes'] = op1.commutes_termwise(op2).astype(np.uint8)
sq = {}
for a, b in (('X', 'Y'), ('Z', 'X'), ('Y', 'Z'), ('Y', 'X'), ('X', 'Z'), ('Z', 'Y')):                  # :596-613
    r = PauliwordOp.from_dictionary({a: 1}) * PauliwordOp.from_dictionary({b: 1})
    sq[a + b] = [''.join('IXZY'[int(x) + 2 * int(z)] for x, z in zip(r.X_block[0], r.Z_block[0])),
                 [float(r.coeff_vec[0].real), float(r.coeff_vec[0].imag)]]
# Appendix-B style small products / rotations
A = PauliwordOp.from_dictionary({'XI': 1, 'ZZ': 2, 'YX': 3}); B = PauliwordOp.from_dictionary({'IY': 5, 'XI': 7j})
for nm, R in (('AB', A * B), ('BA', B * A)):
    known[f'small_{nm}_symp'] = R.symp_matrix.astype(np.uint8); known[f'small_{nm}_coeff'] = R.coeff_vec
known['small_A_symp'] = A.symp_matrix.astype(np.uint8); known['small_A_coeff'] = A.coeff_vec
known['small_B_symp'] = B.symp_matrix.astype(np.uint8); known['small_B_coeff'] = B.coeff_vec
H2 = {'IIII': -0.09706626816762845, 'IIIZ': -0.22343153690813597, 'IIZI': -0.22343153690813597,
      'IIZZ': 0.17441287612261608, 'IZII': 0.17141282644776884, 'IZIZ': 0.12062523483390426,
      'IZZI': 0.16592785033770355, 'ZIII': 0.17141282644776884, 'ZIIZ': 0.16592785033770355,
      'ZIZI': 0.12062523483390426, 'ZZII': 0.16868898170361213, 'XXYY': -0.0453026155037993,
      'XYYX': 0.0453026155037993, 'YXXY': 0.0453026155037993, 'YYXX': -0.0453026155037993}   # test_independent_op.py:5-23
H2op = PauliwordOp.from_dictionary(H2)
S = IndependentOp.symmetry_generators(H2op)
known['H2_symp'] = H2op.symp_matrix.astype(np.uint8); known['H2_coeff'] = H2op.coeff_vec
known['H2_symgen'] = S.symp_matrix.astype(np.uint8)
op = PauliwordOp.from_list(['IZZ', 'ZZI', 'IXX', 'XXI', 'IYY', 'YYI'])                        # :56-66
known['override_symp'] = op.symp_matrix.astype(np.uint8)
known['override_symgen'] = IndependentOp.symmetry_generators(op, commuting_override=True).symp_matrix.astype(np.uint8)
op = PauliwordOp.from_list(['X', 'Y', 'Z'])                                                    # :50-53
known['nosym_symp'] = op.symp_matrix.astype(np.uint8)
known['nosym_symgen'] = IndependentOp.symmetry_generators(op).symp_matrix.astype(np.uint8).reshape(0, 2)
np.savez_compressed(os.path.join(OUT, 'known.npz'), **known)
with open(os.path.join(OUT, 'known_single_qubit.json'), 'w') as f:
    json.dump(sq, f, indent=1)
print('known:', len(known), 'arrays')

# ---------------------------------------------------------------- mul ---------------------------
fam = Family('mul')
for n in (1, 3, 63, 64, 65, 100, 130):
    for (N, M) in ((1, 1), (2, 7), (7, 2), (64, 5), (5, 64), (40, 40)):
        A = PauliwordOp(rand_symp(n, N), dyadic(N)); B = PauliwordOp(rand_symp(n, M), dyadic(M))
        fam.add(**op_arrays(A, 'a'), **op_arrays(B, 'b'), **op_arrays(A * B, 'out'), exact=1)
for n, N, M in ((3, 500, 500), (2, 64, 300), (100, 120, 120)):        # duplicate-heavy / square
    A = PauliwordOp(rand_symp(n, N), dyadic(N))
    B = A if N == M else PauliwordOp(rand_symp(n, M), dyadic(M))
    fam.add(**op_arrays(A, 'a'), **op_arrays(B, 'b'), **op_arrays(A * B, 'out'), exact=1)
for n, N, M in ((10, 50, 30), (70, 30, 50), (100, 64, 64)):           # Gaussian: tolerance rule
    A = PauliwordOp(rand_symp(n, N), gauss(N)); B = PauliwordOp(rand_symp(n, M), gauss(M))
    fam.add(**op_arrays(A, 'a'), **op_arrays(B, 'b'), **op_arrays(A * B, 'out'), exact=0)
# empty operands (SURVEY §8a')
A = PauliwordOp(rand_symp(5, 4), dyadic(4)); E = PauliwordOp(np.zeros((0, 10), dtype=bool), [])
fam.add(**op_arrays(A, 'a'), **op_arrays(E, 'b'), **op_arrays(A * E, 'out'), exact=1)
fam.add(**op_arrays(E, 'a'), **op_arrays(A, 'b'), **op_arrays(E * A, 'out'), exact=1)
fam.save()

# ---------------------------------------------------------------- cleanup -----------------------
fam = Family('cleanup')
for n, t, dens in ((1, 9, 0.5), (3, 500, 0.3), (2, 300, 0.5), (64, 200, 0.02), (65, 100, 0.3), (130, 64, 0.3)):
    symp = rand_symp(n, t, dens); c = dyadic(t)
    P = PauliwordOp(symp, c)
    fam.add(**op_arrays(P, 'in'), **op_arrays(P.cleanup(), 'out'), thr=1e-15)
    r = symplectic_cleanup(symp, c, zero_threshold=None)
    fam.add(**op_arrays(P, 'in'), out_symp=r[0], out_coeff=r[1], thr=-1.0)
    Q = PauliwordOp(rand_symp(n, t // 2 + 1, dens), dyadic(t // 2 + 1))
    fam.add(**op_arrays(P.append(Q), 'in'), **op_arrays(P + Q, 'out'), thr=1e-15)
    D = P - P
    fam.add(in_symp=np.vstack([symp, symp]), in_coeff=np.hstack([c, -c]), **op_arrays(D, 'out'), thr=1e-15)
g = gauss(40); symp = rand_symp(3, 40)
P = PauliwordOp(symp, g)
fam.add(**op_arrays(P, 'in'), **op_arrays(P.cleanup(), 'out'), thr=1e-15)
E = PauliwordOp(np.zeros((0, 6), dtype=bool), [])
fam.add(**op_arrays(E, 'in'), **op_arrays(E.cleanup(), 'out'), thr=1e-15)
fam.save()

# ---------------------------------------------------------------- commute -----------------------
fam = Family('commute')
for n in (1, 3, 63, 64, 65, 100, 130, 200):
    for (N, M) in ((1, 1), (2, 7), (64, 5), (5, 64), (70, 70)):
        A = PauliwordOp(rand_symp(n, N), np.ones(N)); B = PauliwordOp(rand_symp(n, M), np.ones(M))
        fam.add(a_symp=A.symp_matrix, b_symp=B.symp_matrix, out=A.commutes_termwise(B), adj=A.adjacency_matrix)
A = PauliwordOp(rand_symp(5, 4), np.ones(4)); E = PauliwordOp(np.zeros((0, 10), dtype=bool), [])
fam.add(a_symp=A.symp_matrix, b_symp=E.symp_matrix, out=A.commutes_termwise(E).reshape(4, 0), adj=A.adjacency_matrix)
fam.save()

# ---------------------------------------------------------------- rotate ------------------------
fam = Family('rotate')
angles = (0.3, -1.1, 2.0, 0.0, np.pi / 2, -np.pi / 2, np.pi, -np.pi, 3 * np.pi / 2, 2 * np.pi, 5 * np.pi / 2)
for trial in range(24):
    n = int((1, 2, 5, 33, 64, 65, 70, 130)[trial % 8]); t = int(rng.integers(1, 120))
    P0 = PauliwordOp(rand_symp(n, t), dyadic(t)).cleanup()
    q = rng.random(2 * n) < 0.4
    if trial % 2 == 0 and P0.n_terms > 2:
        half = P0.n_terms // 2
        P0 = P0.append(PauliwordOp(P0.symp_matrix[:half] ^ q, dyadic(half))).cleanup()
    if P0.n_terms == 0:
        continue
    Q = PauliwordOp(q.reshape(1, -1), [1])
    for ang in angles:
        R = P0._rotate_by_single_Pword(Q, ang)
        fam.add(**op_arrays(P0, 'in'), q=q, angle=float(ang), **op_arrays(R, 'out'), chain=0)
    rots = [(PauliwordOp((rng.random(2 * n) < 0.4).reshape(1, -1), [1]), float(a)) for a in (np.pi / 2, 0.3, np.pi / 2, -0.7)]
    R = P0.perform_rotations(rots)
    fam.add(**op_arrays(P0, 'in'), q=np.vstack([r.symp_matrix for r, _ in rots]), angle=np.array([a for _, a in rots]),
            **op_arrays(R, 'out'), chain=1)
fam.save()

# ---------------------------------------------------------------- gf2 ---------------------------
fam = Family('gf2')
for R in (1, 5, 63, 64, 65, 200):
    for C in (1, 5, 63, 64, 65, 200):
        for dens in (0.5, 0.05):
            m = rng.random((R, C)) < dens
            if R > 3:
                m[R // 2] = False; m[R - 1] = m[0]
            if not m.any():
                m[0, 0] = True
            fam.add(m=np.packbits(m, axis=1), shape=np.array(m.shape), rref_noswap=np.packbits(_rref_binary(m), axis=1),
                    rref=np.packbits(rref_binary(m), axis=1), cref_noswap=np.packbits(_cref_binary(m), axis=1),
                    cref=np.packbits(cref_binary(m), axis=1))
fam.save()

# ---------------------------------------------------------------- symgen ------------------------
fam = Family('symgen')
cases = [(4, 10, 2), (10, 40, 3), (40, 200, 5), (70, 100, 8), (64, 130, 4), (130, 300, 7)]
for n, t, k in cases:
    symp = rand_symp(n, t); symp[:, :k] = False
    P = PauliwordOp(symp, dyadic(t))
    # scramble with Clifford rotations as symmer/utils.py:141-149 does
    rots = [(PauliwordOp((rng.random(2 * n) < 0.3).reshape(1, -1), [1]), None) for _ in range(6)]
    P = P.perform_rotations(rots)
    S = IndependentOp.symmetry_generators(P, commuting_override=True)
    G = P.generators
    Rm, mask = P.generator_reconstruction(G)
    fam.add(h_symp=P.symp_matrix, symgen=S.symp_matrix, planted=k, gens=G.symp_matrix, recon=Rm, recon_mask=mask)
with open('/root/reference/tests/hamiltonian_data/H3+_STO-3G_SINGLET_JW.json') as f:
    ham = json.load(f)['hamiltonian']
P = PauliwordOp.from_dictionary({k: complex(*v) if isinstance(v, (list, tuple)) else v for k, v in ham.items()})
S = IndependentOp.symmetry_generators(P, commuting_override=True)
G = P.generators
Rm, mask = P.generator_reconstruction(G)
fam.add(h_symp=P.symp_matrix, symgen=S.symp_matrix, planted=-1, gens=G.symp_matrix, recon=Rm, recon_mask=mask)
fam.save()


# ---------------------------------------------------------------- taper --------
from symmer.projection import QubitTapering
from symmer.operators.utils import check_adjmat_noncontextual
fam = Family('taper')
HAM = '/root/reference/tests/hamiltonian_data/'
def load_ham(name):
    with open(HAM + name) as f:
        d = json.load(f)
    H = PauliwordOp.from_dictionary({k: complex(*v) for k, v in d['hamiltonian'].items()})
    return H, np.array(d['data']['hf_array'])
for name in ('H3+_STO-3G_SINGLET_JW.json', 'H3+_STO-3G_SINGLET_BK.json', 'H4_STO-3G_SINGLET_JW.json', 'H4_STO-3G_SINGLET_BK.json',
             'O_STO-3G_TRIPLET_JW.json'):
    H, hf = load_ham(name)
    for target in ('Z', 'X'):
        T = QubitTapering(H, target_sqp=target)
        out = T.taper_it(ref_state=hf)
        rots = np.vstack([r.symp_matrix for r, _ in T.stabilizers.stabilizer_rotations]) if T.stabilizers.stabilizer_rotations else np.zeros((0, 2 * H.n_qubits), dtype=bool)
        fam.add(h_symp=H.symp_matrix, h_coeff=H.coeff_vec, hf=hf, target=np.array(ord(target)), stab_symp=T.stabilizers.symp_matrix,
                stab_coeff=np.asarray(T.stabilizers.coeff_vec), rotations=rots, rotated_stab=T.rotated_stabilizers.symp_matrix,
                rotated_stab_coeff=np.asarray(T.rotated_stabilizers.coeff_vec), free=T.free_qubit_indices,
                **op_arrays(out, 'out'))
        sector = -np.asarray(T.stabilizers.coeff_vec)
        T2 = QubitTapering(H, target_sqp=target)
        out2 = T2.taper_it(sector=sector)
        fam.add(h_symp=H.symp_matrix, h_coeff=H.coeff_vec, hf=np.zeros(0, dtype=int), target=np.array(ord(target)), stab_symp=T2.stabilizers.symp_matrix,
                stab_coeff=np.asarray(T2.stabilizers.coeff_vec), rotations=rots, rotated_stab=T2.rotated_stabilizers.symp_matrix,
                rotated_stab_coeff=np.asarray(T2.rotated_stabilizers.coeff_vec), free=T2.free_qubit_indices,
                **op_arrays(out2, 'out'))
fam.save()

# noncontextuality known answers (tests/test_operators/test_base.py:581-595) + random adjacency checks
nc = {}
cases = [(['XZ', 'ZX', 'ZI', 'IZ'], False), (['XZ', 'ZX', 'XX', 'YY'], True), (['XX', 'YY', 'ZZ', 'II'], True),
         (['II', 'ZZ', 'ZX', 'ZY', 'XZ', 'YZ', 'XX', 'XY', 'YX', 'YY'], False), (['III', 'IIZ', 'ZII', 'IXZ', 'IYZ', 'YYZ'], False),
         (['IZI', 'ZII', 'IIY', 'ZZY', 'XXZ', 'XYZ', 'YXZ', 'YYZ', 'XXX', 'XYX', 'YXX', 'YYX'], True)]
out_nc = []
for plist, expect in cases:
    P = PauliwordOp.from_list(plist)
    assert bool(P.is_noncontextual) == expect
    out_nc.append([plist, bool(P.is_noncontextual)])
for trial in range(20):
    P = PauliwordOp(rand_symp(int(rng.integers(2, 6)), int(rng.integers(4, 12)), 0.4), np.ones(1)).cleanup() if False else None
with open(os.path.join(OUT, 'known_noncontextual.json'), 'w') as f:
    json.dump(out_nc, f)
print('noncontextual:', len(out_nc), 'cases')


# ---------------------------------------------------------------- circuit --------
from symmer.evolution.circuit_symmerlator import CircuitSymmerlator
fam = Family('circuit')
G1 = ['x', 'y', 'z', 'h', 's', 'sdg', 'sx', 'sy', 'sz']; G2 = ['cx', 'cy', 'cz', 'swap']; GR = ['rx', 'ry', 'rz']
for trial in range(12):
    n = int((2, 3, 5, 8, 20, 70)[trial % 6]); depth = int(rng.integers(5, 60))
    cs = CircuitSymmerlator(n)
    names, qa, qb, ang = [], [], [], []
    for d in range(depth):
        r = rng.random()
        if r < 0.55 or n < 2:
            g = G1[int(rng.integers(len(G1)))]; a = int(rng.integers(n)); cs.gate_map[g](a); names.append(g); qa.append(a); qb.append(-1); ang.append(0.0)
        elif r < 0.9:
            g = G2[int(rng.integers(len(G2)))]; a, b = (int(v) for v in rng.choice(n, 2, replace=False)); cs.gate_map[g](a, b)
            names.append(g); qa.append(a); qb.append(b); ang.append(0.0)
        elif trial % 2 == 0:
            g = GR[int(rng.integers(len(GR)))]; a = int(rng.integers(n)); t = float(rng.normal()); cs.gate_map[g](a, t)
            names.append(g); qa.append(a); qb.append(-1); ang.append(t)
    O = PauliwordOp(rand_symp(n, int(rng.integers(1, 30))), dyadic(1)[0] * np.ones(1) if False else dyadic(1)).cleanup() if False else None
    t_op = int(rng.integers(1, 30))
    O = PauliwordOp(rand_symp(n, t_op, 0.4), dyadic(t_op)).cleanup()
    if O.n_terms == 0:
        continue
    R = cs.apply_sequence(O)
    fam.add(n=n, gates=np.array(names), qa=np.array(qa), qb=np.array(qb), angle=np.array(ang), **op_arrays(O, 'in'), **op_arrays(R, 'out'),
            expval=np.array(complex(cs.evaluate(O))))
fam.save()


# ---------------------------------------------------------------- state --------
from symmer.operators import QuantumState
from symmer.operators.base import single_term_expval
fam = Family('state')
np.random.seed(4321)
for n, t_op, t_psi in ((1, 2, 2), (3, 6, 4), (5, 40, 12), (8, 30, 60), (20, 50, 30), (70, 25, 15), (70, 8, 40)):
    P = PauliwordOp(rand_symp(n, t_op, 0.4), dyadic(t_op)).cleanup()
    psi = QuantumState.random(n, t_psi); phi = QuantumState.random(n, t_psi + 3)
    out = P * psi
    bra = psi.dagger * P
    fam.add(n=n, **op_arrays(P, 'p'), psi_m=psi.state_matrix, psi_c=psi.state_op.coeff_vec, phi_m=phi.state_matrix, phi_c=phi.state_op.coeff_vec,
            out_m=out.state_matrix, out_c=out.state_op.coeff_vec, bra_m=bra.state_matrix, bra_c=bra.state_op.coeff_vec,
            inner=np.array(complex(psi.dagger * phi)), expval=np.array(complex(P.expval(psi))),
            term_expvals=np.array([single_term_expval(Pk, psi) for Pk in P]))
fam.save()


# ---------------------------------------------------------------- jordan / reindex (host glue on the GF(2) + commutation kernels) --------
from symmer.operators.utils import check_jordan_independent
rj = np.random.default_rng(4242)
fam = Family('jordan')
# (1) known answers of tests/test_operators/test_operator_utils.py:4-75
for plist, expect in [(['XXX', 'XII', 'IIX', 'IXI', 'ZZI', 'IYY'], False), (['IX', 'IY', 'IZ', 'ZI', 'YI', 'XI'], True),
                      (['IX', 'IY', 'IZ', 'ZI', 'YI', 'XI', 'XX'], False), (['XX', 'YY', 'ZZ'], False),
                      (['XZXIIIZI', 'IZZIZZZX', 'IXXXZXZI', 'IIZIIXZX', 'XIXIIIIZ', 'ZYIIZZIY', 'IIIXIIII', 'ZIZIIYZZ', 'IIZIIIXY'], True)]:
    P = PauliwordOp.from_list(plist)
    assert bool(check_jordan_independent(P)) == expect
    fam.add(kind=np.array(0), symp=P.symp_matrix, expect=np.array(expect))
# (2) random operators judged by the reference
for trial in range(30):
    n = int(rj.integers(1, 7)); t = int(rj.integers(1, 3 * n + 2))
    P = PauliwordOp((rj.random((t, 2 * n)) < 0.4), np.ones(t))
    fam.add(kind=np.array(0), symp=P.symp_matrix, expect=np.array(bool(check_jordan_independent(P))))
# (3) reindex: list and dictionary forms
for trial in range(10):
    n = int(rj.integers(2, 9)); t = int(rj.integers(1, 8))
    P = PauliwordOp((rj.random((t, 2 * n)) < 0.5), rj.integers(-8, 9, t) / 16)
    k = int(rj.integers(2, n + 1))
    chosen = rj.choice(n, k, replace=False)
    shuffled = rj.permutation(chosen)
    if trial % 2 == 0:
        Q = P.reindex([int(v) for v in shuffled])
        fam.add(kind=np.array(1), symp=P.symp_matrix, coeff=P.coeff_vec, keys=np.sort(chosen), vals=shuffled, as_list=np.array(1), out=Q.symp_matrix)
    else:
        Q = P.reindex({int(a): int(b) for a, b in zip(chosen, shuffled)})
        fam.add(kind=np.array(1), symp=P.symp_matrix, coeff=P.coeff_vec, keys=chosen, vals=shuffled, as_list=np.array(0), out=Q.symp_matrix)
# (4) jordan_generator_reconstruction: symmetry generators + one anticommuting clique (docstring example of utils.py:533-541 and random)
def jordan_case(G, n_terms):
    sym = np.all(G.commutes_termwise(G), axis=1)
    S, A = G[sym], G[~sym]
    rows = []
    for _ in range(n_terms):
        term = PauliwordOp.from_list(['I' * G.n_qubits])
        for s in S:
            if rj.random() < 0.5:
                term = term * s
        if A.n_terms and rj.random() < 0.7:
            term = term * A[int(rj.integers(A.n_terms))]
        rows.append(term.symp_matrix[0])
    extra = (rj.random((2, 2 * G.n_qubits)) < 0.5)                  # most likely not reconstructable
    H = PauliwordOp(np.vstack(rows + [extra]), np.ones(n_terms + 2))
    R, mask = H.jordan_generator_reconstruction(G)
    fam.add(kind=np.array(2), g_symp=G.symp_matrix, h_symp=H.symp_matrix, R=np.asarray(R), mask=np.asarray(mask))
jordan_case(PauliwordOp.from_list(['IIZI', 'ZIIZ', 'IXII', 'IIIZ', 'XIIX']), 12)
jordan_case(PauliwordOp.from_list(['ZII', 'IZI', 'IIZ']), 6)
jordan_case(PauliwordOp.from_list(['ZIIII', 'IZIII', 'IIXII', 'IIZXX', 'IIYXX', 'IIIZI']), 15) if check_jordan_independent(PauliwordOp.from_list(['ZIIII', 'IZIII', 'IIXII', 'IIZXX', 'IIYXX', 'IIIZI'])) else None
fam.save()
