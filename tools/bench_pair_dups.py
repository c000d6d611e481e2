# ad-hoc timing (not a test): product + cleanup with the duplicate pairs found from the operand hash tables (pair_dups.hip, the default
# where it applies) against the partial sort + k_find_suspects (SYMGPU_CLEANUP_DIRECT=0), squared operators and general products, ms per call
#     python tools/bench_pair_dups.py [n_qubits]
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from symmer_amd import kernels
from symmer_amd.kernels import DeviceOp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000


def timed(fn, reps=6):
    r = fn(); kernels.sync(); r.free()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); kernels.sync(); ts.append(time.perf_counter() - t0); r.free()
    return sorted(ts)[len(ts) // 2] * 1e3


print(f'n = {n} qubits: shape, keys, ms direct, ms sorted flag pass, terms out')
for N, M in ((1500, 0), (2000, 0), (3000, 0), (5000, 0), (7000, 0), (10000, 0), (600, 600), (1000, 1000), (2000, 2000), (3000, 3000), (5000, 5000), (8000, 3000), (20000, 500)):
    A = DeviceOp.random(N, n, 0.3, seed=5 + N)
    B = A if M == 0 else DeviceOp.random(M, n, 0.3, seed=9 + M)
    keys = N * (N + 1) // 2 if M == 0 else N * M
    out = []
    for env in ('1', '0'):
        os.environ['SYMGPU_CLEANUP_DIRECT'] = env
        out.append(timed(lambda: kernels.mul_cleanup_handles(A, B, True, 1e-15)))
    os.environ.pop('SYMGPU_CLEANUP_DIRECT')
    r = kernels.mul_cleanup_handles(A, B, True, 1e-15)
    print(f"{'P*P ' + str(N) if M == 0 else f'{N} x {M}'}: {keys:.3g} keys  {out[0]:.3f}  {out[1]:.3f}  ({r.n_terms} terms)", flush=True)
    r.free(); A.free()
    if M:
        B.free()
