# ad-hoc: fused product + cleanup of DUPLICATE-HEAVY operands (every product row occurs many times), lazy flow against the filed one
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from symmer_amd import kernels, _lib
from symmer_amd.kernels import DeviceOp
lib = _lib.lib()
def timed(fn, reps=5):
    fn(); kernels.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    kernels.sync(); return (time.perf_counter() - t0) / reps
rng = np.random.default_rng(2)
for n, N, pool in ((1000, 3000, 40), (1000, 10000, 300), (100, 10000, 10000), (12, 8000, 8000)):
    base = DeviceOp.random(pool, n, 0.3, seed=11)
    rows, _ = base.download()
    A = DeviceOp.upload(rows[rng.integers(0, pool, N)], rng.standard_normal(N) + 0j)
    out = {}
    for env in (None, '0'):
        if env is None: os.environ.pop('SYMGPU_CLEANUP_LAZY', None)
        else: os.environ['SYMGPU_CLEANUP_LAZY'] = env
        def run():
            h = ctypes.c_void_p(); _lib.check(lib.symgpu_mul_cleanup_dev(A.handle, A.handle, 1, 1e-15, 1, ctypes.byref(h))); r = DeviceOp(h); run.n = r.n_terms; r.free()
        out[env] = timed(run)
    os.environ.pop('SYMGPU_CLEANUP_LAZY', None)
    print(f'P*P n={n} N={N} (rows drawn from {pool}): {N*N:.1e} pairs -> {run.n} terms: lazy {out[None]*1e3:.3f} ms, filed {out["0"]*1e3:.3f} ms', flush=True)
    A.free(); base.free()
