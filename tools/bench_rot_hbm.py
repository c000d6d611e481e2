# ad-hoc timing (not a test): the one-launch rotation of operators beyond the chip's 38 MB of LDS + registers — rows left in memory and read
# again when they are written (csrc/rotate_resident.hip, `hbm` form) — against the multi-launch kernels (SYMGPU_ROT_RESIDENT=0); us per call,
# non-Clifford (0.3) and Clifford (pi / 2), operands resident, duplicate status and hashes known
#     python tools/bench_rot_hbm.py            (under rocprofv3 --kernel-trace: profiles/r06_rotation_hbm_kernel_trace.txt)
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from symmer_amd import kernels, packing
from symmer_amd.kernels import DeviceOp


def timed(fn, reps=30):
    fn(); kernels.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    kernels.sync()
    return (time.perf_counter() - t0) / reps * 1e6


for n, N in ((2000, 100000), (3000, 100000), (1000, 200000), (1000, 400000), (1000, 800000), (500, 1000000)):
    A = DeviceOp.random(N, n, 0.3, seed=1000 + n)
    P = kernels.cleanup_dev(A); A.free()
    rng = np.random.default_rng(7 + n)
    q = packing.pack_rows(rng.random((1, 2 * n)) < 0.3)[0]
    out = []
    for env in (None, '0'):
        if env:
            os.environ['SYMGPU_ROT_RESIDENT'] = env
        else:
            os.environ.pop('SYMGPU_ROT_RESIDENT', None)
        for ang in (0.3, np.pi / 2):
            def rot():
                r, a = kernels.rotate_single_dev(P, q, ang)
                if r is not None:
                    r.free()
            out.append(timed(rot))
    os.environ.pop('SYMGPU_ROT_RESIDENT', None)
    row = 16 * ((n + 63) // 64) + 16
    print(f'n={n} terms={P.n_terms} ({P.n_terms * row / 1e6:.0f} MB): one launch {out[0]:.1f} us (Clifford {out[1]:.1f}), multi-launch {out[2]:.1f} us (Clifford {out[3]:.1f})', flush=True)
    P.free()
