# ad-hoc timing (not a test): the Four-Russians commutation kernel's tile height (R = 16 / 24 / 48 rows per 16-lane slot) and launch
# (s = stream-K persistent workgroups, o = one tile per workgroup) over operator length n and term count N, next to what the library
# picks by itself (`default`, csrc/commute_m4r.hip m4r_pick + commute_m4r7.hip launch_m7s) and the register-tile kernel (commute.hip).
# N x N np.bool_ table, operands resident, milliseconds per call.
#     python tools/bench_m4r_pick.py [n ...]  > profiles/rNN_m4r_pick.txt
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from symmer_amd import kernels, _lib
from symmer_amd.kernels import DeviceOp

lib = _lib.lib()
QUBITS = [int(a) for a in sys.argv[1:]] or [20, 100, 150, 200, 250, 300, 400, 500, 700, 1000, 2000]
TERMS = [20000, 30000, 50000, 100000]
SWITCHES = ('SYMGPU_COMMUTE_M4R', 'SYMGPU_M4R_R', 'SYMGPU_M4R_STREAM')


def timed(fn, reps=8):
    fn(); fn(); kernels.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    kernels.sync()
    return (time.perf_counter() - t0) / reps * 1e3


print('n N ' + ' '.join(f'R{r}{m}' for r in (16, 24, 48) for m in 'so') + ' default regtile  default/best')
for n in QUBITS:
    for N in TERMS:
        if n >= 1000 and N > 70000:
            continue
        C = DeviceOp.random(N, n, 0.3, seed=1239)
        buf = ctypes.c_void_p()
        _lib.check(lib.symgpu_dev_alloc(N * N + 64, ctypes.byref(buf)))
        call = lambda: _lib.check(lib.symgpu_commutes_dev(C.handle, 0, N, C.handle, buf))
        cells = []
        os.environ['SYMGPU_COMMUTE_M4R'] = '1'
        for r in ('16', '24', '48'):
            for st in ('1', '0'):
                os.environ['SYMGPU_M4R_R'] = r; os.environ['SYMGPU_M4R_STREAM'] = st
                cells.append(timed(call))
        for k in SWITCHES:
            os.environ.pop(k, None)
        t_default = timed(call)
        os.environ['SYMGPU_COMMUTE_M4R'] = '0'
        t_reg = timed(call)
        os.environ.pop('SYMGPU_COMMUTE_M4R')
        print(f'{n} {N} ' + ' '.join(f'{t:.3f}' for t in cells) + f' {t_default:.3f} {t_reg:.3f}  {t_default / min(cells + [t_reg]):.3f}', flush=True)
        _lib.check(lib.symgpu_dev_free(buf)); C.free()
