# Shape sweep of the five data-parallel paths away from the BASELINE shapes: product, product + cleanup (squared),
# plain cleanup, commutation, rotation at n in {20, 64, 100, 130, 500, 1000, 1100, 2000, 3000} x N in {10^3, 10^4, 10^5}, each with the
# fraction of the HBM peak its algorithmic bytes reach (SURVEY 8d's bytes per unit).  Run on the GPU box:
#     python tools/sweep_shapes.py > profiles/rNN_sweep.json          (one JSON object; --table prints the DESIGN.md table from it)
# The reference's calls are shape-agnostic (base.py:764-794, utils.py:230-279); its real inputs are 4-30-qubit molecular Hamiltonians.
import sys, os, time, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

HBM = 8.0e12
QUBITS = [20, 64, 100, 130, 500, 1000, 1100, 2000, 3000]
TERMS = [1000, 10000, 100000]


def table(path):
    d = json.load(open(path))
    for wl, unit in (('product', 'pairs/s'), ('mul_cleanup', 'pairs/s'), ('cleanup', 'rows/s'), ('commutes', 'pairs/s'), ('rotation', 'terms/s')):
        print(f'\n**{wl}** — seconds per call (fraction of the 8 TB/s HBM peak on the algorithmic bytes)\n')
        print('| n \\ N | ' + ' | '.join(f'{t:,}' for t in TERMS) + ' |')
        print('|---|' + '---|' * len(TERMS))
        for n in QUBITS:
            cells = []
            for t in TERMS:
                e = d['results'].get(f'{wl}/{n}/{t}')
                cells.append('—' if not e else (e['skipped'] if 'skipped' in e else f"{e['seconds'] * 1e3:.3g} ms ({e['frac']:.2f})"))
            print(f'| {n} | ' + ' | '.join(cells) + ' |')


def main():
    if len(sys.argv) > 2 and sys.argv[1] == '--table':
        return table(sys.argv[2])
    from symmer_amd import kernels, _lib, packing
    from symmer_amd.kernels import DeviceOp
    lib = _lib.lib()
    only = set(sys.argv[1:])
    res = {}

    def timed(fn, reps):
        fn(); kernels.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        kernels.sync()
        return (time.perf_counter() - t0) / reps

    def put(key, seconds, units, nbytes, **kw):
        res[key] = dict(seconds=seconds, units=units, units_per_s=units / seconds, bytes=nbytes, GBps=nbytes / seconds / 1e9, frac=nbytes / seconds / HBM, **kw)
        print(key, json.dumps(res[key]), file=sys.stderr, flush=True)

    for n in QUBITS:
        wq = (n + 63) // 64
        row = 16 * wq + 16
        for N in TERMS:
            A = DeviceOp.random(N, n, 0.3, seed=1000 + n + N)
            # -- all-pairs product, N x N, streamed through a ring of two output slabs of <= 6.5 GB
            if not only or 'product' in only:
                slab = int(max(1, min(N, 6.5e9 // (N * row))))
                if slab >= 16:
                    slab -= slab % 16
                ring = [DeviceOp.alloc(slab * N, wq, with_coeff=True) for _ in range(2)]

                def prod():
                    for k, o0 in enumerate(range(0, N, slab)):
                        _lib.check(lib.symgpu_mul_allpairs_dev(A.handle, A.handle, o0, min(N, o0 + slab), 1, ring[k & 1].handle))
                t = timed(prod, 3 if N >= 100000 else 10)
                put(f'product/{n}/{N}', t, N * N, N * N * row, slab_rows=slab)
                for r in ring:
                    r.free()
            # -- squared product + cleanup (the reference's P * P); 10^5 terms squared = 5 x 10^9 keys: beyond one call
            if not only or 'mul_cleanup' in only:
                if N <= 10000:
                    out = [None]

                    def mc():
                        if out[0] is not None:
                            out[0].free()
                        out[0] = kernels.mul_cleanup_handles(A, A, True, 1e-15)
                    t = timed(mc, 5 if N >= 10000 else 20)
                    n_out = out[0].n_terms
                    out[0].free()
                    put(f'mul_cleanup/{n}/{N}', t, N * N, n_out * row, terms_out=n_out, bytes_are='kept rows + coefficients written (the physical floor; product rows never exist)')
                else:
                    res[f'mul_cleanup/{n}/{N}'] = {'skipped': '1e10 pairs'}
            # -- plain cleanup of N rows holding 2.7 copies of every distinct row
            if not only or 'cleanup' in only:
                rng = np.random.default_rng(n + N)
                D = kernels.op_gather(A, rng.integers(0, max(1, int(N / 2.7)), N))
                out = [None]

                def cl():
                    if out[0] is not None:
                        out[0].free()
                    out[0] = kernels.cleanup_dev(D, 1e-15)
                t = timed(cl, 20)
                n_out = out[0].n_terms
                out[0].free(); D.free()
                put(f'cleanup/{n}/{N}', t, N, (N + n_out) * row, terms_out=n_out)
            # -- commutation table N x N (np.bool_ bytes)
            if not only or 'commutes' in only:
                buf = ctypes.c_void_p()
                _lib.check(lib.symgpu_dev_alloc(N * N, ctypes.byref(buf)))
                t = timed(lambda: _lib.check(lib.symgpu_commutes_dev(A.handle, 0, N, A.handle, buf)), 3 if N >= 100000 else 20)
                _lib.check(lib.symgpu_dev_free(buf))
                put(f'commutes/{n}/{N}', t, N * N, N * N, word_steps=wq)
            # -- one non-Clifford rotation of a duplicate-free operator
            if not only or 'rotation' in only:
                rng = np.random.default_rng(7 + n)
                q = packing.pack_rows(rng.random((1, 2 * n)) < 0.3)[0]
                P = kernels.cleanup_dev(A, 1e-15)
                T = P.n_terms
                r0, allc = kernels.rotate_single_dev(P, q, 0.3)
                n_out = T if allc else r0.n_terms
                if r0 is not None:
                    r0.free()

                def rot():
                    r, a = kernels.rotate_single_dev(P, q, 0.3)
                    if r is not None:
                        r.free()
                t = timed(rot, 30)
                put(f'rotation/{n}/{N}', t, T, (T + n_out) * row, terms_in=T, terms_out=n_out)
                P.free()
            A.free()
    name = ctypes.create_string_buffer(256)
    try:
        _lib.check(lib.symgpu_device_name(name, 256))
    except Exception:                                              # noqa: BLE001
        pass
    print(json.dumps({'what': 'tools/sweep_shapes.py: seconds per call (C ABI, operands resident) and fraction of 8 TB/s on the algorithmic bytes', 'device': name.value.decode(),
                      'qubits': QUBITS, 'terms': TERMS, 'degraded_kernels': _lib.degraded(), 'results': res}))


if __name__ == '__main__':
    main()
