"""ctypes binding of the C oracle (``oracle/oracle_c.c``).  TEST INFRASTRUCTURE ONLY — see that file's
header.  Operates on packed ``uint64[T, 2*Wq]`` rows and ``complex128`` coefficients."""
import ctypes, os, subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'liboracle.so')
_lib = None

_u64p = ctypes.POINTER(ctypes.c_uint64)
_f64p = ctypes.POINTER(ctypes.c_double)
_i64p = ctypes.POINTER(ctypes.c_int64)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE, 'liboracle.so'])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, 'oracle_c.c')):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_cleanup.restype = ctypes.c_int64
        _lib.orc_rref.restype = ctypes.c_int64
        _lib.orc_symmetry_generators.restype = ctypes.c_int64
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


def _rows(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.complex128)


def commutes(A, B):
    A, B = _rows(A), _rows(B)
    N, M, wq = A.shape[0], B.shape[0], A.shape[1] // 2
    out = np.zeros((N, M), dtype=np.uint8)
    lib().orc_commutes(_p(A, _u64p), ctypes.c_int64(N), _p(B, _u64p), ctypes.c_int64(M), ctypes.c_int(wq), _p(out, _u8p))
    return out.astype(bool)


def ycount(A):
    A = _rows(A)
    out = np.zeros(A.shape[0], dtype=np.int64)
    lib().orc_ycount(_p(A, _u64p), ctypes.c_int64(A.shape[0]), ctypes.c_int(A.shape[1] // 2), _p(out, _i64p))
    return out


def mul_allpairs(inner, ci, outer, co, inner_is_left=True):
    inner, outer, ci, co = _rows(inner), _rows(outer), _c(ci), _c(co)
    Ni, No, W = inner.shape[0], outer.shape[0], inner.shape[1]
    rows = np.zeros((Ni * No, W), dtype=np.uint64)
    coeff = np.zeros(Ni * No, dtype=np.complex128)
    lib().orc_mul_allpairs(_p(inner, _u64p), _p(ci, _f64p), ctypes.c_int64(Ni), _p(outer, _u64p), _p(co, _f64p),
                           ctypes.c_int64(No), ctypes.c_int(W // 2), ctypes.c_int(1 if inner_is_left else 0),
                           _p(rows, _u64p), _p(coeff, _f64p))
    return rows, coeff


def mul_allpairs_coeff(inner, ci, outer, co, inner_is_left=True):
    """The coefficients of :func:`mul_allpairs` without materialising the product rows (full-size slabs: 2.56e7 pairs)."""
    inner, outer, ci, co = _rows(inner), _rows(outer), _c(ci), _c(co)
    Ni, No, W = inner.shape[0], outer.shape[0], inner.shape[1]
    coeff = np.zeros(Ni * No, dtype=np.complex128)
    lib().orc_mul_allpairs(_p(inner, _u64p), _p(ci, _f64p), ctypes.c_int64(Ni), _p(outer, _u64p), _p(co, _f64p),
                           ctypes.c_int64(No), ctypes.c_int(W // 2), ctypes.c_int(1 if inner_is_left else 0),
                           None, _p(coeff, _f64p))
    return coeff


def cleanup(rows, coeff, thr=1e-15):
    """thr=None keeps every merged row (free-function default of utils.py:233)."""
    rows, coeff = _rows(rows), _c(coeff)
    T, W = rows.shape
    out_rows = np.zeros((T, W), dtype=np.uint64)
    out_coeff = np.zeros(T, dtype=np.complex128)
    n = lib().orc_cleanup(_p(rows, _u64p), _p(coeff, _f64p), ctypes.c_int64(T), ctypes.c_int(W),
                          ctypes.c_double(0.0 if thr is None else thr), ctypes.c_int(0 if thr is None else 1),
                          _p(out_rows, _u64p), _p(out_coeff, _f64p))
    return out_rows[:n].copy(), out_coeff[:n].copy()


def mul(A, a, B, b, thr=1e-15):
    """``A * B`` with the fewer-term operand as the outer index (base.py:847-852)."""
    A, B = _rows(A), _rows(B)
    if A.shape[0] < B.shape[0]:
        rows, coeff = mul_allpairs(B, b, A, a, inner_is_left=False)
    else:
        rows, coeff = mul_allpairs(A, a, B, b, inner_is_left=True)
    return cleanup(rows, coeff, thr)


def rref(rows, want_pivots=False):
    rows = _rows(rows).copy()
    R, Wc = rows.shape
    piv = np.zeros(R, dtype=np.int64)
    n = lib().orc_rref(_p(rows, _u64p), ctypes.c_int64(R), ctypes.c_int64(Wc), _p(piv, _i64p))
    return (rows, int(n), piv) if want_pivots else (rows, int(n))


def symmetry_generators(H, n):
    H = _rows(H)
    M, wq = H.shape[0], H.shape[1] // 2
    out = np.zeros((2 * n, 2 * wq), dtype=np.uint64)
    nx = ctypes.c_int64(0)
    k = lib().orc_symmetry_generators(_p(H, _u64p), ctypes.c_int64(M), ctypes.c_int(n), ctypes.c_int(wq),
                                      _p(out, _u64p), ctypes.byref(nx))
    return out[:k].copy(), int(nx.value)
