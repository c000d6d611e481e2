"""NumPy-facing wrappers of the C-ABI: packed ``uint64`` rows + ``complex128`` coefficients in, same out.

Every function here runs on the GPU through ``libsymgpu.so``; there is no CPU path (a missing library or
GPU raises :class:`symmer_amd._lib.SymgpuError`).  Reference seams replaced (file:line in /root/reference):
``symplectic_cleanup`` utils.py:230, ``matmul_GF2``/``commutes_termwise`` utils.py:9 / base.py:938,
``_multiply_by_operator`` base.py:764, ``_rotate_by_single_Pword`` base.py:1090, ``_rref_binary`` utils.py:292,
``IndependentOp.symmetry_generators`` independent_op.py:90.
"""
import ctypes
import numpy as np
from . import _lib
from ._lib import addr, check

c_i64, c_int = ctypes.c_int64, ctypes.c_int


def _rows(a):
    a = np.ascontiguousarray(a, dtype='<u8')
    assert a.ndim == 2 and a.shape[1] % 2 == 0 and a.shape[1] >= 2, 'packed rows must be uint64[T, 2*Wq]'
    return a


def _coeff(c):
    return np.ascontiguousarray(c, dtype=np.complex128)


class DeviceOp:
    """Owner of a device-resident operator handle (``symgpu_op_t``).  ``shared`` is set once a handle is attached to more than one
    Python object (``copy()``): the in-place primitives (``set_coeff``, ``scale``) then work on a clone."""

    def __init__(self, handle):
        self.handle = handle
        self.shared = False

    def __deepcopy__(self, memo):
        # a handle is never duplicated by value (two owners of one symgpu_op_t would free it twice): copies share the object
        self.shared = True
        return self

    def __reduce__(self):
        raise TypeError('a DeviceOp is a pointer into this process\'s GPU memory and cannot be pickled: pickle the PauliwordOp that owns it '
                        '(it is brought to the host), or download() the rows')

    @classmethod
    def upload(cls, rows, coeff=None):
        rows = _rows(rows)
        coeff = None if coeff is None else _coeff(coeff)
        if coeff is not None:
            assert coeff.shape[0] == rows.shape[0]
        h = ctypes.c_void_p()
        check(_lib.lib().symgpu_op_upload(addr(rows), addr(coeff), rows.shape[0], rows.shape[1] // 2, ctypes.byref(h)))
        return cls(h)

    @classmethod
    def upload_bool(cls, symp_matrix, coeff=None):
        """The reference layout itself (bool[T, 2n]) -> packed rows, packed on the device (``symgpu_op_upload_bool``)."""
        symp = np.ascontiguousarray(symp_matrix, dtype=np.bool_)
        assert symp.ndim == 2 and symp.shape[1] % 2 == 0 and symp.shape[1] >= 2
        coeff = None if coeff is None else _coeff(coeff)
        h = ctypes.c_void_p()
        check(_lib.lib().symgpu_op_upload_bool(addr(symp), addr(coeff), symp.shape[0], symp.shape[1] // 2, ctypes.byref(h)))
        return cls(h)

    @classmethod
    def alloc(cls, capacity, wq, with_coeff=True):
        h = ctypes.c_void_p()
        check(_lib.lib().symgpu_op_alloc(int(capacity), int(wq), 1 if with_coeff else 0, ctypes.byref(h)))
        return cls(h)

    @classmethod
    def random(cls, n_terms, n_qubits, density=0.3, seed=1234):
        h = ctypes.c_void_p()
        check(_lib.lib().symgpu_op_random(int(n_terms), int(n_qubits), float(density), int(seed), ctypes.byref(h)))
        return cls(h)

    def info(self):
        t, wq, cap = c_i64(0), c_int(0), c_i64(0)
        check(_lib.lib().symgpu_op_info(self.handle, ctypes.addressof(t), ctypes.addressof(wq), ctypes.addressof(cap)))
        return t.value, wq.value, cap.value

    @property
    def n_terms(self):
        return self.info()[0]

    def set_rows(self, t):
        check(_lib.lib().symgpu_op_set_rows(self.handle, int(t)))

    def download(self, with_coeff=True):
        t, wq, _ = self.info()
        rows = np.empty((t, 2 * wq), dtype='<u8')
        coeff = np.empty(t, dtype=np.complex128) if with_coeff else None
        check(_lib.lib().symgpu_op_download(self.handle, addr(rows), addr(coeff), t))
        return (rows, coeff) if with_coeff else rows

    def download_coeff(self):
        t = self.info()[0]
        coeff = np.empty(t, dtype=np.complex128)
        check(_lib.lib().symgpu_op_download(self.handle, None, addr(coeff), t))
        return coeff

    def download_bool(self, n_qubits):
        """bool[T, 2n] in the reference layout, unpacked on the device (``symgpu_op_download_bool``)."""
        t = self.info()[0]
        out = np.empty((t, 2 * int(n_qubits)), dtype=np.bool_)
        check(_lib.lib().symgpu_op_download_bool(self.handle, int(n_qubits), addr(out), t))
        return out

    def clone(self):
        h = ctypes.c_void_p()
        check(_lib.lib().symgpu_op_clone(self.handle, ctypes.byref(h)))
        return DeviceOp(h)

    def set_coeff(self, coeff):
        coeff = _coeff(coeff)
        assert coeff.shape[0] == self.info()[0]
        check(_lib.lib().symgpu_op_set_coeff(self.handle, addr(coeff)))

    def scale(self, const, conjugate_first=False):
        const = complex(const)
        check(_lib.lib().symgpu_op_scale(self.handle, const.real, const.imag, 1 if conjugate_first else 0))

    def ycount(self):
        out = np.zeros(self.info()[0], dtype=np.int64)
        check(_lib.lib().symgpu_op_ycount(self.handle, addr(out)))
        return out

    def checksum(self, with_coeff=True):
        _, wq, _ = self.info()
        x = np.zeros(2 * wq, dtype='<u8')
        c = np.zeros(1, dtype=np.complex128)
        check(_lib.lib().symgpu_op_checksum(self.handle, addr(x), addr(c) if with_coeff else None))
        return x, complex(c[0])

    def free(self):
        if self.handle is not None and self.handle.value:
            _lib.load().symgpu_op_free(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def concat_dev(parts, with_coeff=True):
    """Device operators stacked in order into a new one (device-to-device copies)."""
    sizes = [p.info() for p in parts]
    wq = sizes[0][1]
    out = DeviceOp.alloc(sum(t for t, _, _ in sizes), wq, with_coeff)
    at = 0
    for p, (t, w, _) in zip(parts, sizes):
        assert w == wq, 'operands packed with different Wq'
        if t:
            check(_lib.lib().symgpu_op_copy_rows(out.handle, at, p.handle, 0, t))
        at += t
    return out


def slice_dev(op, begin, end, with_coeff=True):
    """Rows [begin, end) of a device operator as a new one."""
    out = DeviceOp.alloc(end - begin, op.info()[1], with_coeff)
    if end > begin:
        check(_lib.lib().symgpu_op_copy_rows(out.handle, 0, op.handle, int(begin), int(end - begin)))
    return out


def transfer_counters():
    """(bytes host -> device, bytes device -> host, operator uploads, operator downloads) since the library was loaded."""
    v = c_i64(0)
    out = []
    for which in (7, 8, 9, 10):
        check(_lib.lib().symgpu_debug_counter(which, ctypes.addressof(v)))
        out.append(v.value)
    return tuple(out)


def sync():
    check(_lib.lib().symgpu_sync())


def ycount(rows):
    rows = _rows(rows)
    out = np.zeros(rows.shape[0], dtype=np.int64)
    check(_lib.lib().symgpu_ycount(addr(rows), rows.shape[0], rows.shape[1] // 2, addr(out)))
    return out


_device_memory = {}


def device_memory_bytes():
    """Total memory of the calling thread's current device (cached per device)."""
    d = _lib.current_device() if _lib._initialised_device is not None else -1
    if d not in _device_memory:
        free_b, total_b = ctypes.c_int64(0), ctypes.c_int64(0)
        check(_lib.lib().symgpu_mem_info(ctypes.addressof(free_b), ctypes.addressof(total_b)))
        _device_memory[_lib.current_device()] = int(total_b.value)
        d = _lib.current_device()
    return _device_memory[d]


def commutes(a_rows, b_rows):
    """bool[N, M]: True where a_rows[i] commutes with b_rows[j]."""
    a_rows, b_rows = _rows(a_rows), _rows(b_rows)
    assert a_rows.shape[1] == b_rows.shape[1], 'operands packed with different Wq'
    n, m = a_rows.shape[0], b_rows.shape[0]
    out = np.empty((n, m), dtype=np.uint8)
    if n and m:
        same = a_rows is b_rows
        check(_lib.lib().symgpu_commutes(addr(a_rows), n, addr(a_rows if same else b_rows), m, a_rows.shape[1] // 2, addr(out)))
    return out.view(np.bool_)


def commutes_handles(a, b):
    """bool[N, M] for two device operators (rows only are used): the table is computed into device memory and read back once."""
    (n, wq, _), (m, wq_b, _) = a.info(), b.info()
    assert wq == wq_b, 'operands packed with different Wq'
    out = np.empty((n, m), dtype=np.uint8)
    if n and m:
        buf = ctypes.c_void_p()
        check(_lib.lib().symgpu_dev_alloc(n * m, ctypes.byref(buf)))
        try:
            check(_lib.lib().symgpu_commutes_dev(a.handle, 0, n, b.handle, buf))
            check(_lib.lib().symgpu_dev_download(buf, addr(out), n * m))
        finally:
            _lib.lib().symgpu_dev_free(buf)
    return out.view(np.bool_)


def mul_allpairs(inner, ci, outer, co, inner_is_left=True):
    """Uncleaned product: row ``o*Ni + i`` = ``inner[i] ^ outer[o]`` with its phase-corrected coefficient."""
    inner, outer, ci, co = _rows(inner), _rows(outer), _coeff(ci), _coeff(co)
    assert inner.shape[1] == outer.shape[1]
    ni, no = inner.shape[0], outer.shape[0]
    rows = np.empty((ni * no, inner.shape[1]), dtype='<u8')
    coeff = np.empty(ni * no, dtype=np.complex128)
    if ni and no:
        check(_lib.lib().symgpu_mul_allpairs(addr(inner), addr(ci), ni, addr(outer), addr(co), no, inner.shape[1] // 2,
                                             1 if inner_is_left else 0, addr(rows), addr(coeff)))
    return rows, coeff


def _thr_args(zero_threshold):
    return (0.0, 0) if zero_threshold is None else (float(zero_threshold), 1)


def cleanup(rows, coeff, zero_threshold=1e-15):
    """First-occurrence dedup + sequential coefficient sums + strict threshold (None keeps everything)."""
    rows, coeff = _rows(rows), _coeff(coeff)
    assert rows.shape[0] == coeff.shape[0]
    if rows.shape[0] == 0:
        return rows.copy(), coeff.copy()
    thr, use = _thr_args(zero_threshold)
    op = DeviceOp.upload(rows, coeff)
    out = ctypes.c_void_p()
    try:
        check(_lib.lib().symgpu_cleanup_dev(op.handle, thr, use, ctypes.byref(out)))
        return DeviceOp(out).download()
    finally:
        op.free()


MAX_PAIRS_PER_CALL = 1 << 31      # one device call sorts 32-bit pair indices; larger products are tiled over the outer operand


def mul_cleanup_handles(a, b, inner_is_left=True, zero_threshold=1e-15, max_pairs=None):
    """Fused product + cleanup of two device operators (``a`` inner, ``b`` outer; ``a is b`` squares) -> new DeviceOp; the product rows are
    never materialised.  Products with more than ``max_pairs`` pairs are tiled over the outer (slow) index: every slab is cleaned on the
    device, the concatenation of the cleaned slabs is cleaned once more (same first-occurrence order; coefficient sums associate per slab,
    i.e. within 1e-16 relative).  Nothing returns to the host."""
    ni, no = a.info()[0], b.info()[0]
    thr, use = _thr_args(zero_threshold)
    max_pairs = MAX_PAIRS_PER_CALL if max_pairs is None else int(max_pairs)
    out = ctypes.c_void_p()
    if ni * no <= max_pairs:
        check(_lib.lib().symgpu_mul_cleanup_dev(a.handle, b.handle, 1 if inner_is_left else 0, thr, use, ctypes.byref(out)))
        return DeviceOp(out)
    slab = max(1, max_pairs // ni)
    parts = []
    for o0 in range(0, no, slab):
        piece = slice_dev(b, o0, min(no, o0 + slab))
        part = ctypes.c_void_p()
        # partial sums must not be thresholded: a term may only cancel across slabs
        check(_lib.lib().symgpu_mul_cleanup_dev(a.handle, piece.handle, 1 if inner_is_left else 0, 0.0, 0, ctypes.byref(part)))
        piece.free()
        parts.append(DeviceOp(part))
    stacked = concat_dev(parts)
    for p in parts:
        p.free()
    check(_lib.lib().symgpu_cleanup_dev(stacked.handle, thr, use, ctypes.byref(out)))
    stacked.free()
    return DeviceOp(out)


def mul_cleanup(inner, ci, outer, co, inner_is_left=True, zero_threshold=1e-15, max_pairs=None):
    """Host arrays in, host arrays out around :func:`mul_cleanup_handles` (``outer is inner``: one device operand serves both factors,
    which lets the library sort half of the pairs, cleanup.hip)."""
    same = outer is inner and co is ci
    inner, outer, ci, co = _rows(inner), _rows(outer), _coeff(ci), _coeff(co)
    assert inner.shape[1] == outer.shape[1]
    if inner.shape[0] == 0 or outer.shape[0] == 0:
        return np.empty((0, inner.shape[1]), dtype='<u8'), np.empty(0, dtype=np.complex128)
    a = DeviceOp.upload(inner, ci)
    b = a if same else DeviceOp.upload(outer, co)
    try:
        res = mul_cleanup_handles(a, b, inner_is_left, zero_threshold, max_pairs)
        try:
            return res.download()
        finally:
            res.free()
    finally:
        a.free()
        if b is not a:
            b.free()


import functools
import math


@functools.lru_cache(maxsize=256)
def rotation_args(angle, threshold=1e-18):
    """(cos, sin, clifford_k) for ``_rotate_by_single_Pword`` (base.py:1146-1156): clifford_k = round(2*angle/pi) if the
    angle is a multiple of pi/2 to within ``threshold``, else -1.  Negative multiples keep the reference's behaviour
    (``int_part in [2,3]`` is not reduced mod 4): only the parity and membership in {2,3} matter, so a negative k is
    mapped to k & 1 (parity preserved, never 2 or 3)."""
    multiple = angle * 2 / np.pi
    k = round(multiple)
    if abs(k - multiple) <= threshold:
        ck = k if k in (2, 3) else (k % 2)
        return 0.0, 0.0, int(ck)             # the Clifford branch never uses cos / sin (base.py:1139-1154): not computed
    # (np.cos / np.sin of a Python float and math.cos / math.sin are the same libm call; the results are cached per angle: a Trotter circuit
    # repeats a handful of angles thousands of times)
    return float(np.cos(angle)), float(np.sin(angle)), -1


def rotate_single_resident(op, q_addr, angle, zero_threshold=1e-15, clifford_threshold=1e-18):
    """``rotate_single_dev`` for the drop-in class, without its per-call conveniences (the API call cost 1.6x the C-ABI
    step): the generator's packed row comes as an address the caller keeps, cos / sin come from the cache, and the result's term count is
    read once and returned.  -> (DeviceOp or None, all_commute, n_terms)."""
    cos_t, sin_t, k = rotation_args(angle, clifford_threshold)
    out = ctypes.c_void_p()
    allc, t = c_int(0), c_i64(0)
    check(_lib.lib().symgpu_rotate_single_dev_n(op.handle, q_addr, cos_t, sin_t, k, zero_threshold, ctypes.byref(out), ctypes.addressof(allc),
                                                ctypes.addressof(t)))
    if allc.value:
        return None, True, 0
    return DeviceOp(out), False, t.value


def rotate_single_dev(op, q_row, angle, zero_threshold=1e-15, clifford_threshold=1e-18):
    """Device-resident rotation: returns (DeviceOp or None, all_commute).  ``clifford_threshold`` is the reference's
    ``threshold`` argument (base.py:1146): how close 2*angle/pi must be to an integer for the Clifford branch."""
    q_row = np.ascontiguousarray(q_row, dtype='<u8').reshape(-1)
    cos_t, sin_t, k = rotation_args(angle, clifford_threshold)
    out = ctypes.c_void_p()
    allc = c_int(0)
    check(_lib.lib().symgpu_rotate_single_dev(op.handle, addr(q_row), cos_t, sin_t, k, float(zero_threshold), ctypes.byref(out),
                                              ctypes.addressof(allc)))
    if allc.value:
        return None, True
    return DeviceOp(out), False


# symgpu_rotate_clifford_chain_dev: one single-workgroup launch for the whole run up to 128 terms, two launches per rotation up to 262,144 (3.5 us per rotation at 1 term,
# 5 us at 64, 29 us at 1,000; n = 1000), the per-rotation kernels back to back without a host read-back above that.
CLIFFORD_CHAIN_MAX_TERMS = 1 << 22
CLIFFORD_CHAIN_KERNEL_LIMIT = 8192


def rotate_clifford_chain_dev(op, q_rows, ks):
    """K Clifford rotations of a CLEAN device operator (straight from a cleanup: no duplicate rows, |c| > 1e-15) with at most
    CLIFFORD_CHAIN_MAX_TERMS terms without any host round trip between them.  ``q_rows`` uint64[K, 2*Wq], ``ks`` the clifford_k of every rotation as
    returned by :func:`rotation_args` (0..3).  Returns a new DeviceOp."""
    q_rows = np.ascontiguousarray(q_rows, dtype='<u8')
    ks = np.ascontiguousarray(ks, dtype=np.int32)
    assert q_rows.ndim == 2 and q_rows.shape[0] == ks.shape[0]
    out = ctypes.c_void_p()
    check(_lib.lib().symgpu_rotate_clifford_chain_dev(op.handle, addr(q_rows), addr(ks), q_rows.shape[0], ctypes.byref(out)))
    return DeviceOp(out)


def perform_rotations_dev(op, q_rows, cos_t, sin_t, ks, clean, zero_threshold=1e-15):
    """``perform_rotations`` on a device operator in one library call (``symgpu_perform_rotations_dev``): returns
    (new DeviceOp or None if nothing changed, rotations done, uint8 flags of the single rotations that acted, clean flag).
    The call processes all K rotations, also through the states "no terms" and "0 * I" (it follows the reference's alternation itself)."""
    q_rows = np.ascontiguousarray(q_rows, dtype='<u8')
    cos_t, sin_t = np.ascontiguousarray(cos_t, dtype=np.float64), np.ascontiguousarray(sin_t, dtype=np.float64)
    ks = np.ascontiguousarray(ks, dtype=np.int32)
    K = q_rows.shape[0]
    acted = np.zeros(K, dtype=np.uint8)
    out = ctypes.c_void_p()
    n_done, clean_out = c_i64(0), c_int(0)
    check(_lib.lib().symgpu_perform_rotations_dev(op.handle, addr(q_rows), addr(cos_t), addr(sin_t), addr(ks), K, float(zero_threshold),
                                                  1 if clean else 0, ctypes.byref(out), addr(acted), ctypes.addressof(n_done), ctypes.addressof(clean_out)))
    return (DeviceOp(out) if out.value else None), n_done.value, acted, bool(clean_out.value)


def cleanup_dev(op, zero_threshold=1e-15):
    thr, use = _thr_args(zero_threshold)
    out = ctypes.c_void_p()
    check(_lib.lib().symgpu_cleanup_dev(op.handle, thr, use, ctypes.byref(out)))
    return DeviceOp(out)


def rref(matrix_words, want_pivots=False):
    """``_rref_binary`` on uint64[R, Wc] packed rows -> (reduced, xor_count[, pivot column per row or -1])."""
    m = np.array(matrix_words, dtype='<u8', order='C', copy=True)
    assert m.ndim == 2
    R, wc = m.shape
    count = c_i64(0)
    piv = np.full(R, -1, dtype=np.int64)
    if R and wc:
        check(_lib.lib().symgpu_rref(addr(m), R, wc, ctypes.addressof(count), addr(piv)))
    return (m, count.value, piv) if want_pivots else (m, count.value)


def symmetry_kernel(h_rows, n_qubits):
    """Packed generators of the Z2 symmetries of the operator with packed rows ``h_rows`` -> (rows, xor_count)."""
    h_rows = _rows(h_rows)
    wq = h_rows.shape[1] // 2
    out = np.zeros((2 * n_qubits, 2 * wq), dtype='<u8')
    k, count = c_i64(0), c_i64(0)
    check(_lib.lib().symgpu_symmetry_kernel(addr(h_rows), h_rows.shape[0], int(n_qubits), wq, addr(out), 2 * n_qubits,
                                            ctypes.addressof(k), ctypes.addressof(count)))
    return out[:k.value].copy(), count.value


def symmetry_kernel_handle(op, n_qubits):
    """The same on a device-resident operator (``symgpu_symmetry_kernel_dev``): only the generators come back."""
    wq = op.info()[1]
    out = np.zeros((2 * n_qubits, 2 * wq), dtype='<u8')
    k, count = c_i64(0), c_i64(0)
    check(_lib.lib().symgpu_symmetry_kernel_dev(op.handle, int(n_qubits), addr(out), 2 * n_qubits, ctypes.addressof(k), ctypes.addressof(count)))
    return out[:k.value].copy(), count.value


# ---- f2: generator routines on resident operators (csrc/genrec.hip) -----------------------------------------------------
def gf2_rank_dev(op):
    """Rank over GF(2) of the operator's symplectic rows (= non-zero rows of ``_rref_binary``, utils.py:292-315)."""
    r = c_i64(0)
    check(_lib.lib().symgpu_op_gf2_rank(op.handle, ctypes.addressof(r)))
    return r.value


def generators_dev(op):
    """``PauliwordOp.generators`` (base.py:1436-1456) -> new DeviceOp: non-zero rows of ``_rref_binary(symp_matrix)``, coefficients 1."""
    out = ctypes.c_void_p()
    check(_lib.lib().symgpu_generators_dev(op.handle, ctypes.byref(out)))
    return DeviceOp(out)


def generator_reconstruction_dev(gens, op, n_qubits):
    """``PauliwordOp.generator_reconstruction`` (base.py:523-560) of ``op`` in ``gens`` -> (int64[T, g], bool[T])."""
    g, t = gens.info()[0], op.info()[0]
    recon = np.empty((t, g), dtype=np.int64)
    mask = np.empty(t, dtype=np.uint8)
    check(_lib.lib().symgpu_generator_reconstruction_dev(gens.handle, op.handle, int(n_qubits), addr(recon), addr(mask)))
    return recon, mask.view(np.bool_)


# ---- f3 / f4: projection, noncontextuality test, state inner product -------------------------------------------------
def sector_signs(eigenvalues):
    """Stabiliser eigenvalues as ints in {-1, 0, +1}.  The reference multiplies by the eigenvalue itself (projection/base.py:68-71); the
    sign-mask form of the device kernel is that product only for these three values, so anything else is refused — with an exception, not
    an assert (`python -O`), and before anything has consumed the values; entries within 1e-12 of an allowed value (a sector that went
    through floating point) are rounded as the reference's own `IndependentOp` would store them."""
    ev = np.asarray(eigenvalues).ravel()
    if ev.size == 0:
        return np.zeros(0, dtype=np.int64)
    if np.iscomplexobj(ev):
        if np.any(np.abs(ev.imag) > 1e-12):
            raise ValueError(f'stabiliser eigenvalues must be -1, 0 or +1, got {ev}')
        ev = ev.real
    rounded = np.rint(np.asarray(ev, dtype=float))
    if np.any(np.abs(ev - rounded) > 1e-12) or not np.all(np.isin(rounded, (-1.0, 0.0, 1.0))):
        raise ValueError(f'stabiliser eigenvalues must be -1, 0 or +1, got {ev}')
    return rounded.astype(np.int64)


def project_dev(op, stab_rows, eigenvalues, keep_qubits, n_qubits, zero_threshold=1e-15):
    """``S3Projection._perform_projection`` (projection/base.py:44-84) on a device operator: ``stab_rows`` uint64[k, 2*Wq] the fixed
    single-qubit stabilisers, ``eigenvalues`` int[k] their sector, ``keep_qubits`` the ascending indices of the qubits that stay.
    Returns (cleaned projected DeviceOp, number of terms that commuted with every stabiliser)."""
    stab_rows = np.ascontiguousarray(stab_rows, dtype='<u8').reshape(-1, stab_rows.shape[-1]) if len(stab_rows) else np.zeros((0, 2), dtype='<u8')
    k = stab_rows.shape[0]
    wq = op.info()[1]
    neg = np.zeros(2 * wq, dtype='<u8')
    eigenvalues = sector_signs(eigenvalues)
    for row, ev in zip(stab_rows, eigenvalues):
        if ev == -1:
            neg |= row                                             # (projection/base.py:69: the column index list has one entry per stabiliser)
    keep = np.ascontiguousarray(keep_qubits, dtype=np.int32)
    thr, use = _thr_args(zero_threshold)
    out = ctypes.c_void_p()
    n_surv = c_i64(0)
    check(_lib.lib().symgpu_project_dev(op.handle, addr(stab_rows) if k else None, k, addr(neg), addr(keep), keep.shape[0], int(n_qubits), thr, use,
                                        ctypes.byref(out), ctypes.addressof(n_surv)))
    return DeviceOp(out), n_surv.value


def noncontextual_dev(op):
    res = c_int(0)
    check(_lib.lib().symgpu_noncontextual_dev(op.handle, ctypes.addressof(res)))
    return bool(res.value)


def state_inner_dev(a, b):
    """Inner product of two CLEANED device states (operators whose X blocks are the basis strings): sum of c_a * c_b over the rows present
    in both, added in the order of ``a``."""
    out = np.zeros(2, dtype=np.float64)
    check(_lib.lib().symgpu_state_inner_dev(a.handle, b.handle, addr(out)))
    return complex(out[0], out[1])


# ---- cleanups that also return the first-occurrence index of every output term (hash-partitioned multi-GPU cleanup, parallel.py) ----
def _first_index(op):
    t = op.n_terms
    first = np.zeros(t, dtype='<u8')
    check(_lib.lib().symgpu_op_first_index(op.handle, addr(first), t))
    return first


def mul_cleanup_indexed(inner, ci, outer, co, inner_is_left=True, zero_threshold=None):
    """Fused product + cleanup -> (rows, coeff, i_first, o_first): ``inner[i_first[t]] ^ outer[o_first[t]]`` is output row t, and
    ``o_first * Ni + i_first`` is the pair index of its first occurrence (base.py:783-792, utils.py:271)."""
    inner, outer, ci, co = _rows(inner), _rows(outer), _coeff(ci), _coeff(co)
    if inner.shape[0] == 0 or outer.shape[0] == 0:
        z = np.zeros(0, dtype=np.int64)
        return np.empty((0, inner.shape[1]), dtype='<u8'), np.empty(0, dtype=np.complex128), z, z
    thr, use = _thr_args(zero_threshold)
    a = DeviceOp.upload(inner, ci)
    b = DeviceOp.upload(outer, co)
    out = ctypes.c_void_p()
    try:
        check(_lib.lib().symgpu_mul_cleanup_indexed_dev(a.handle, b.handle, 1 if inner_is_left else 0, thr, use, ctypes.byref(out)))
        res = DeviceOp(out)
        rows, coeff = res.download()
        first = _first_index(res)
        res.free()
    finally:
        a.free(); b.free()
    return rows, coeff, (first & 0xFFFFFFFF).astype(np.int64), (first >> 32).astype(np.int64)


def cleanup_indexed(rows, coeff, zero_threshold=1e-15):
    """First-occurrence cleanup -> (rows, coeff, first): ``first[t]`` is the position of output term t's first input row."""
    rows, coeff = _rows(rows), _coeff(coeff)
    if rows.shape[0] == 0:
        return rows.copy(), coeff.copy(), np.zeros(0, dtype=np.int64)
    thr, use = _thr_args(zero_threshold)
    op = DeviceOp.upload(rows, coeff)
    out = ctypes.c_void_p()
    try:
        check(_lib.lib().symgpu_cleanup_indexed_dev(op.handle, thr, use, ctypes.byref(out)))
        res = DeviceOp(out)
        r, c = res.download()
        first = _first_index(res).astype(np.int64)
        res.free()
    finally:
        op.free()
    return r, c, first


# ---- device-resident pieces of the hash-partitioned multi-GPU cleanup (csrc/partition.hip) -----------------------------------------
def op_gather(op, idx):
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    out = ctypes.c_void_p()
    check(_lib.lib().symgpu_op_gather(op.handle, addr(idx), idx.shape[0], ctypes.byref(out)))
    return DeviceOp(out)


def mul_cleanup_indexed_dev(inner, outer, inner_is_left=True, zero_threshold=None):
    thr, use = _thr_args(zero_threshold)
    out = ctypes.c_void_p()
    check(_lib.lib().symgpu_mul_cleanup_indexed_dev(inner.handle, outer.handle, 1 if inner_is_left else 0, thr, use, ctypes.byref(out)))
    return DeviceOp(out)


def part_global_index(part, inner_idx, outer_idx, ni_global):
    inner_idx = np.ascontiguousarray(inner_idx, dtype=np.int64); outer_idx = np.ascontiguousarray(outer_idx, dtype=np.int64)
    check(_lib.lib().symgpu_part_global_index(part.handle, addr(inner_idx), inner_idx.shape[0], addr(outer_idx), outer_idx.shape[0], int(ni_global)))


def merge_indexed_dev(parts, key_bits=0, do_cleanup=True, zero_threshold=1e-15):
    """Indexed device operators -> one, ordered by index; ``do_cleanup``: shared rows merged at their smallest index + threshold."""
    arr = (ctypes.c_void_p * len(parts))(*[p.handle for p in parts])
    thr, use = _thr_args(zero_threshold)
    out = ctypes.c_void_p()
    check(_lib.lib().symgpu_merge_indexed_dev(ctypes.addressof(arr), len(parts), int(key_bits), 1 if do_cleanup else 0, thr, use, ctypes.byref(out)))
    return DeviceOp(out)


def op_first_index(op):
    return _first_index(op)


def op_set_first_index(op, first):
    first = np.ascontiguousarray(first, dtype='<u8')
    assert first.shape[0] == op.n_terms
    check(_lib.lib().symgpu_op_set_first_index(op.handle, addr(first)))
