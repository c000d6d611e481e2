"""One process, several MI355X: the all-pairs kernels sharded over the devices of a node without a launcher (SURVEY.md §8b threading row:
"multi-GPU = one host process, one thread (or ncclGroupStart/End) driving <= 8 devices — no MPI launcher needed"; §8e partitioning).

The reference has nothing here (``symmer/process_handler.py:100-115`` is a CPU fork pool that never touches this path).  What is
sharded is what SURVEY §8e / BASELINE's north star shard:

* ``commutes_termwise`` / ``adjacency_matrix`` (base.py:938-971): the LEFT term axis in contiguous blocks of ``ceil(N / G)`` rows; every
  device holds a ``1 / G`` shard of the right operand, one grouped RCCL all-gather over xGMI assembles all of it on every device
  (``symgpu_comm_allgather_ops``), every device computes its ``[N / G, M]`` block, and the blocks come back to the host in parallel (one
  PCIe link per device: the 40 GB result of cfg5 is the whole cost of the API call on one GPU).
* the product + cleanup behind ``__mul__`` (base.py:821-859): the OUTER index in contiguous blocks — pair ``(i, o)`` has index ``o * Ni + i``
  in the reference, so device-major order is the reference's order (``parallel.Communicator.mul_cleanup_sharded`` explains why that makes
  the result identical): a fused product + cleanup per device without threshold, the cleaned parts copied to device 0 (peer copies),
  one cleanup of their concatenation with the threshold.

:class:`DeviceGroup` holds the arithmetic (block bounds, shard padding, placement of gathered rows and of result blocks); what touches
a device goes through a small backend object, so that ``tests/test_multi_device.py`` drives the same arithmetic for 8 "devices" on the
CPU with a checker backend.  With one device the group degenerates to the single-device calls (same kernels, same bytes).

``SYMGPU_DEVICES`` = number of devices the drop-in classes may use (default: all visible; 1 switches the sharding off).
"""
import ctypes
import os
import threading
import numpy as np
from . import _lib, packing
from .parallel import shard_bounds


# ------------------------------------------------------------------------------------------------------------------------------------
class HipBackend:
    """The devices of this process (``symgpu_init_all``): one context / stream / allocator per device, RCCL communicators from
    ``ncclCommInitAll``.  If RCCL cannot be brought up the right operand is staged through the host instead (``degraded`` says so)."""

    def __init__(self, n_devices):
        from . import kernels
        self.k = kernels
        self.lib = _lib.load()
        _lib.check(self.lib.symgpu_init_all(int(n_devices)))
        n = ctypes.c_int(0)
        _lib.check(self.lib.symgpu_n_initialised(ctypes.addressof(n)))
        self.n = min(int(n_devices), n.value) if n_devices > 0 else n.value
        self.degraded = None
        self._rccl = False
        if self.n > 1 or os.environ.get('SYMGPU_FORCE_COMM', '0') == '1':
            rc = self.lib.symgpu_comm_init_all(self.n)
            if rc == _lib.OK:
                self._rccl = True
            else:
                self.degraded = f'RCCL unavailable in single-process mode ({_lib.last_error()}): right operand staged through host memory'

    def use(self, d):
        _lib.check(self.lib.symgpu_set_device(int(d)))

    def current(self):
        """The calling thread's current device (DeviceGroup restores it when a sharded call returns)."""
        return _lib.current_device()

    def sync(self, d):
        self.use(d)
        _lib.check(self.lib.symgpu_sync())

    # -- operands -----------------------------------------------------------------------------------------------------------------
    def shard_from(self, d, source, r0, r1, capacity, with_coeff):
        """Rows [r0, r1) of ``source`` (host ``(rows, coeff)`` or a DeviceOp on any device) as an operator of ``capacity`` rows on device d."""
        self.use(d)
        wq = (source.info()[1] if isinstance(source, self.k.DeviceOp) else source[0].shape[1] // 2)
        op = self.k.DeviceOp.alloc(max(1, capacity), wq, with_coeff)
        if r1 > r0:
            if isinstance(source, self.k.DeviceOp):
                _lib.check(self.lib.symgpu_op_copy_rows(op.handle, 0, source.handle, int(r0), int(r1 - r0)))     # peer copy when the devices differ
            else:
                rows, coeff = source
                rows = np.ascontiguousarray(rows[r0:r1], dtype='<u8')
                c = None if (coeff is None or not with_coeff) else np.ascontiguousarray(coeff[r0:r1], dtype=np.complex128)
                _lib.check(self.lib.symgpu_op_write(op.handle, 0, rows.ctypes.data, None if c is None else c.ctypes.data, int(r1 - r0)))
        op.set_rows(r1 - r0)
        return op

    def alloc(self, d, capacity, wq, with_coeff):
        self.use(d)
        return self.k.DeviceOp.alloc(max(1, capacity), wq, with_coeff)

    def allgather(self, shards, fulls, n_rows_total):
        """fulls[d] <- all shards, shard d at rows [d * Ts, ...); grouped RCCL all-gather, or host staging if RCCL is not available."""
        if self._rccl:
            hs = (ctypes.c_void_p * self.n)(*[s.handle for s in shards])
            hf = (ctypes.c_void_p * self.n)(*[f.handle for f in fulls])
            _lib.check(self.lib.symgpu_comm_allgather_ops(ctypes.addressof(hs), ctypes.addressof(hf), self.n))
        else:
            ts = shards[0].info()[2]
            for d, f in enumerate(fulls):
                for s_idx, s in enumerate(shards):
                    t = s.info()[0]
                    if t:
                        _lib.check(self.lib.symgpu_op_copy_rows(f.handle, s_idx * ts, s.handle, 0, t))
                f.set_rows(ts * self.n)
        for f in fulls:
            f.set_rows(n_rows_total)

    # -- kernels (asynchronous on the device's stream) -------------------------------------------------------------------------------------
    def commutes_block(self, d, a, a0, a1, b):
        self.use(d)
        m = b.info()[0]
        buf = ctypes.c_void_p()
        _lib.check(self.lib.symgpu_dev_alloc(max(1, (a1 - a0) * m), ctypes.byref(buf)))
        if a1 > a0 and m:
            _lib.check(self.lib.symgpu_commutes_dev(a.handle, int(a0), int(a1), b.handle, buf))
        return buf

    def fetch_blocks(self, bufs, out, bounds):
        """Device d's block -> out[b0:b1], all devices at once (one host thread per device: every device has its own PCIe link, and the
        copies of a single thread into pageable memory would run one after the other)."""
        m = out.shape[1]

        def one(d):
            b0, b1 = bounds[d]
            try:
                if b1 > b0 and m:
                    self.use(d)                                    # the current device is per thread
                    _lib.check(self.lib.symgpu_dev_download(bufs[d], out[b0:b1].ctypes.data, (b1 - b0) * m))
            finally:
                self.lib.symgpu_dev_free(bufs[d])
        if self.n == 1:
            one(0)
            return
        errs = []

        def guarded(d):
            try:
                one(d)
            except Exception as exc:                               # noqa: BLE001 - re-raised on the caller's thread
                errs.append(exc)
        threads = [threading.Thread(target=guarded, args=(d,)) for d in range(self.n)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errs:
            raise errs[0]

    def mul_cleanup(self, d, inner, outer, inner_is_left, zero_threshold):
        self.use(d)
        return self.k.mul_cleanup_handles(inner, outer, inner_is_left, zero_threshold)

    def concat_on(self, d, parts):
        """The operators (on any devices) stacked in order on device d."""
        self.use(d)
        sizes = [p.info() for p in parts]
        out = self.k.DeviceOp.alloc(max(1, sum(t for t, _, _ in sizes)), sizes[0][1], True)
        at = 0
        for p, (t, _, _) in zip(parts, sizes):
            if t:
                _lib.check(self.lib.symgpu_op_copy_rows(out.handle, at, p.handle, 0, t))
            at += t
        out.set_rows(at)
        return out

    def cleanup(self, d, op, zero_threshold):
        self.use(d)
        return self.k.cleanup_dev(op, zero_threshold)

    def free(self, op):
        op.free()

    def n_rows(self, op):
        return op.info()[0]

    def width(self, source):
        return source.info()[1] if isinstance(source, self.k.DeviceOp) else source[0].shape[1] // 2

    def n_source_rows(self, source):
        return source.info()[0] if isinstance(source, self.k.DeviceOp) else source[0].shape[0]


# ------------------------------------------------------------------------------------------------------------------------------------
class DeviceGroup:
    """The sharding arithmetic of the all-pairs kernels over ``backend.n`` devices.  Operands are ``(rows, coeff)`` host arrays (packed
    uint64 rows, complex128 or None) or operators resident on one of the devices."""

    def __init__(self, backend):
        self.b = backend
        self.n = backend.n

    @property
    def degraded(self):
        return self.b.degraded

    def _replicate(self, source, with_coeff):
        """The whole operand on every device: 1 / G shards + one all-gather.  Returns (fulls, Ts)."""
        b, G = self.b, self.n
        m = b.n_source_rows(source)
        ts, bounds = shard_bounds(m, G)
        ts = max(1, ts)
        shards = [b.shard_from(d, source, bounds[d][0], bounds[d][1], ts, with_coeff) for d in range(G)]
        fulls = [b.alloc(d, ts * G, b.width(source), with_coeff) for d in range(G)]
        b.allgather(shards, fulls, m)
        for s in shards:
            b.free(s)
        return fulls

    def _home(self):
        """The caller's current device: a sharded call visits every device and must leave the thread where it found it (an operator
        created afterwards would otherwise land on the last device visited and refuse to meet the caller's older operators)."""
        return self.b.current()

    def commutes(self, a, b_src=None):
        """bool[N, M]: ``a`` against ``b_src`` (None: against itself — the all-gather that assembles the right operand is then also the
        distribution of the left blocks).  The thread's current device is the same before and after."""
        home = self._home()
        try:
            return self._commutes(a, b_src)
        finally:
            self.b.use(home)

    def _commutes(self, a, b_src):
        be, G = self.b, self.n
        same = b_src is None
        fulls = self._replicate(a if same else b_src, with_coeff=False)
        n = be.n_source_rows(a)
        m = n if same else be.n_source_rows(b_src)
        _, bounds = shard_bounds(n, G)
        out = np.empty((n, m), dtype=np.uint8)
        lefts, bufs = [], []
        for d in range(G):                                          # every device's block is launched before any result is awaited
            b0, b1 = bounds[d]
            if same:
                bufs.append(be.commutes_block(d, fulls[d], b0, b1, fulls[d]))
            else:
                left = be.shard_from(d, a, b0, b1, max(1, b1 - b0), False)
                lefts.append(left)
                bufs.append(be.commutes_block(d, left, 0, b1 - b0, fulls[d]))
        be.fetch_blocks(bufs, out, bounds)
        for op in lefts + fulls:
            be.free(op)
        return out.view(np.bool_)

    def mul_cleanup(self, inner, outer, inner_is_left=True, zero_threshold=1e-15, same=False):
        """Fused product + cleanup with the OUTER index in contiguous blocks over the devices; the result is an operator on the CALLER's
        current device, which is also the thread's current device afterwards.  ``same``: both factors are one operand (``P * P``)."""
        home = self._home()
        try:
            return self._mul_cleanup(inner, outer, inner_is_left, zero_threshold, same, home)
        finally:
            self.b.use(home)

    def _mul_cleanup(self, inner, outer, inner_is_left, zero_threshold, same, home):
        be, G = self.b, self.n
        inner_fulls = self._replicate(inner, with_coeff=True)
        no = be.n_source_rows(inner if same else outer)
        _, bounds = shard_bounds(no, G)
        parts, blocks = [], []
        for d in range(G):
            b0, b1 = bounds[d]
            if b1 <= b0:
                continue
            if same and G == 1:
                block = inner_fulls[0]                              # one device: the squared-operator path of the library (half of the pairs)
            else:
                block = be.shard_from(d, inner_fulls[d] if same else outer, b0, b1, b1 - b0, True)
                blocks.append(block)
            # partial sums must not be thresholded: a term may only cancel across devices
            parts.append(be.mul_cleanup(d, inner_fulls[d], block, inner_is_left, zero_threshold if G == 1 else None))
        if G == 1:
            res = parts[0]
        else:
            for d in range(G):
                be.sync(d)                                          # the parts are complete before the home device copies them
            cat = be.concat_on(home, parts)
            res = be.cleanup(home, cat, zero_threshold)
            be.free(cat)
            for p in parts:
                be.free(p)
        for op in blocks + inner_fulls:
            be.free(op)
        return res


# ------------------------------------------------------------------------------------------------------------------------------------
_group = None
_group_tried = False
MIN_PAIRS_COMMUTES = 1 << 30          # below these sizes one device finishes before the operands have been distributed
MIN_PAIRS_PRODUCT = 1 << 28


def n_devices_wanted():
    want = os.environ.get('SYMGPU_DEVICES')
    have = _lib.device_count()
    return max(1, min(have, int(want))) if want else have


def under_launcher(environ=None):
    """One process per GPU (torchrun / bench.py --gpus N): every rank owns ONE device (``symmer_amd.parallel``); a rank must not bring up
    contexts and an RCCL clique over all eight behind its launcher's back."""
    env = os.environ if environ is None else environ
    return int(env.get('WORLD_SIZE', '1') or 1) > 1 or 'RANK' in env


def group():
    """The process-wide :class:`DeviceGroup` over the visible devices, or None when there is only one, ``SYMGPU_DEVICES=1``, or the
    process is one rank of a launcher.  Creating it leaves the thread's current device where it was."""
    global _group, _group_tried
    if not _group_tried:
        _group_tried = True
        if under_launcher():
            return None
        n = n_devices_wanted()
        if n > 1:
            _lib.init()
            home = _lib.current_device()
            try:
                _group = DeviceGroup(HipBackend(n))
            finally:
                _lib.set_device(home)
    return _group


def product_uses_devices(n_pairs, row_bytes, device_bytes, environ=None):
    """Does ``A * B`` go to the device group?  The sharded product + cleanup computes per-device GENERAL sub-products (the squared-operator
    half-pairs path is lost), copies every cleaned part to one device and cleans the concatenation there: measured on one MI355X
    (tools/bench_devices_product.py) it is slower than the single-device call wherever that call can run.  So it is taken
    only on request (``SYMGPU_DEVICES_PRODUCT=1``, above MIN_PAIRS_PRODUCT pairs) or when one device cannot hold the product:
    ``n_pairs`` result rows of ``row_bytes`` (packed row + coefficient) plus 16 bytes of sort keys and indices per pair against half of
    the device's memory.  ``commutes_termwise`` is different: its blocks are independent and come back over eight PCIe links."""
    env = os.environ if environ is None else environ
    if n_pairs < MIN_PAIRS_PRODUCT:
        return False
    if env.get('SYMGPU_DEVICES_PRODUCT', '0') == '1':
        return True
    return n_pairs * (row_bytes + 16) > device_bytes // 2


def reset():
    global _group, _group_tried
    _group, _group_tried = None, False
