"""Multi-GPU sharding of the all-pairs kernels — one process per GPU.

The LEFT term axis is split into contiguous blocks, one per rank; every rank owns a 1/G shard of the packed RIGHT
operand, and ONE all-gather assembles the full right operand on every rank (RCCL ``ncclAllGather`` over xGMI through
``symgpu_comm_allgather_op``); each rank then computes its ``[N/G, M]`` block locally.  There is no reduction
collective anywhere on the path.  The reference has nothing comparable (its only parallelism is a CPU fork pool,
``symmer/process_handler.py``, off this path).

Control plane (unique-id exchange, barriers, max-over-ranks timing): either a few lines of plain TCP
(``control='tcp'``, default of ``bench.py``: no PyTorch in the GPU processes at all — PyTorch wheels bundle their own HIP
runtime and RCCL, which must not be mixed with the system ones this library links to) or ``torch.distributed`` with
the gloo backend (``control='gloo'``).  With ``data_plane='gloo-host'`` the all-gather itself runs over gloo on host
arrays, which is what the CPU tests (world_size 2) exercise.

If RCCL cannot be brought up on EVERY rank (agreed over the control plane before anybody enters the collective
``ncclCommInitRank``), the communicator falls back to ``data_plane='host-staged'``: the same all-gather, but D2H ->
control plane -> H2D.  The step still exchanges the shards (3.2 MB per rank in the benchmark, a few ms next to a 0.45 s
step); callers report the degraded data plane (``Communicator.degraded``).  Nothing ever silently replicates data.
"""
import os
import ctypes
import numpy as np
from . import packing


def shard_bounds(n_rows, world):
    """Contiguous equal blocks of ``ceil(n_rows/world)`` rows; tail ranks may be short or empty.
    Returns (Ts, [(begin, end)] per rank)."""
    ts = (n_rows + world - 1) // world if world > 0 else n_rows
    return ts, [(min(n_rows, r * ts), min(n_rows, (r + 1) * ts)) for r in range(world)]


class _TcpControl:
    """Minimal rank-0-rooted control plane over TCP: broadcast of bytes, max-reduce of a float, barrier."""

    N_CANDIDATE_PORTS = 8

    def __init__(self, rank, world, addr, port, timeout=300.0):
        """``port`` is the first of N_CANDIDATE_PORTS candidates (stride 101): rank 0 listens on the first one it can bind, the
        other ranks probe the candidates until one answers the handshake (magic derived from port and world size), so a port
        that happens to be taken by an unrelated service on the node does not break the job."""
        import socket, struct, time
        self.rank, self.world, self._struct = rank, world, struct
        self.peers = []
        host = addr if addr not in ('localhost',) else '127.0.0.1'
        ports = [1024 + (port - 1024 + 101 * k) % 64000 for k in range(self.N_CANDIDATE_PORTS)]
        magic = struct.pack('<4sHH', b'SYMG', port & 0xFFFF, world & 0xFFFF)
        if rank == 0:
            srv = None
            for p in ports:
                try:
                    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    srv.bind((host, p))
                    break
                except OSError:
                    srv.close()
                    srv = None
            if srv is None:
                raise OSError(f'control plane: none of the ports {ports} could be bound on {host}')
            srv.listen(world + 8)
            srv.settimeout(timeout)
            conns = {}
            while len(conns) < world - 1:
                c, _ = srv.accept()
                try:
                    c.settimeout(10.0)
                    hello = self._recv(c, len(magic) + 4)
                    r = struct.unpack('<i', hello[len(magic):])[0]
                    if hello[:len(magic)] != magic or not (0 < r < world) or r in conns:
                        raise ConnectionError('not a peer of this job')
                    c.sendall(magic)
                    c.settimeout(timeout)
                    conns[r] = c
                except (OSError, ConnectionError):
                    c.close()
            self.peers = [conns[r] for r in sorted(conns)]
            srv.close()
        else:
            deadline = time.time() + timeout
            c = None
            while c is None:
                for p in ports:
                    try:
                        s = socket.create_connection((host, p), timeout=5.0)
                    except OSError:
                        continue
                    try:
                        s.settimeout(10.0)
                        s.sendall(magic + struct.pack('<i', rank))
                        if self._recv(s, len(magic)) == magic:
                            c = s
                            break
                        s.close()
                    except (OSError, ConnectionError):
                        s.close()
                if c is None:
                    if time.time() > deadline:
                        raise TimeoutError(f'control plane: rank 0 did not answer on any of {ports} at {host}')
                    time.sleep(0.2)
            c.settimeout(timeout)
            self.peers = [c]

    @staticmethod
    def _recv(c, n):
        buf = bytearray(n)
        view, got = memoryview(buf), 0
        while got < n:
            k = c.recv_into(view[got:], n - got)
            if k == 0:
                raise ConnectionError('control-plane peer closed the connection')
            got += k
        return bytes(buf)

    def bcast(self, payload, nbytes):
        if self.rank == 0:
            for c in self.peers:
                c.sendall(payload)
            return payload
        return self._recv(self.peers[0], nbytes)

    def max(self, x):
        s = self._struct
        if self.rank == 0:
            vals = [float(x)] + [s.unpack('<d', self._recv(c, 8))[0] for c in self.peers]
            m = max(vals)
            for c in self.peers:
                c.sendall(s.pack('<d', m))
            return m
        self.peers[0].sendall(s.pack('<d', float(x)))
        return s.unpack('<d', self._recv(self.peers[0], 8))[0]

    def allgather(self, payload):
        """Equal-sized byte strings from every rank, concatenated in rank order, on every rank (through rank 0)."""
        n = len(payload)
        if self.rank == 0:
            parts = [payload] + [self._recv(c, n) for c in self.peers]
            whole = b''.join(parts)
            for c in self.peers:
                c.sendall(whole)
            return whole
        self.peers[0].sendall(payload)
        return self._recv(self.peers[0], n * self.world)

    def close(self):
        for c in self.peers:
            try:
                c.close()
            except OSError:
                pass
        self.peers = []


class _stdout_to_stderr:
    """RCCL prints a version banner on file descriptor 1 when a communicator is created; callers such as bench.py promise ONE
    JSON line on stdout, so the C-level stdout is pointed at stderr while RCCL initialises.  The redirect is process wide, so it
    is only ever entered and left on the MAIN thread (never inside a watchdog thread that may not come back)."""

    def __enter__(self):
        import sys
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        try:
            ctypes.CDLL(None).fflush(None)                          # the banner sits in the C stdio buffer: flush it while redirected
        except Exception:
            pass
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


class CollectiveHang(RuntimeError):
    """A collective did not return on some rank within its time limit.  The stream it was enqueued on cannot be used again, so
    there is no data plane to fall back to in this process: every rank raises this after the ranks have agreed, and the caller
    (bench.py) reports it and ends the process with a non-zero status."""


def _run_with_watchdog(fn, timeout):
    """Run ``fn()`` on a daemon thread and wait at most ``timeout`` seconds for it, with file descriptor 1 pointed at stderr for
    the duration (on THIS thread, restored whatever happens to the worker).  Returns (status, error text, thread): status 0 =
    returned, 1 = raised, 2 = still running."""
    import threading
    result = {}

    def _body():
        try:
            fn()
            result['ok'] = True
        except Exception as exc:                               # noqa: BLE001 - reported through the caller's agreement
            result['error'] = str(exc) or type(exc).__name__

    th = threading.Thread(target=_body, daemon=True)
    with _stdout_to_stderr():
        th.start()
        th.join(timeout)
    if th.is_alive():
        return 2, 'did not return within the time limit', th
    if 'error' in result:
        return 1, result['error'], None
    return 0, None, None


class Communicator:
    def __init__(self, rank=0, world=1, data_plane='none'):
        self.rank, self.world, self.data_plane = rank, world, data_plane
        self._dist = None
        self._tcp = None
        self.rccl_error = None
        self.degraded = None      # text when the data plane is not the one asked for (host-staged instead of RCCL)
        self.gathers = False      # True when the right operand must be assembled with the all-gather
        self._hung_thread = None  # a watchdog thread that is still inside RCCL: the communicator must not be destroyed under it
        self._gather_checked = False   # the first all-gather runs under a watchdog and is agreed on by all ranks
        self.needs_hard_exit = False

    # ---- construction ------------------------------------------------------------------------------------
    @classmethod
    def from_env(cls, data_plane='rccl', control='tcp'):
        """RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as set by ``torch.distributed.run``."""
        world = int(os.environ.get('WORLD_SIZE', '1'))
        rank = int(os.environ.get('RANK', '0'))
        # SYMGPU_FORCE_COMM=1 exercises the whole multi-rank path (control plane, RCCL init, all-gather) with ONE rank
        forced = os.environ.get('SYMGPU_FORCE_COMM', '0') == '1'
        if world == 1 and not forced:
            return cls(0, 1, 'none')
        addr = os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        mport = int(os.environ.setdefault('MASTER_PORT', '29531'))
        self = cls(rank, world, data_plane)
        self.gathers = True
        if control == 'gloo':
            import torch.distributed as dist
            if not dist.is_initialized():
                dist.init_process_group(backend='gloo', rank=rank, world_size=world)
            self._dist = dist
        else:
            # MASTER_PORT itself belongs to the launcher's store; use a derived port for our own socket
            port = int(os.environ.get('SYMGPU_CONTROL_PORT', 1024 + (mport + 7919) % 60000))
            self._tcp = _TcpControl(rank, world, addr, port)
        if data_plane == 'rccl':
            self._init_rccl()
        return self

    def _bcast_bytes(self, payload, nbytes):
        if self._tcp is not None:
            return self._tcp.bcast(payload, nbytes)
        import torch
        t = torch.tensor(list(payload), dtype=torch.uint8) if self.rank == 0 else torch.zeros(nbytes, dtype=torch.uint8)
        self._dist.broadcast(t, src=0)
        return bytes(t.tolist())

    def _fallback(self, why):
        """RCCL is unavailable: gather through host memory over the control plane instead (still a real exchange)."""
        self.rccl_error = why
        self.data_plane = 'host-staged'
        self.degraded = f'host-staged all-gather over the control plane (RCCL unavailable: {why})'

    def _init_rccl(self):
        """RCCL communicator over the control plane.  Every step that can fail locally is followed by an agreement over the
        control plane, and the first agreement — "librccl loads on this rank" — happens BEFORE any rank enters the collective
        ``ncclCommInitRank``: a rank that failed locally would otherwise leave its peers blocked inside it.  Either ALL ranks
        end up with a communicator or all of them fall back to the host-staged data plane."""
        from . import _lib
        self.rccl_error = None
        local = None
        try:
            with _stdout_to_stderr():
                _lib.check(_lib.load().symgpu_comm_available())
        except Exception as exc:
            local = str(exc)
        if self.max_over_ranks(1.0 if local else 0.0) > 0.0:
            return self._fallback(local or 'librccl could not be loaded on another rank')
        raw = (ctypes.c_uint8 * 128)()
        if self.rank == 0:
            try:
                with _stdout_to_stderr():
                    _lib.check(_lib.lib().symgpu_comm_unique_id(ctypes.addressof(raw)))
            except Exception as exc:                              # broadcast an all-zero id: everybody falls back
                local = str(exc)
                raw = (ctypes.c_uint8 * 128)()
        ident = self._bcast_bytes(bytes(raw), 128)
        if not any(ident):
            return self._fallback(local or 'rank 0 could not create an RCCL unique id')
        # ncclCommInitRank is collective and has no timeout of its own: run it on a watchdog thread, so that a bring-up that never
        # returns on some rank (fabric / IPC trouble) ends in the host-staged plane and a `degraded` line instead of a hung job
        raw = (ctypes.c_uint8 * 128)(*ident)
        status, err, th = _run_with_watchdog(lambda: _lib.check(_lib.lib().symgpu_comm_init(ctypes.addressof(raw), self.rank, self.world)),
                                             float(os.environ.get('SYMGPU_RCCL_INIT_TIMEOUT', '180')))
        if status == 2:
            # given up: should ncclCommInitRank come back later, the library destroys its communicator instead of installing it
            _lib.load().symgpu_comm_abandon()
            self._hung_thread = th
            local = 'ncclCommInitRank ' + err
        elif status == 1:
            local = err
        if self.max_over_ranks(1.0 if local else 0.0) > 0.0:
            if local is None:
                _lib.load().symgpu_comm_destroy()
            return self._fallback(local or 'RCCL initialisation failed on another rank')

    # ---- control plane -----------------------------------------------------------------------------------
    def barrier(self):
        if self._tcp is not None:
            self._tcp.max(0.0)
        elif self._dist is not None:
            self._dist.barrier()

    def max_over_ranks(self, x):
        if self._tcp is not None:
            return self._tcp.max(x)
        if self._dist is None:
            return x
        import torch
        t = torch.tensor([float(x)], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t[0])

    # ---- data plane ----------------------------------------------------------------------------------------
    def allgather_op(self, shard, full, n_rows_total):
        """Device-resident: gather every rank's shard (capacity Ts) into ``full`` and trim it to ``n_rows_total``."""
        from . import _lib
        if not self.gathers:
            raise ValueError('allgather_op without a communicator: use the shard directly')
        if self.data_plane == 'rccl':
            def gather():
                _lib.check(_lib.lib().symgpu_comm_allgather_op(shard.handle, full.handle))
            if self._gather_checked:
                gather()
            else:
                def first():
                    gather()
                    _lib.check(_lib.lib().symgpu_sync())            # the collective is asynchronous: a hang shows at the synchronisation
                self._first_gather(first)
        if self.data_plane != 'rccl':
            self._allgather_op_host(shard, full)
        full.set_rows(n_rows_total)

    def _first_gather(self, do_gather):
        """The first collective after the bring-up, under a watchdog, followed by an agreement of all ranks (the same pattern as the
        initialisation): every rank returned -> RCCL stays; an error on any rank -> all ranks destroy their communicator and use
        the host-staged plane (the caller then gathers through it); no return within SYMGPU_RCCL_GATHER_TIMEOUT seconds on any
        rank -> :class:`CollectiveHang` on every rank."""
        from . import _lib
        status, err, th = _run_with_watchdog(do_gather, float(os.environ.get('SYMGPU_RCCL_GATHER_TIMEOUT', '120')))
        if status == 2:
            self._hung_thread = th
        worst = int(self.max_over_ranks(float(status)))
        self._gather_checked = True
        if worst == 2:
            raise CollectiveHang('the first RCCL all-gather ' + (err if status == 2 else 'did not return on another rank'))
        if worst == 1:
            _lib.load().symgpu_comm_destroy()
            self._fallback(('the first RCCL all-gather failed: ' + err) if status == 1 else 'the first RCCL all-gather failed on another rank')

    def mul_cleanup_sharded(self, inner_full, outer_block, inner_is_left=True, zero_threshold=1e-15):
        """Fused product + cleanup of ``inner_full`` (all of one operand, on every rank — gather it with :meth:`allgather_op` if it
        is sharded) with the OUTER operand sharded in contiguous blocks, ranks in order ("cleanup across GPUs").
        Pair ``(i, o)`` has index ``o * Ni + i`` in the reference (``_multiply_by_operator``, base.py:764-794): with the outer index
        sharded contiguously every pair of rank r precedes every pair of rank r + 1, so rank-major order IS the reference's order,
        and the first-occurrence order / sequential sums of ``symplectic_cleanup`` (utils.py:271-278) are those of
        (1) a local fused product + cleanup per rank WITHOUT threshold (a term may only cancel across ranks),
        (2) an all-gather of the cleaned partial operators (padded to the longest, then compacted in rank order),
        (3) one cleanup of the concatenation with the threshold — on every rank (the result is replicated).
        Sums associate per rank (exact for dyadic coefficients, <= 1e-16 relative otherwise), as in the tiled ``kernels.mul_cleanup``.
        Returns a DeviceOp."""
        from . import _lib, kernels
        lib = _lib.lib()
        out = ctypes.c_void_p()
        _lib.check(lib.symgpu_mul_cleanup_dev(inner_full.handle, outer_block.handle, 1 if inner_is_left else 0, 0.0, 0, ctypes.byref(out)))
        part = kernels.DeviceOp(out)
        if not self.gathers:
            res = kernels.cleanup_dev(part, zero_threshold)
            part.free()
            return res
        n_loc, wq, _ = part.info()
        counts = np.frombuffer(self._allgather_bytes(np.int64(n_loc).tobytes()), dtype=np.int64)
        ts = max(1, int(counts.max()))
        shard = kernels.DeviceOp.alloc(ts, wq, with_coeff=True)
        _lib.check(lib.symgpu_op_copy_rows(shard.handle, 0, part.handle, 0, n_loc))
        shard.set_rows(n_loc)
        part.free()
        full = kernels.DeviceOp.alloc(ts * self.world, wq, with_coeff=True)
        self.allgather_op(shard, full, ts * self.world)
        shard.free()
        # compact: rank r's rows [r * ts, r * ts + counts[r]) one after the other (the padding rows are identity rows with coefficient
        # 0: left in place they would be harmless for the sums but could take the identity's first-occurrence position)
        total = int(counts.sum())
        cat = kernels.DeviceOp.alloc(max(1, total), wq, with_coeff=True)
        pos = 0
        for r in range(self.world):
            _lib.check(lib.symgpu_op_copy_rows(cat.handle, pos, full.handle, r * ts, int(counts[r])))
            pos += int(counts[r])
        cat.set_rows(total)
        full.free()
        res = kernels.cleanup_dev(cat, zero_threshold)
        cat.free()
        return res

    def mul_cleanup_hash_partitioned(self, inner_full, outer_full, inner_is_left=True, zero_threshold=1e-15, stats=None, gather=True):
        """Fused product + cleanup of two device operators that are COMPLETE on every rank (after :meth:`allgather_op`), the pairs partitioned
        over the ranks by the GF(2)-linear class of their product row (:func:`hash_partition_local_dev`): all duplicates of a row meet on one
        rank — also the twins (i, o) / (o, i) of a squared operator, which the contiguous split of :meth:`mul_cleanup_sharded` leaves on
        different ranks — no key and no partial product crosses the wire.  The rank's share is computed, ordered and merged on the device.
        ``gather=False`` returns this rank's share (a DeviceOp carrying the pair index of each term's first occurrence: the result stays
        sharded); otherwise the shares are all-gathered (rows and coefficients through :meth:`allgather_op`, the 8-byte indices over the
        control plane) and ordered by pair index on every rank: the replicated result in the reference's first-occurrence order."""
        from . import _lib, kernels
        share = hash_partition_local_dev(inner_full, outer_full, self.rank, self.world, inner_is_left, zero_threshold, stats)
        if not gather or not self.gathers:
            return share
        n_loc, wq, _ = share.info()
        counts = np.frombuffer(self._allgather_bytes(np.int64(n_loc).tobytes()), dtype=np.int64)
        ts = max(1, int(counts.max()))
        g_loc = np.zeros(ts, dtype='<u8'); g_loc[:n_loc] = kernels.op_first_index(share)
        all_g = np.frombuffer(self._allgather_bytes(g_loc.tobytes()), dtype='<u8').reshape(self.world, ts)
        shard = kernels.DeviceOp.alloc(ts, wq, with_coeff=True)
        _lib.check(_lib.lib().symgpu_op_copy_rows(shard.handle, 0, share.handle, 0, n_loc))
        shard.set_rows(n_loc)
        share.free()
        full = kernels.DeviceOp.alloc(ts * self.world, wq, with_coeff=True)
        self.allgather_op(shard, full, ts * self.world)
        shard.free()
        parts = []
        try:
            for r in range(self.world):
                part = kernels.DeviceOp.alloc(max(1, int(counts[r])), wq, with_coeff=True)
                _lib.check(_lib.lib().symgpu_op_copy_rows(part.handle, 0, full.handle, r * ts, int(counts[r])))
                part.set_rows(int(counts[r]))
                kernels.op_set_first_index(part, all_g[r, :counts[r]])
                parts.append(part)
            full.free()
            key_bits = max(1, int(inner_full.n_terms * outer_full.n_terms - 1).bit_length())
            return kernels.merge_indexed_dev(parts, key_bits, False)
        finally:
            for p in parts:
                p.free()

    def verify_allgather(self, shard, full, n_rows_total):
        """Self-check of the RCCL data plane (call once, outside any timed region): gather the same shards a second time through
        host memory over the control plane and compare them with the device result of ``allgather_op`` — rows and coefficients,
        bit for bit, at their global positions.  If ANY rank sees a difference, all ranks leave RCCL for the host-staged plane
        (``degraded`` says why).  Returns True when the data plane in use after the call delivered the right operand."""
        from . import _lib
        if not self.gathers or self.data_plane != 'rccl':
            return True
        t, wq, ts = shard.info()
        try:
            r, c = shard.download()
            fr, fc = full.download()
        except _lib.SymgpuError:
            r, c = shard.download(with_coeff=False), None
            fr, fc = full.download(with_coeff=False), None
        rows = np.zeros((ts, 2 * wq), dtype='<u8'); rows[:t] = r
        exp_rows = np.frombuffer(self._allgather_bytes(rows.tobytes()), dtype='<u8').reshape(self.world * ts, 2 * wq)[:n_rows_total]
        ok = fr.shape == exp_rows.shape and np.array_equal(fr, exp_rows)
        if c is not None:
            coeff = np.zeros(ts, dtype=np.complex128); coeff[:t] = c
            exp_c = np.frombuffer(self._allgather_bytes(coeff.tobytes()), dtype=np.complex128)[:n_rows_total]
            ok = ok and np.array_equal(fc.view(np.float64), exp_c.view(np.float64))
        if self.max_over_ranks(0.0 if ok else 1.0) > 0.0:
            _lib.load().symgpu_comm_destroy()
            self._fallback('the RCCL all-gather delivered rows that differ from the shards')
            self.allgather_op(shard, full, n_rows_total)
            return False
        return True

    def _allgather_bytes(self, payload):
        if self._tcp is not None:
            return self._tcp.allgather(payload)
        import torch
        mine = torch.frombuffer(bytearray(payload), dtype=torch.uint8)
        parts = [torch.zeros_like(mine) for _ in range(self.world)]
        self._dist.all_gather(parts, mine)
        return b''.join(bytes(p.numpy().tobytes()) for p in parts)

    def _allgather_op_host(self, shard, full):
        """The same gather through host memory: shard (padded with identity rows to its capacity Ts) -> D2H -> control plane
        -> H2D into ``full`` at the rows' global indices."""
        from . import _lib
        t, wq, ts = shard.info()
        rows = np.zeros((ts, 2 * wq), dtype='<u8')
        coeff = np.zeros(ts, dtype=np.complex128)
        try:
            r, c = shard.download()
        except _lib.SymgpuError:                                  # rows-only operand (commutation needs no coefficients)
            r, c = shard.download(with_coeff=False), None
        rows[:t] = r
        all_rows = np.frombuffer(self._allgather_bytes(rows.tobytes()), dtype='<u8').reshape(self.world * ts, 2 * wq)
        all_coeff = None
        if c is not None:
            coeff[:t] = c
            all_coeff = np.frombuffer(self._allgather_bytes(coeff.tobytes()), dtype=np.complex128)
        _lib.check(_lib.lib().symgpu_op_write(full.handle, 0, all_rows.ctypes.data, None if all_coeff is None else all_coeff.ctypes.data,
                                              self.world * ts))

    def allgather_rows_host(self, local_rows, n_rows_total):
        """Host arrays over gloo (CPU tests / PCIe staging): returns the full ``uint64[n_rows_total, W]`` array."""
        local_rows = np.ascontiguousarray(local_rows, dtype='<u8')
        if self.world == 1:
            return local_rows
        assert self._dist is not None, 'host all-gather needs the gloo control plane'
        import torch
        ts, _ = shard_bounds(n_rows_total, self.world)
        W = local_rows.shape[1]
        pad = np.zeros((ts, W), dtype=np.int64)
        pad[:local_rows.shape[0]] = local_rows.view(np.int64)
        parts = [torch.zeros((ts, W), dtype=torch.int64) for _ in range(self.world)]
        self._dist.all_gather(parts, torch.from_numpy(pad))
        return np.concatenate([p.numpy() for p in parts], axis=0)[:n_rows_total].view('<u8')

    def close(self):
        hung = self._hung_thread is not None and self._hung_thread.is_alive()
        if self.data_plane == 'rccl' and self.gathers and not hung:
            from . import _lib
            _lib.load().symgpu_comm_destroy()
        if self._tcp is not None:
            self._tcp.max(0.0)
            self._tcp.close()
            self._tcp = None
        if self._dist is not None and self._dist.is_initialized():
            self._dist.barrier()
            self._dist.destroy_process_group()
            self._dist = None
        # a thread of this process still inside RCCL: tearing the runtime down under it at interpreter exit can crash or hang — the
        # caller should finish its output and then leave through hard_exit_if_hung()
        self.needs_hard_exit = hung

    HUNG_EXIT_STATUS = 4

    def hard_exit_if_hung(self, status=None):
        """Call after the last line of output: if a watchdog thread never came back from RCCL, end the process without running the
        finalisers (library shutdown under a thread that is inside RCCL is not safe).  The exit status is NON-ZERO (4) by default:
        a thread stuck inside RCCL means the run did not use the data plane it was asked to use (the JSON line says ``degraded``),
        and a launcher that only reads exit codes must see that."""
        if getattr(self, 'needs_hard_exit', False):
            import sys
            sys.stdout.flush(); sys.stderr.flush()
            os._exit(self.HUNG_EXIT_STATUS if status is None else status)


def padded_random_shard(my_rows, ts, n_qubits, seed):
    """Synthetic right-operand shard with capacity ``ts`` (the all-gather pads the unused tail with zeros)."""
    from .kernels import DeviceOp
    op = DeviceOp.random(ts, n_qubits, 0.3, seed)
    op.set_rows(my_rows)
    return op


def sharded_commutes(a_rows_global, b_rows_local, n_b_total, comm, kernel=None):
    """This rank's block of ``commutes_termwise``: rows ``[begin, end)`` of the left operand against the WHOLE right
    operand (gathered from the per-rank shards).  ``kernel(a_block, b_full) -> bool[rows, M]`` defaults to the HIP
    commutation kernel; the CPU tests inject a checker to exercise the sharding/gather logic without a GPU."""
    if kernel is None:
        from . import kernels
        kernel = kernels.commutes
    _, bounds = shard_bounds(a_rows_global.shape[0], comm.world)
    b0, b1 = bounds[comm.rank]
    b_full = comm.allgather_rows_host(b_rows_local, n_b_total)
    return (b0, b1), kernel(np.ascontiguousarray(a_rows_global[b0:b1]), b_full)


def sharded_product(a_rows_global, a_coeff_global, b_rows_local, b_coeff_local, n_b_total, comm, kernel=None):
    """This rank's slab of the uncleaned product ``A * B`` with A (left factor) sharded along its term axis and B
    gathered: returns ((begin, end), rows, coeff) where row ``o*(end-begin) + i`` = ``A[begin+i] ^ B[o]``."""
    if kernel is None:
        from . import kernels
        kernel = kernels.mul_allpairs
    _, bounds = shard_bounds(a_rows_global.shape[0], comm.world)
    b0, b1 = bounds[comm.rank]
    b_full = comm.allgather_rows_host(b_rows_local, n_b_total)
    c = np.ascontiguousarray(b_coeff_local, dtype=np.complex128).view(np.float64).reshape(-1, 2)
    c_full = comm.allgather_rows_host(c.view('<u8'), n_b_total).view(np.float64).reshape(-1, 2).copy().view(np.complex128).reshape(-1)
    rows, coeff = kernel(np.ascontiguousarray(a_rows_global[b0:b1]), a_coeff_global[b0:b1], b_full, c_full, True)
    return (b0, b1), rows, coeff


def sharded_mul_cleanup(inner_rows, inner_coeff, outer_rows_local, outer_coeff_local, comm, inner_is_left=True, zero_threshold=1e-15,
                        mul_kernel=None, cleanup_kernel=None):
    """Host arrays, gloo (CPU tests / the algorithm of :meth:`Communicator.mul_cleanup_sharded`): this rank holds ALL of the inner
    operand and its contiguous block of the outer one (ranks in order).  ``mul_kernel(inner, ci, outer, co, inner_is_left, thr)`` =
    fused product + cleanup (thr None: keep everything), ``cleanup_kernel(rows, coeff, thr)`` = first-occurrence cleanup (the CPU
    tests inject their reference checker).  Returns the cleaned product (rows, coeff), identical on every rank."""
    if mul_kernel is None or cleanup_kernel is None:
        from . import kernels
        mul_kernel = mul_kernel or (lambda a, ca, b, cb, left, thr: kernels.mul_cleanup(a, ca, b, cb, left, thr))
        cleanup_kernel = cleanup_kernel or kernels.cleanup
    rows, coeff = mul_kernel(inner_rows, inner_coeff, outer_rows_local, outer_coeff_local, inner_is_left, None)
    rows = np.ascontiguousarray(rows, dtype='<u8'); coeff = np.ascontiguousarray(coeff, dtype=np.complex128)
    if comm.world == 1:
        return cleanup_kernel(rows, coeff, zero_threshold)
    W = inner_rows.shape[1]
    counts = np.frombuffer(comm._allgather_bytes(np.int64(rows.shape[0]).tobytes()), dtype=np.int64)
    ts = max(1, int(counts.max()))
    pad_r = np.zeros((ts, W), dtype='<u8'); pad_r[:rows.shape[0]] = rows
    pad_c = np.zeros(ts, dtype=np.complex128); pad_c[:coeff.shape[0]] = coeff
    all_r = np.frombuffer(comm._allgather_bytes(pad_r.tobytes()), dtype='<u8').reshape(comm.world, ts, W)
    all_c = np.frombuffer(comm._allgather_bytes(pad_c.tobytes()), dtype=np.complex128).reshape(comm.world, ts)
    cat_r = np.concatenate([all_r[r, :counts[r]] for r in range(comm.world)], axis=0)
    cat_c = np.concatenate([all_c[r, :counts[r]] for r in range(comm.world)])
    return cleanup_kernel(cat_r, cat_c, zero_threshold)


# ---- cleanup across GPUs by HASH PARTITION ("cleanup across GPUs") -----------------------------------
# sharded_mul_cleanup above splits the OUTER index in contiguous blocks: every rank cleans its own slab, gathers every other rank's cleaned
# slab and cleans the concatenation again.  That is right for products whose duplicates sit close together, but the twins (i, o) / (o, i)
# of a squared operator live on different ranks, nothing merges locally, and every rank receives and re-cleans (nearly) the whole product.
#
# Here the PAIRS are partitioned by a GF(2)-LINEAR class of their product row: cls(row) = parities of the row under a few fixed random
# masks, so cls(inner[i] ^ outer[o]) = cls(inner[i]) ^ cls(outer[o]) — the owner of a pair, (cls_i ^ cls_o) mod G, is known from the two
# operand tables alone, equal product rows always have the same owner, and the pairs of one owner are a union of full sub-products
# (inner rows of class a) x (outer rows whose class XOR a maps to the owner).  No key is generated for a pair that another rank owns and
# NOTHING is exchanged before the result: every duplicate of a row meets its partners on its owner, which applies the threshold there.
# What does cross the wire is each rank's share of the FINAL result (rows, coefficients and the pair index of each row's first
# occurrence), all-gathered and put in the reference's first-occurrence order (utils.py:271) by that index.
# Precondition: both operands complete on every rank (the all-gather of the right operand that the product path performs anyway).
# Sums associate per sub-product (in pair order inside it, the sub-products' sums then in order of their first pair): bit-exact for
# dyadic coefficients, <= 1e-16 relative otherwise — the same statement as for the tiled kernels.mul_cleanup.
def linear_row_classes(rows, n_bits, seed=0x51A55E5):
    """GF(2)-linear class in [0, 2^n_bits) of every packed row: bit k = parity of the row under the k-th fixed random mask."""
    rows = np.ascontiguousarray(rows, dtype='<u8')
    rng = np.random.default_rng(seed)
    masks = rng.integers(0, 1 << 63, size=(n_bits, rows.shape[1]), dtype=np.uint64) ^ (rng.integers(0, 2, size=(n_bits, rows.shape[1]), dtype=np.uint64) << np.uint64(63))
    cls = np.zeros(rows.shape[0], dtype=np.int64)
    for k in range(n_bits):
        par = packing.popcount_rows(rows & masks[k][None, :]) & 1
        cls |= par << k
    return cls


def hash_partition_local(inner_rows, inner_coeff, outer_rows, outer_coeff, rank, world, inner_is_left=True, zero_threshold=1e-15,
                         mul_kernel=None, cleanup_kernel=None, stats=None):
    """Rank ``rank``'s share of the cleaned product: the terms whose product row's linear class maps to this rank — complete (every duplicate
    of such a row is among this rank's pairs), thresholded, in first-occurrence order, with the pair index ``o * Ni + i`` of each term's
    first occurrence.  No communication.  -> (rows, coeff, first_pair_index)."""
    if mul_kernel is None or cleanup_kernel is None:
        from . import kernels
        mul_kernel = mul_kernel or (lambda a, ca, b, cb, left: kernels.mul_cleanup_indexed(a, ca, b, cb, left, None))
        cleanup_kernel = cleanup_kernel or kernels.cleanup_indexed
    inner_rows = np.ascontiguousarray(inner_rows, dtype='<u8'); outer_rows = np.ascontiguousarray(outer_rows, dtype='<u8')
    inner_coeff = np.ascontiguousarray(inner_coeff, dtype=np.complex128); outer_coeff = np.ascontiguousarray(outer_coeff, dtype=np.complex128)
    Ni, No, W = inner_rows.shape[0], outer_rows.shape[0], inner_rows.shape[1]
    G = world
    n_bits = 1
    while (1 << n_bits) < G:
        n_bits += 1
    if G & (G - 1):
        n_bits += 2                                                # a class count that is not a multiple of G: four times as many classes even the shares out
    n_cls = 1 << n_bits
    cls_i = linear_row_classes(inner_rows, n_bits); cls_o = linear_row_classes(outer_rows, n_bits)
    parts_r, parts_c, parts_g = [], [], []
    pairs_owned = 0
    for a in range(n_cls):
        ia = np.flatnonzero(cls_i == a)                                                    # ascending: local pair order = global pair order restricted
        oa = np.flatnonzero(((cls_o ^ a) % G) == rank)
        if ia.size == 0 or oa.size == 0:
            continue
        pairs_owned += ia.size * oa.size
        r, c, i_f, o_f = mul_kernel(inner_rows[ia], inner_coeff[ia], outer_rows[oa], outer_coeff[oa], inner_is_left)
        parts_r.append(r); parts_c.append(c)
        parts_g.append(oa[np.asarray(o_f, dtype=np.int64)] * Ni + ia[np.asarray(i_f, dtype=np.int64)])   # pair index of the first occurrence (base.py:783-792)
    if parts_r:
        rows = np.concatenate(parts_r, axis=0); coeff = np.concatenate(parts_c); g = np.concatenate(parts_g)
        order = np.argsort(g, kind='stable')
        rows, coeff, g = np.ascontiguousarray(rows[order]), coeff[order], g[order]
        rows, coeff, first = cleanup_kernel(rows, coeff, zero_threshold)                   # duplicates across this rank's sub-products; the threshold: final
        g = g[np.asarray(first, dtype=np.int64)]
    else:
        rows, coeff, g = np.zeros((0, W), dtype='<u8'), np.zeros(0, dtype=np.complex128), np.zeros(0, dtype=np.int64)
    if stats is not None:
        stats.update(pairs_owned=int(pairs_owned), pairs_total=int(Ni) * int(No), keys_exchanged=0,
                     bytes_sent=int(rows.nbytes + coeff.nbytes + g.nbytes) if G > 1 else 0)
    return np.ascontiguousarray(rows, dtype='<u8'), np.ascontiguousarray(coeff), g


def hash_partition_local_dev(inner, outer, rank, world, inner_is_left=True, zero_threshold=1e-15, stats=None, classes=None, max_pairs=None):
    """:func:`hash_partition_local` with the operands and every intermediate on the DEVICE (``csrc/partition.hip``): ``inner`` / ``outer`` are
    complete DeviceOps; sub-operands are gathered on the device, each sub-product is one indexed fused product + cleanup, the parts are ordered
    by pair index and merged there.  Only the operands' rows visit the host once, for their class bits (``classes`` = (cls_i, cls_o) skips
    that).  Returns a DeviceOp that carries the pair index of every term's first occurrence (``kernels.op_first_index``).
    One indexed product call handles fewer than 2^32 pairs (``symgpu_mul_cleanup_indexed_dev``): a sub-product beyond ``max_pairs`` is cut
    along its OUTER index into pieces below it — the pieces are further parts of the same merge, which orders everything by pair index."""
    from . import kernels
    max_pairs = int(max_pairs) if max_pairs else (1 << 31)
    Ni, wq, _ = inner.info()
    No = outer.info()[0]
    G = world
    n_bits = 1
    while (1 << n_bits) < G:
        n_bits += 1
    if G & (G - 1):
        n_bits += 2
    if classes is None:
        cls_i = linear_row_classes(inner.download(with_coeff=False), n_bits)
        cls_o = cls_i if outer is inner else linear_row_classes(outer.download(with_coeff=False), n_bits)
    else:
        cls_i, cls_o = classes
    parts, pairs_owned = [], 0
    try:
        for a in range(1 << n_bits):
            ia = np.flatnonzero(cls_i == a)
            oa = np.flatnonzero(((cls_o ^ a) % G) == rank)
            if ia.size == 0 or oa.size == 0:
                continue
            pairs_owned += ia.size * oa.size
            sub_i = kernels.op_gather(inner, ia)
            try:
                step = max(1, max_pairs // int(ia.size))                                   # outer rows per call: ia.size * step <= max_pairs < 2^32
                for o0 in range(0, int(oa.size), step):
                    oc = oa[o0:o0 + step]
                    sub_o = kernels.op_gather(outer, oc)
                    try:
                        part = kernels.mul_cleanup_indexed_dev(sub_i, sub_o, inner_is_left, None)
                    finally:
                        sub_o.free()
                    kernels.part_global_index(part, ia, oc, Ni)
                    parts.append(part)
            finally:
                sub_i.free()
        key_bits = max(1, int(Ni * No - 1).bit_length())
        if parts:
            res = kernels.merge_indexed_dev(parts, key_bits, True, zero_threshold)
        else:
            res = kernels.DeviceOp.alloc(1, wq, with_coeff=True)
            kernels.op_set_first_index(res, np.zeros(0, dtype='<u8'))
    finally:
        for p in parts:
            p.free()
    if stats is not None:
        stats.update(pairs_owned=int(pairs_owned), pairs_total=int(Ni) * int(No), keys_exchanged=0,
                     bytes_sent=int(res.n_terms * (16 * wq + 16 + 8)) if G > 1 else 0)
    return res


def hash_partitioned_mul_cleanup(inner_rows, inner_coeff, outer_rows, outer_coeff, comm, inner_is_left=True, zero_threshold=1e-15,
                                 mul_kernel=None, cleanup_kernel=None, stats=None):
    """``inner * outer`` (or ``outer * inner``) + cleanup with the PAIRS partitioned over the ranks by the linear class of their product
    row.  Host arrays in, the cleaned product (rows, coeff) out, identical on every rank and equal to the single-process result.
    ``mul_kernel(inner, ci, outer, co, inner_is_left) -> (rows, coeff, i_first, o_first)`` = fused product + cleanup WITHOUT threshold plus the
    first pair of every output row (default ``kernels.mul_cleanup_indexed``); ``cleanup_kernel(rows, coeff, thr) -> (rows, coeff, first)``
    (default ``kernels.cleanup_indexed``).  The CPU tests inject their own checker kernels.  ``stats`` (dict, optional) receives
    ``pairs_owned``, ``pairs_total``, ``keys_exchanged`` (always 0) and ``bytes_sent`` of this rank."""
    G = comm.world
    rows, coeff, g = hash_partition_local(inner_rows, inner_coeff, outer_rows, outer_coeff, comm.rank, G, inner_is_left, zero_threshold,
                                          mul_kernel, cleanup_kernel, stats)
    if G == 1:
        return rows, coeff
    W = rows.shape[1]
    # the result: every rank's share, all-gathered and ordered by first pair index
    counts = np.frombuffer(comm._allgather_bytes(np.int64(rows.shape[0]).tobytes()), dtype=np.int64)
    ts = max(1, int(counts.max()))
    pad_r = np.zeros((ts, W), dtype='<u8'); pad_r[:rows.shape[0]] = rows
    pad_c = np.zeros(ts, dtype=np.complex128); pad_c[:coeff.shape[0]] = coeff
    pad_g = np.zeros(ts, dtype=np.int64); pad_g[:g.shape[0]] = g
    all_r = np.frombuffer(comm._allgather_bytes(pad_r.tobytes()), dtype='<u8').reshape(G, ts, W)
    all_c = np.frombuffer(comm._allgather_bytes(pad_c.tobytes()), dtype=np.complex128).reshape(G, ts)
    all_g = np.frombuffer(comm._allgather_bytes(pad_g.tobytes()), dtype=np.int64).reshape(G, ts)
    cat_r = np.concatenate([all_r[r, :counts[r]] for r in range(G)], axis=0)
    cat_c = np.concatenate([all_c[r, :counts[r]] for r in range(G)])
    cat_g = np.concatenate([all_g[r, :counts[r]] for r in range(G)])
    order = np.argsort(cat_g, kind='stable')
    return np.ascontiguousarray(cat_r[order]), np.ascontiguousarray(cat_c[order])
