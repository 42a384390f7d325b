"""Multi-GPU seam: independent SMC runs / SMC^2 islands sharded over one
process per GPU (the reference's ``multiSMC`` -> ``utils.distribute_work`` fan
out to loky worker processes, particles/core.py:431, utils.py:158-186).

The data path has no exchange step: each rank owns ``islands_per_rank`` whole
filters.  The only collective is the gather of the per-island log-evidences,
done with RCCL over xGMI inside libsmc_hip (``smc_comm_*``).

Host-side rendezvous needs no framework: the launcher's environment (``RANK``,
``WORLD_SIZE``, ``MASTER_ADDR``, ``MASTER_PORT`` -- what ``torch.distributed.run``
or any other one-process-per-GPU launcher exports) is all it reads.  Rank 0
listens on a TCP port of its own and the other ranks of the node find it through
a key file; over that star every host-side operation is one ``exchange`` (an
all-gather of a few bytes through rank 0): barrier, the max-over-ranks of the
timings, and handing the 128-byte RCCL unique id to everybody.  No torch, no
gloo, no MPI in the product path.

If RCCL cannot be initialised the Group RAISES: a scale run must not silently
measure something that never touched xGMI.  ``SMC_ALLOW_HOST_GATHER=1`` (or
``device_collective=False``, CPU tests) routes the 8 bytes per island through the
TCP star instead and says so in ``fallback_reason``.
"""
import ctypes
import os
import socket
import struct
import tempfile
import time

import numpy as np

from . import _lib
from ._lib import DeviceArray, check, lib


class _StdoutToStderr:
    """RCCL prints a version banner on STDOUT when it initialises; a launch line that promises ONE JSON line on
    stdout (bench.py) must not carry it: file descriptor 1 points at stderr while the library initialises."""

    def __enter__(self):
        import sys
        try:
            sys.stdout.flush()
            self._saved = os.dup(1)
            os.dup2(2, 1)
        except OSError:
            self._saved = None
        return self

    def __exit__(self, *exc):
        if self._saved is not None:
            os.dup2(self._saved, 1)
            os.close(self._saved)
        return False


def env_rank_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_islands(n_total, rank, world):
    """Contiguous block partition of ``n_total`` islands: (first, count) of
    ``rank``.  Island ids are global, so the Philox streams (and hence the
    results) do not depend on how many GPUs the job runs on."""
    base, extra = divmod(n_total, world)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def _send_msg(sock, payload, key=None, ctx=b""):
    """Length-prefixed message; with `key` (the launch's secret) an HMAC-SHA256 tag over ctx + payload follows the
    payload -- ctx = direction and the connection's message counter, so a message can be neither replayed nor
    reflected."""
    if key is not None:
        import hashlib
        import hmac
        payload = payload + hmac.new(key, ctx + payload, hashlib.sha256).digest()
    sock.sendall(struct.pack("<I", len(payload)) + payload)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("rendezvous peer closed the connection")
        buf += chunk
    return bytes(buf)


def _recv_msg(sock, key=None, ctx=b"", maxlen=1 << 31):
    (n,) = struct.unpack("<I", _recv_exact(sock, 4))
    if n > maxlen:                       # (the handshake's messages are tiny: nobody gets to make rank 0 buffer gigabytes)
        raise ConnectionError("rendezvous: message of %d bytes where at most %d are expected" % (n, maxlen))
    data = _recv_exact(sock, n)
    if key is None:
        return data
    # every message behind the handshake is authenticated with the launch's secret (the 0600 key file): a process that
    # merely reaches the port cannot feed the ranks a pickle (allgather_obj / gather_outputs unpickle what arrives)
    import hashlib
    import hmac
    if n < 32 or not hmac.compare_digest(hmac.new(key, ctx + data[:-32], hashlib.sha256).digest(), data[-32:]):
        raise ConnectionError("rendezvous: message authentication failed")
    return data[:-32]


def _mac(key, *parts):
    import hashlib
    import hmac
    return hmac.new(key, b"|".join(parts), hashlib.sha256).hexdigest().encode()


class _Star:
    """All ranks of one node connected to rank 0.  The one primitive: ``exchange(payload)``
    -> list of every rank's payload in rank order (an all-gather through rank 0)."""

    def __init__(self, rank, world, timeout=120.0):
        self.rank, self.world = rank, world
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        # the launcher may itself be listening on MASTER_PORT (torchrun's store does): rank 0
        # binds a port of its own and publishes it in a file keyed by what all ranks of THIS
        # launch share -- MASTER_PORT and the launcher's pid
        # (launchers that put a wrapper between themselves and the ranks -- `bash -c`, srun -- give every rank a
        #  different parent: SMC_RDZV_KEY, exported to all ranks of ONE launch, replaces the parent's pid)
        launch = os.environ.get("SMC_RDZV_KEY") or str(os.getppid())
        key = "smc_rdzv_%s_%s_%s" % (os.environ.get("MASTER_PORT", "0"),
                                     os.environ.get("TORCHELASTIC_RUN_ID", "x"), launch)
        self._keyfile = os.path.join(tempfile.gettempdir(), key)
        self.peers = []
        self.sock = None
        # rank 0 draws a RANDOM secret per launch and publishes it with its port in the 0600 key file.  The secret never
        # crosses the socket: a peer says which rank it is, rank 0 answers with a random challenge, the peer returns
        # HMAC(secret, rank, challenge) and gets HMAC(secret, "ok", rank, challenge) back -- both ends have then proved
        # they read the file.  The key file of a crashed earlier launch (same name: same MASTER_PORT, run id and parent)
        # carries another secret and a dead port: a peer that reads it is refused (or fails the challenge of the new
        # rank 0) and reads again.  Every later message is authenticated under (direction, per-connection counter).
        nonce = None
        if rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr if addr not in ("localhost",) else "127.0.0.1", 0))
            srv.listen(world)
            srv.settimeout(timeout)
            # a stale file from a crashed launch goes first; the new one is created exclusively
            # (no following of a symlink somebody planted in the shared temp directory), mode 0600
            import secrets
            nonce = secrets.token_hex(12)
            fd = None
            for attempt in range(50):              # (somebody re-creating the name in between: try again)
                try:
                    os.unlink(self._keyfile)
                except OSError:
                    pass
                try:
                    fd = os.open(self._keyfile, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
                    break
                except FileExistsError:
                    time.sleep(0.01)
            if fd is None:
                srv.close()
                raise RuntimeError("rendezvous: cannot create %s (somebody keeps re-creating it)" % self._keyfile)
            with os.fdopen(fd, "w") as fh:
                fh.write("%d %d %s\n" % (srv.getsockname()[1], os.getpid(), nonce))
            conns = {}
            try:
                while len(conns) < world - 1:
                    c, _ = srv.accept()
                    try:
                        c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        c.settimeout(timeout)
                        hello = _recv_msg(c, maxlen=64).decode(errors="replace")
                        if not hello.isdigit() or not 0 < int(hello) < world or int(hello) in conns:
                            raise ValueError("not a rank of this launch")
                        challenge = secrets.token_hex(16).encode()
                        _send_msg(c, challenge)
                        import hmac as _hmac
                        answer = _recv_msg(c, maxlen=128)
                        if not _hmac.compare_digest(answer, _mac(nonce.encode(), b"hello", hello.encode(), challenge)):
                            raise ValueError("challenge failed")
                        hello = [hello]
                    except (OSError, ConnectionError, ValueError, struct.error):
                        c.close()                      # a half-open or garbled hello, a stranger: not our business
                        continue
                    _send_msg(c, _mac(nonce.encode(), b"ok", hello[0].encode(), challenge))
                    # (the rendezvous timeout ends here: the collectives that follow wait as long as the slowest
                    #  rank's work takes -- uneven shares, a first-use build on one rank)
                    c.settimeout(None)
                    conns[int(hello[0])] = c
            except Exception:
                for c in conns.values():
                    c.close()
                srv.close()
                self.close()
                raise
            srv.close()
            self.peers = [conns[r] for r in range(1, world)]
            self._key = nonce.encode()
        else:
            t0 = time.time()
            s = None
            while s is None:
                # (re-)read the file until it names a live listener of THIS launch: a stale file is
                # replaced by rank 0 a moment later, and a refused connection means exactly that
                try:
                    with open(self._keyfile) as fh:
                        f = fh.read().split()
                    if len(f) != 3 or not f[0].isdigit() or not f[1].isdigit():
                        raise ValueError("not a key file")
                    os.kill(int(f[1]), 0)              # the rank 0 that wrote it is alive (else: OSError)
                    nonce = f[2]
                    s = socket.create_connection((addr, int(f[0])), timeout=timeout)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    s.settimeout(timeout)
                    _send_msg(s, b"%d" % rank)
                    challenge = _recv_msg(s, maxlen=128)
                    _send_msg(s, _mac(nonce.encode(), b"hello", b"%d" % rank, challenge))
                    import hmac as _hmac
                    if not _hmac.compare_digest(_recv_msg(s, maxlen=128), _mac(nonce.encode(), b"ok", b"%d" % rank, challenge)):
                        raise ValueError("rejected")   # (a live process that is not this launch's rank 0)
                    s.settimeout(None)
                except (OSError, ValueError, IndexError, ConnectionError):
                    if s is not None:
                        s.close()
                    s = None
                    if time.time() - t0 > timeout:
                        raise TimeoutError("rendezvous: rank 0 never published a live %s" % self._keyfile)
                    time.sleep(0.01)
            self.sock = s
            self._key = nonce.encode()

    def exchange(self, payload=b""):
        # (one round = one message up and one down on every connection: both ends count rounds)
        seq = struct.pack("<Q", getattr(self, "_round", 0))
        self._round = getattr(self, "_round", 0) + 1
        if self.rank == 0:
            parts = [payload] + [_recv_msg(c, self._key, b"u" + seq) for c in self.peers]
            blob = b"".join(struct.pack("<I", len(p)) + p for p in parts)
            for c in self.peers:
                _send_msg(c, blob, self._key, b"d" + seq)
            return parts
        _send_msg(self.sock, payload, self._key, b"u" + seq)
        blob = _recv_msg(self.sock, self._key, b"d" + seq)
        parts, off = [], 0
        for _ in range(self.world):
            (n,) = struct.unpack_from("<I", blob, off)
            parts.append(blob[off + 4:off + 4 + n])
            off += 4 + n
        return parts

    def close(self):
        for c in self.peers:
            c.close()
        if self.sock is not None:
            self.sock.close()
        if self.rank == 0:
            try:
                os.unlink(self._keyfile)
            except OSError:
                pass
        self.peers, self.sock = [], None


class Group:
    """Process group of a one-node launch: a TCP star for the host side, RCCL (inside
    libsmc_hip) for the device-side gather of the evidences.  ``device_collective=False``
    keeps the gather on the star as well (CPU-only tests)."""

    def __init__(self, device_collective=True):
        self.rank, self.local_rank, self.world = env_rank_world()
        self.star = _Star(self.rank, self.world) if self.world > 1 else None
        self.comm = None
        self.fallback_reason = None if device_collective else "device_collective=False"
        if device_collective:
            err = self._init_rccl()
            if err:
                if os.environ.get("SMC_ALLOW_HOST_GATHER") != "1":
                    self.close()
                    raise RuntimeError("RCCL evidence gather unavailable (%s); set SMC_ALLOW_HOST_GATHER=1 "
                                       "to gather over the host rendezvous instead" % err)
                self.fallback_reason = err

    def _init_rccl(self):
        """Every rank learns the unique id (or that rank 0 could not make one) in ONE exchange, then
        all attempt ncclCommInitRank together and agree on the outcome: no rank is left waiting in
        a collective the others never enter.  Returns None, or the reason RCCL is not usable."""
        uid = ctypes.create_string_buffer(128)
        msg = b""
        if self.rank == 0:
            try:
                with _StdoutToStderr():
                    check(lib().smc_comm_unique_id(uid))
                msg = b"\x01" + uid.raw
            except Exception as e:
                msg = b"\x00" + str(e).encode()
        first = self.star.exchange(msg)[0] if self.star else msg
        if first[:1] != b"\x01":
            return "rank 0: " + first[1:].decode(errors="replace")
        uid = ctypes.create_string_buffer(first[1:129], 128)
        h = _lib.c_vp()
        err = b""
        try:
            with _StdoutToStderr():
                check(lib().smc_comm_create(_lib.ctx().h, self.world, self.rank, uid, ctypes.byref(h)))
        except Exception as e:
            err = ("rank %d: %s" % (self.rank, e)).encode()
        errs = self.star.exchange(err) if self.star else [err]
        bad = [e for e in errs if e]
        if bad:
            if not err:
                lib().smc_comm_destroy(h)
            return bad[0].decode(errors="replace")
        self.comm = h
        return None

    # ---- host-side helpers (TCP star) -----------------------------------------
    def barrier(self):
        if self.star:
            self.star.exchange(b"")

    def allgather_str(self, text):
        """Every rank's string, in rank order (host side)."""
        if not self.star:
            return [text]
        return [p.decode(errors="replace") for p in self.star.exchange(text.encode())]

    def _allreduce_host(self, v, op):
        a = np.atleast_1d(np.asarray(v, dtype=np.float64))
        if self.star:
            parts = self.star.exchange(a.tobytes())
            a = op(np.stack([np.frombuffer(p, dtype=np.float64) for p in parts]), axis=0)
        return float(a[0]) if np.ndim(v) == 0 else a

    def allreduce_max_host(self, v):
        """Element-wise maximum over ranks of a scalar or vector (the timings)."""
        return self._allreduce_host(v, np.max)

    def allreduce_min_host(self, v):
        return self._allreduce_host(v, np.min)

    def broadcast_host(self, arr):
        """Rank 0's array on every rank (host side; a few bytes: seeds, flags)."""
        a = np.ascontiguousarray(arr)
        if not self.star:
            return a
        blob = self.star.exchange(a.tobytes() if self.rank == 0 else b"")[0]
        return np.frombuffer(blob, dtype=a.dtype).reshape(a.shape).copy()

    def allgather_obj(self, obj):
        """Every rank's picklable object, in rank order (host side).  Trust boundary: the bytes unpickled here arrived
        over the star, whose every message carries an HMAC keyed by the launch's nonce (a 0600 file in the temp
        directory): only processes of the launching user can produce them."""
        import pickle
        if not self.star:
            return [obj]
        return [pickle.loads(p) for p in self.star.exchange(pickle.dumps(obj, protocol=4))]

    def gather_outputs(self, mine, total):
        """multiSMC's collection step (utils.py:178-186): rank r holds the outputs of the runs
        ``shard_islands(total, r, world)`` in run order; every rank gets all ``total`` of them in
        run order.  Numeric outputs -- floats or float arrays of one shape, the same on every
        rank -- cross the ranks through the evidence all-gather (RCCL; blocks padded to the
        largest share, since ncclAllGather wants equal counts); anything else is pickled over the
        host star.  Which of the two is one agreement round over the star, so that no rank enters
        a collective the others do not."""
        shares = [shard_islands(total, r, self.world)[1] for r in range(self.world)]
        if len(mine) != shares[self.rank]:
            raise ValueError("gather_outputs: rank %d holds %d outputs, its share is %d"
                             % (self.rank, len(mine), shares[self.rank]))
        shape, scalar = None, False
        try:
            arrs = [np.asarray(v) for v in mine]
            if all(a.dtype.kind == "f" and a.dtype.itemsize == 8 for a in arrs) and len({a.shape for a in arrs}) <= 1:
                shape = arrs[0].shape if arrs else "any"
                scalar = all(np.ndim(v) == 0 and not isinstance(v, np.ndarray) for v in mine)
        except Exception:
            shape = None
        votes = self.allgather_obj((shape, scalar))
        shapes = {v[0] for v in votes if v[0] != "any"}
        if None not in shapes and len(shapes) == 1:
            shp = shapes.pop()
            per = int(np.prod(shp)) if shp else 1
            block = np.full(max(shares) * per, np.nan)
            if mine:
                block[:len(mine) * per] = np.stack([np.asarray(v, dtype=np.float64).reshape(per) for v in mine]).ravel()
            allv = self.gather_evidence(block).reshape(self.world, max(shares), per)
            rows = [allv[r, i] for r in range(self.world) for i in range(shares[r])]
            if all(v[1] for v in votes):
                return [float(v[0]) for v in rows]
            return [v.reshape(shp).copy() for v in rows]
        return [o for part in self.allgather_obj(list(mine)) for o in part]

    # ---- the collective of the path ------------------------------------------
    def gather_evidence(self, local_logLt):
        """All ranks' per-island log-evidences, concatenated in rank order.
        Every rank must contribute the same number of islands."""
        local = np.ascontiguousarray(local_logLt, dtype=np.float64)
        if self.comm is not None:
            send = DeviceArray.from_numpy(local)
            recv = DeviceArray((self.world * local.size,))
            check(lib().smc_comm_allgather_f64(self.comm, send.ptr, local.size, recv.ptr))
            return recv.get()
        if self.star is None:
            return local.copy()
        parts = self.star.exchange(local.tobytes())
        return np.concatenate([np.frombuffer(p, dtype=np.float64) for p in parts])

    def migrate_islands(self, pf, src_global):
        """Global theta-level resampling across GPUs (smc_samplers.py:319-361 on a sharded
        population): global island slot g continues from the state of global island
        ``src_global[g]`` -- the array is the same on every rank, every rank holds
        ``pf.n_islands`` consecutive slots (rank r: r M .. r M + M - 1).  Whole island states
        travel packed (smc_filter_pack_islands) through ONE all-to-all of byte blocks over RCCL
        (ncclSend / ncclRecv pairs, smc_comm_alltoallv), or over the host star when the group
        has no device collective.  Philox streams stay tied to the slot, as with
        ``SMC.permute_islands``: the result equals the single-process permutation."""
        M, W, r = pf.n_islands, self.world, self.rank
        src = np.ascontiguousarray(src_global, dtype=np.int64)
        if src.shape != (M * W,) or src.min() < 0 or src.max() >= M * W:
            raise ValueError("migrate_islands: one source per global island slot")
        nb = _lib.c_i64()
        check(lib().smc_filter_island_bytes(pf._f, ctypes.byref(nb)))
        nb = int(nb.value)
        # what I send to rank p: the sources (mine) of p's slots, in slot order; what I receive from
        # p: my slots whose source lives on p, in slot order -- both sides enumerate the same pairs
        send_idx, recv_idx = [], []
        for p in range(W):
            sp = src[p * M:(p + 1) * M]
            send_idx.append((sp[sp // M == r] - r * M).astype(np.int64))
            mine = src[r * M:(r + 1) * M]
            recv_idx.append(np.nonzero(mine // M == p)[0].astype(np.int64))
        sc = np.array([len(v) * nb for v in send_idx], dtype=np.int64)
        rc = np.array([len(v) * nb for v in recv_idx], dtype=np.int64)
        sd = np.concatenate([[0], np.cumsum(sc)[:-1]]).astype(np.int64)
        rd = np.concatenate([[0], np.cumsum(rc)[:-1]]).astype(np.int64)
        s_all = np.ascontiguousarray(np.concatenate(send_idx)) if sc.sum() else np.zeros(0, dtype=np.int64)
        r_all = np.ascontiguousarray(np.concatenate(recv_idx)) if rc.sum() else np.zeros(0, dtype=np.int64)
        assert len(r_all) == M
        sbuf = DeviceArray((max(1, int(sc.sum()) // 8),), dtype=np.int64)
        rbuf = DeviceArray((max(1, int(rc.sum()) // 8),), dtype=np.int64)
        P64 = _lib.P(_lib.c_i64)
        if len(s_all):
            check(lib().smc_filter_pack_islands(pf._f, s_all.ctypes.data_as(P64), len(s_all), sbuf.ptr))
        if self.comm is not None:
            check(lib().smc_comm_alltoallv(self.comm, sbuf.ptr, sc.ctypes.data_as(P64), sd.ctypes.data_as(P64),
                                           rbuf.ptr, rc.ctypes.data_as(P64), rd.ctypes.data_as(P64)))
        else:       # host path: everybody's send buffer to everybody, each picks its blocks
            mine = sbuf.get().tobytes()[:int(sc.sum())]
            hdr = np.concatenate([sc, sd]).tobytes()
            parts = self.star.exchange(hdr + mine) if self.star else [hdr + mine]
            out = bytearray(int(rc.sum()))
            for p, blob in enumerate(parts):
                h = np.frombuffer(blob[:16 * W], dtype=np.int64)
                cnt, dsp = int(h[r]), int(h[W + r])
                assert cnt == rc[p]
                out[int(rd[p]):int(rd[p]) + cnt] = blob[16 * W + dsp:16 * W + dsp + cnt]
            host = np.frombuffer(bytes(out) + b"\0" * (-len(out) % 8), dtype=np.int64)
            rbuf = DeviceArray.from_numpy(host if host.size else np.zeros(1, dtype=np.int64))
        check(lib().smc_filter_unpack_islands(pf._f, r_all.ctypes.data_as(P64), M, rbuf.ptr))
        pf._invalidate()

    def move_islands(self, src_pf, dst_pf, dst_slots, src_slots):
        """Whole filters from one sharded batch into another: global slot ``dst_slots[i]`` of ``dst_pf`` (every rank
        holds ``dst_pf.n_islands`` consecutive slots) continues from global slot ``src_slots[i]`` of ``src_pf`` (likewise
        sharded, possibly with another count per rank).  The two arrays are the same on every rank.  Packed island
        states, one all-to-all (RCCL send / recv pairs, or the host star), as ``migrate_islands`` -- of which this is the
        two-batch form: the chains of a waste-free move start from resampled members of the population and their states
        become the next population (smc_samplers.py:669-684).  A fresh ``dst_pf`` is fast-forwarded to ``src_pf``'s time."""
        dst = np.ascontiguousarray(dst_slots, dtype=np.int64)
        src = np.ascontiguousarray(src_slots, dtype=np.int64)
        Ms, Md, W, r = src_pf.n_islands, dst_pf.n_islands, self.world, self.rank
        if dst.shape != src.shape or (len(dst) and (dst.min() < 0 or dst.max() >= Md * W or src.min() < 0 or src.max() >= Ms * W)):
            raise ValueError("move_islands: one source slot per destination slot, inside the two batches")
        if dst_pf._n == 0 and src_pf._n > 0:
            check(lib().smc_filter_fast_forward(dst_pf._f, src_pf._n))
            dst_pf.t = dst_pf._n = src_pf._n
        if dst_pf._n != src_pf._n:
            raise ValueError("move_islands: the two batches are at different time steps")
        nb = _lib.c_i64()
        check(lib().smc_filter_island_bytes(src_pf._f, ctypes.byref(nb)))
        nb2 = _lib.c_i64()
        check(lib().smc_filter_island_bytes(dst_pf._f, ctypes.byref(nb2)))
        if nb.value != nb2.value:
            raise ValueError("move_islands: the two batches hold filters of different shapes")
        nb = int(nb.value)
        send_idx, recv_idx = [], []
        for p in range(W):
            sel = (dst // Md == p) & (src // Ms == r)          # what I send to p: my sources of p's slots, in pair order
            send_idx.append((src[sel] - r * Ms).astype(np.int64))
            sel = (dst // Md == r) & (src // Ms == p)          # what I receive from p: my slots whose source lives on p
            recv_idx.append((dst[sel] - r * Md).astype(np.int64))
        sc = np.array([len(v) * nb for v in send_idx], dtype=np.int64)
        rc = np.array([len(v) * nb for v in recv_idx], dtype=np.int64)
        sd = np.concatenate([[0], np.cumsum(sc)[:-1]]).astype(np.int64)
        rd = np.concatenate([[0], np.cumsum(rc)[:-1]]).astype(np.int64)
        s_all = np.ascontiguousarray(np.concatenate(send_idx)) if sc.sum() else np.zeros(0, dtype=np.int64)
        r_all = np.ascontiguousarray(np.concatenate(recv_idx)) if rc.sum() else np.zeros(0, dtype=np.int64)
        sbuf = DeviceArray((max(1, int(sc.sum()) // 8),), dtype=np.int64)
        rbuf = DeviceArray((max(1, int(rc.sum()) // 8),), dtype=np.int64)
        P64 = _lib.P(_lib.c_i64)
        if len(s_all):
            check(lib().smc_filter_pack_islands(src_pf._f, s_all.ctypes.data_as(P64), len(s_all), sbuf.ptr))
        if self.comm is not None:
            check(lib().smc_comm_alltoallv(self.comm, sbuf.ptr, sc.ctypes.data_as(P64), sd.ctypes.data_as(P64),
                                           rbuf.ptr, rc.ctypes.data_as(P64), rd.ctypes.data_as(P64)))
        else:
            mine = sbuf.get().tobytes()[:int(sc.sum())]
            hdr = np.concatenate([sc, sd]).tobytes()
            parts = self.star.exchange(hdr + mine) if self.star else [hdr + mine]
            out = bytearray(int(rc.sum()))
            for p, blob in enumerate(parts):
                h = np.frombuffer(blob[:16 * W], dtype=np.int64)
                cnt, dsp = int(h[r]), int(h[W + r])
                assert cnt == rc[p]
                out[int(rd[p]):int(rd[p]) + cnt] = blob[16 * W + dsp:16 * W + dsp + cnt]
            host = np.frombuffer(bytes(out) + b"\0" * (-len(out) % 8), dtype=np.int64)
            rbuf = DeviceArray.from_numpy(host if host.size else np.zeros(1, dtype=np.int64))
        if len(r_all):
            check(lib().smc_filter_unpack_islands(dst_pf._f, r_all.ctypes.data_as(P64), len(r_all), rbuf.ptr))
        dst_pf._invalidate()

    @property
    def evidence_path(self):
        if self.comm is not None:
            return "rccl"
        return "none" if self.world == 1 else "host-fallback: %s" % self.fallback_reason

    def close(self):
        if self.comm is not None:
            lib().smc_comm_destroy(self.comm)
            self.comm = None
        if self.star is not None:
            self.star.close()
            self.star = None


def log_mean_exp_host(v):
    """Combine island evidences: log of the mean of exp(v) (the SMC^2 /
    multi-run estimator of the evidence; resampling.py:291-317 on M values)."""
    v = np.asarray(v, dtype=np.float64)
    m = v.max()
    return float(m + np.log(np.mean(np.exp(v - m))))
