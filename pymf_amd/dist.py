"""One-process-per-GPU world description and the rendezvous plumbing.

The data path never touches this module's transport: ranks exchange ONE
128-byte RCCL unique id at start-up, the row counts of their blocks, rank 0's
NumPy RNG state for the lazy init_w / init_h draws, bench barriers and a
max-reduce of timings; everything per iteration is an ncclAllReduce inside
libpymf_hip.  The transport is a few lines of TCP (standard library only, no
torch): rank 0 listens on MASTER_ADDR at the first free port of
[PYMF_DIST_PORT or MASTER_PORT + 1, +32), the other ranks connect to it, and
every collective is a gather to rank 0 followed by a broadcast (a star: the
payloads are a few hundred bytes).  The launch contract
(`python -m torch.distributed.run ...`) provides RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR / MASTER_PORT; nothing else of torch is used.

Who may join: the listener binds to the interface MASTER_ADDR resolves to (loopback for a one-node
launch) and admits a connection only after a challenge-response on a shared secret -- rank 0 sends a
random nonce, the peer answers HMAC-SHA256(key, nonce | rank).  The key comes from PYMF_DIST_SECRET;
a loopback rendezvous without it derives one from the launcher's env (every local user could do the
same: the protection there is the loopback bind), a NON-loopback rendezvous without it is refused.
Frames are length-prefixed and capped (PYMF_DIST_MAX_FRAME, default 1 GiB).  This is job plumbing for
a trusted cluster network, not a hardened service: anyone holding the secret can feed the sums.
"""
import hashlib
import hmac
import json
import os
import socket
import struct
import sys
import threading
import time

import numpy as np

_MAGIC = b"PYMFAMD2"
_PORT_SPAN = 32
_MAX_FRAME = int(os.environ.get("PYMF_DIST_MAX_FRAME", str(1 << 30)))
_HANDSHAKE_TIMEOUT = 2.0


class World(object):
    """rank/size + the row partition of the reference's m x n `data`."""

    def __init__(self, rank=0, size=1, local_rank=0, nccl_id=None):
        self.rank, self.size, self.local_rank, self.nccl_id = rank, size, local_rank, nccl_id

    def row_range(self, m_global):
        """Contiguous row block of this rank: rows are independent units given H."""
        return shard_rows(m_global, self.rank, self.size)


def shard_rows(m_global, rank, size):
    base, rem = divmod(int(m_global), int(size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


_WORLD = World()
_PEERS = None        # rank 0: {rank: socket};  other ranks: {0: socket}
_LISTENER = None


def world():
    return _WORLD


def transport():
    """How the per-iteration sums cross the ranks: "rccl" (default: ncclAllReduce over xGMI inside libpymf_hip, with the
    one-shot IPC all-reduce in front of it for payloads up to 256 KiB when it passes its self-test -- oneshot()), "host"
    (env PYMF_DIST_TRANSPORT=host: through this module's TCP star and pmf_set_host_allreduce -- plumbing checks where
    several ranks share one GPU; slow) or "ipc" (the host transport with the one-shot IPC all-reduce in front of it: how
    the one-shot kernel is exercised by two processes on ONE GPU, where RCCL refuses to form a communicator)."""
    t = os.environ.get("PYMF_DIST_TRANSPORT", "rccl").lower()
    if t not in ("rccl", "host", "ipc"):
        raise ValueError("PYMF_DIST_TRANSPORT must be 'rccl', 'host' or 'ipc'")
    return t


def oneshot():
    """Put the one-shot IPC all-reduce in front of RCCL?  Yes for 2..8 ranks of ONE node (env PYMF_DIST_ONESHOT=0 says
    no); it still has to reproduce RCCL's sums in Context.ipc_selftest() on every rank before it carries anything."""
    if os.environ.get("PYMF_DIST_ONESHOT", "1") == "0" or _WORLD is None or not (2 <= _WORLD.size <= 8):
        return False
    return int(os.environ.get("LOCAL_WORLD_SIZE", str(_WORLD.size)) or _WORLD.size) == _WORLD.size


class Watchdog(object):
    """Time-box for the start-up steps nobody can interrupt from inside: `ncclCommInitRank` (inside pmf_ctx_create),
    `hipIpcOpenMemHandle`, the transports' self-test.  They are collective -- a rank that waits for a peer which died, or for
    a fabric that never answers, would sit there for ever (RCCL) or for the in-kernel polling limit -- so a rank that is
    still inside after `seconds` (env PYMF_DIST_INIT_TIMEOUT, default 240) says so on stderr and ends its PROCESS with
    status 70: the launcher (torch.distributed.run, or bench.py's own) then stops the other ranks and reports a failed
    job instead of hanging.  ctypes releases the GIL during the C calls, so the timer thread does get to run."""

    EXIT_STATUS = 70
    DEFAULT_SECONDS = 240.0
    _active = 0            # time-boxes currently armed in this process: a nested one (setup_collectives inside make_context) rides on the outer

    @classmethod
    def default_seconds(cls):
        """PYMF_DIST_INIT_TIMEOUT in seconds; unset, empty or not a number -> 240 (0 or negative: no time-box)."""
        raw = os.environ.get("PYMF_DIST_INIT_TIMEOUT", "").strip()
        if not raw:
            return cls.DEFAULT_SECONDS
        try:
            return float(raw)
        except ValueError:
            sys.stderr.write("pymf_amd.dist: PYMF_DIST_INIT_TIMEOUT=%r is not a number of seconds; using %.0f\n" % (raw, cls.DEFAULT_SECONDS))
            return cls.DEFAULT_SECONDS

    def __init__(self, what, seconds=None):
        self.what = what
        self.explicit = seconds is not None
        self.seconds = float(seconds) if seconds is not None else self.default_seconds()
        self._t = None
        self._armed = False

    def _fire(self):
        try:
            sys.stderr.write("pymf_amd.dist: rank %d of %d is still in '%s' after %.0f s (PYMF_DIST_INIT_TIMEOUT); a peer rank died or "
                             "the transport does not come up -- giving up instead of hanging\n"
                             % (_WORLD.rank if _WORLD else 0, _WORLD.size if _WORLD else 1, self.what, self.seconds))
            for f in (sys.stdout, sys.stderr):     # what the process has buffered so far is not lost with it
                try:
                    f.flush()
                except Exception:
                    pass
        finally:
            os._exit(self.EXIT_STATUS)

    def __enter__(self):
        # one timer per start-up: a default-length box inside an armed one would only be a second, independent deadline
        # for the same hang (an explicit `seconds` -- the tests' short boxes -- is always honoured)
        if self.seconds > 0 and (self.explicit or Watchdog._active == 0):
            self._t = threading.Timer(self.seconds, self._fire)
            self._t.daemon = True
            self._t.start()
            Watchdog._active += 1
            self._armed = True
        return self

    def __exit__(self, *exc):
        if self._t is not None:
            self._t.cancel()
        if self._armed:
            Watchdog._active -= 1
            self._armed = False
        return False


LAST_SETUP = {}      # what setup_collectives() decided on this rank (bench.py prints it: config.collective_setup)


def same_node():
    """True iff every rank runs on the same host (one all-gather of hostname + boot id): HIP IPC handles mean nothing on
    another node, and LOCAL_WORLD_SIZE is not set by every launcher (srun, mpirun)."""
    try:
        with open("/proc/sys/kernel/random/boot_id") as fh:
            boot = fh.read().strip()
    except OSError:
        boot = ""
    me = (socket.gethostname() + "|" + boot).encode()
    return all(p == me for p in allgather_bytes(me, tag="node"))


def make_context(algo, m_local, n, k, share_gpu=False):
    """The device context of this rank, with the transports behind its cross-rank sums set up: RCCL communicator (inside
    pmf_ctx_create) unless the transport is "host" / "ipc" (ranks that cannot form one, e.g. sharing a GPU), then
    setup_collectives().  Collective over all ranks; time-boxed (Watchdog)."""
    from . import _lib
    w = _WORLD
    if w.size == 1:
        return _lib.Context(algo, m_local, n, k, device=w.local_rank)
    with Watchdog("context creation (ncclCommInitRank) + transport set-up"):
        if transport() in ("host", "ipc"):
            dev = 0 if share_gpu else w.local_rank % max(1, _lib.device_count())
            ctx = _lib.Context(algo, m_local, n, k, device=dev)
        else:
            ctx = _lib.Context(algo, m_local, n, k, device=w.local_rank, rank=w.rank, nranks=w.size, nccl_id=w.nccl_id)
        setup_collectives(ctx)
    return ctx


def setup_collectives(ctx):
    """After a multi-rank Context exists on every rank: the transports behind its cross-rank sums.  Collective (every
    rank calls it at the same point).  Returns Context.collective_name."""
    w, t = _WORLD, transport()
    LAST_SETUP.clear()
    LAST_SETUP.update(transport=t, ranks=w.size if w else 1, oneshot="not attempted")
    if w is None or w.size == 1:
        return getattr(ctx, "collective_name", "none")
    with Watchdog("transport set-up (IPC export / import / self-test)"):
        return _setup_collectives(ctx, w, t)


def _setup_collectives(ctx, w, t):
    if t in ("host", "ipc"):
        ctx.set_host_allreduce(allreduce_sum_array)
    want = t == "ipc" or (t == "rccl" and oneshot())
    if want and not same_node():
        want = False
        LAST_SETUP["oneshot"] = "not attempted: the ranks are not all on one node"
    if want:
        # Every rank makes the SAME sequence of exchanges here whatever fails locally -- a rank that skipped one would leave
        # the others waiting: export (local) -> all-gather of the handles (an empty one = "could not") -> import (local) ->
        # vote -> self-test against the other transport (all ranks, all rounds) -> vote.
        handle, stage = b"", "export"
        try:
            handle = ctx.ipc_export(w.rank, w.size)        # (also allocates the self-test's buffers: nothing below allocates)
        except Exception:
            handle = b""
        parts = allgather_bytes(handle, tag="ipc-hdl")
        ok = len(parts) == w.size and all(len(p) == len(parts[0]) and len(p) > 0 for p in parts)
        if ok:
            stage = "import"
            try:
                ctx.ipc_import(parts)
            except Exception:
                ok = False
        ok = allreduce_max(0.0 if ok else 1.0, tag="ipc-map") == 0.0
        if ok:
            stage = "self-test"
            try:
                ok = bool(ctx.ipc_selftest())
            except Exception:
                ok = False
            ok = allreduce_max(0.0 if ok else 1.0, tag="ipc-test") == 0.0
        if not ok:
            try:
                ctx.set_option("oneshot_allreduce", 0)
            except Exception:
                pass
        other = "rccl" if t == "rccl" else "host"
        LAST_SETUP["oneshot"] = "passed" if ok else "failed at %s -> %s" % (stage, other)
        if ok and os.environ.get("PYMF_DIST_FOLD", "1") == "0":
            # A/B knob: the per-iteration sum as a k_ipc_allreduce launch of its own instead of folded into the slab-reduce
            # and H-step launches (bit-identical either way)
            ctx.set_option("fold_exchange", 0)
        LAST_SETUP["fold_exchange"] = bool(ok) and os.environ.get("PYMF_DIST_FOLD", "1") != "0"
    return getattr(ctx, "collective_name", "")


# ---- framing ---------------------------------------------------------------------------------
def _send(sock, payload):
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(n - len(buf), 1 << 20))
        if not chunk:
            raise ConnectionError("pymf_amd.dist: peer closed the connection")
        buf += chunk
    return bytes(buf)


def _recv(sock):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    if n > _MAX_FRAME:
        raise ConnectionError("pymf_amd.dist: a %d-byte frame exceeds PYMF_DIST_MAX_FRAME (%d)" % (n, _MAX_FRAME))
    return _recv_exact(sock, n)


def _is_loopback(ip):
    return ip == "localhost" or ip.startswith("127.") or ip == "::1"


def _key(addr, port, size, bind_ip):
    """HMAC key of the rendezvous: PYMF_DIST_SECRET, or -- loopback only -- a value every rank of this
    launch can derive from its env."""
    secret = os.environ.get("PYMF_DIST_SECRET", "")
    if secret:
        return hashlib.sha256(b"pymf_amd.dist|" + secret.encode()).digest()
    if not _is_loopback(bind_ip):
        raise RuntimeError("pymf_amd.dist: MASTER_ADDR=%s is not a loopback address; set PYMF_DIST_SECRET (the same random "
                           "value on every rank) so that only this job's ranks can join the rendezvous" % addr)
    s = "%s|%s|%d|%s" % (addr, port, size, os.environ.get("TORCHELASTIC_RUN_ID", ""))
    return hashlib.sha256(s.encode()).digest()


def _mac(key, *parts):
    return hmac.new(key, b"|".join(parts), hashlib.sha256).digest()


# ---- rendezvous ------------------------------------------------------------------------------
def _serve(bind_ip, base_port, size, key, timeout):
    global _LISTENER
    lst = None
    for port in range(base_port, base_port + _PORT_SPAN):
        s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        try:
            s.bind((bind_ip, port))                  # the interface MASTER_ADDR names, never all of them
            s.listen(size + 8)
            lst = s
            break
        except OSError:
            s.close()
    if lst is None:
        raise RuntimeError("pymf_amd.dist: no free port on %s in [%d, %d); set PYMF_DIST_PORT" %
                           (bind_ip, base_port, base_port + _PORT_SPAN))
    _LISTENER = lst
    peers = {}
    deadline = time.time() + timeout
    while len(peers) < size - 1:
        lst.settimeout(max(0.1, deadline - time.time()))
        try:
            conn, _ = lst.accept()
        except socket.timeout:
            raise RuntimeError("pymf_amd.dist: only %d of %d ranks joined within %.0f s" %
                               (len(peers) + 1, size, timeout))
        try:
            conn.settimeout(_HANDSHAKE_TIMEOUT)      # a stranger can hold the accept loop this long at most
            nonce = os.urandom(16)
            conn.sendall(_MAGIC + nonce)
            hello = _recv_exact(conn, len(_MAGIC) + 4 + 32)
            rbytes = hello[len(_MAGIC):len(_MAGIC) + 4]
            (r,) = struct.unpack("<i", rbytes)
            good = hello[:len(_MAGIC)] == _MAGIC and hmac.compare_digest(hello[-32:], _mac(key, b"join", nonce, rbytes))
            if not good or r < 1 or r >= size or r in peers:
                conn.close()                         # not one of this job's ranks
                continue
            conn.sendall(_MAGIC + _mac(key, b"ack", nonce, rbytes))
            conn.settimeout(None)
            conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            peers[r] = conn
        except (OSError, ConnectionError, struct.error):
            conn.close()
    return peers


def _join(addr, base_port, rank, key, timeout):
    deadline = time.time() + timeout
    rbytes = struct.pack("<i", rank)
    while time.time() < deadline:
        for port in range(base_port, base_port + _PORT_SPAN):
            try:
                s = socket.create_connection((addr, port), timeout=2.0)
            except OSError:
                continue
            try:
                s.settimeout(5.0)
                greet = _recv_exact(s, len(_MAGIC) + 16)
                if greet[:len(_MAGIC)] == _MAGIC:
                    nonce = greet[len(_MAGIC):]
                    s.sendall(_MAGIC + rbytes + _mac(key, b"join", nonce, rbytes))
                    ack = _recv_exact(s, len(_MAGIC) + 32)
                    if ack[:len(_MAGIC)] == _MAGIC and hmac.compare_digest(ack[len(_MAGIC):], _mac(key, b"ack", nonce, rbytes)):
                        s.settimeout(None)
                        s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        return {0: s}
            except (OSError, ConnectionError):
                pass
            s.close()
        time.sleep(0.05)
    raise RuntimeError("pymf_amd.dist: rank %d could not reach rank 0 at %s:[%d,%d) within %.0f s" %
                       (rank, addr, base_port, base_port + _PORT_SPAN, timeout))


def init_from_env(make_nccl_id=None, timeout=None):
    """Read the launcher's env, connect the ranks, hand rank 0's RCCL unique id to every rank."""
    global _WORLD, _PEERS, _SEQ
    size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if size == 1:
        _WORLD = World(0, 1, local_rank, None)
        return _WORLD
    if _PEERS is not None:
        return _WORLD
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    mport = os.environ.get("MASTER_PORT", "29500")
    base = int(os.environ.get("PYMF_DIST_PORT", "0") or 0) or int(mport) + 1
    timeout = float(timeout if timeout is not None else os.environ.get("PYMF_DIST_TIMEOUT", "300"))
    try:
        bind_ip = "127.0.0.1" if addr == "localhost" else socket.gethostbyname(addr)
    except OSError:
        raise RuntimeError("pymf_amd.dist: MASTER_ADDR=%s does not resolve" % addr)
    local_size = int(os.environ.get("LOCAL_WORLD_SIZE", str(size)) or size)
    if size > local_size and _is_loopback(bind_ip):
        # a multi-node launch whose MASTER_ADDR resolves to loopback ON THIS HOST (the Debian / Ubuntu /etc/hosts
        # default for the host's own name): rank 0 would listen where no other node can reach it and the job would
        # die after the timeout with "only 1 of N ranks joined" -- fail at once instead
        raise RuntimeError("pymf_amd.dist: WORLD_SIZE=%d > LOCAL_WORLD_SIZE=%d (multi-node) but MASTER_ADDR=%s resolves to "
                           "the loopback address %s here; pass an address the other nodes can reach (IPv4)"
                           % (size, local_size, addr, bind_ip))
    key = _key(addr, mport, size, bind_ip)
    _WORLD = World(rank, size, local_rank, None)
    _SEQ = 0
    _PEERS = _serve(bind_ip, base, size, key, timeout) if rank == 0 else _join(bind_ip, base, rank, key, timeout)
    if make_nccl_id is None and transport() in ("host", "ipc"):
        return _WORLD                                  # no RCCL communicator will be created
    if make_nccl_id is None:
        from . import _lib
        make_nccl_id = _lib.nccl_unique_id
    ident = broadcast_bytes(bytes(make_nccl_id()) if rank == 0 else b"", tag="ncclid")
    _WORLD.nccl_id = ident
    return _WORLD


# ---- collectives (star through rank 0; every rank must make the same calls in the same order) --
_SEQ = 0             # collectives made so far on this rank (every frame carries it, with the operation's tag)


class CollectiveMismatch(RuntimeError):
    """The ranks made DIFFERENT collective calls at the same point (e.g. one rank alone called something collective):
    raised on every rank instead of pairing unrelated payloads or waiting for ever."""


def _star(payload, tag, reply_for):
    """One round trip through rank 0, the form of every collective here.  Every frame carries (sequence number, tag): rank 0
    checks that all ranks are making the SAME call and otherwise fails the collective on every rank (CollectiveMismatch).
    Rank 0 gets every rank's payload (`parts`, in rank order) and answers rank r with reply_for(parts, r); returns
    (parts on rank 0 / None elsewhere, this rank's reply)."""
    global _SEQ
    w = _WORLD
    _SEQ += 1
    head = struct.pack("<Q8s", _SEQ, tag.encode()[:8])
    if w.rank == 0:
        frames = [_recv(_PEERS[r]) for r in range(1, w.size)]
        bad = [(r + 1, f[:16]) for r, f in enumerate(frames) if f[:16] != head]
        if bad:
            r, h = bad[0]
            try:
                q, tg = struct.unpack("<Q8s", h)
                theirs = "call %d '%s'" % (q, tg.rstrip(b"\0").decode("utf-8", "replace"))
            except struct.error:
                theirs = "a malformed frame"
            msg = ("pymf_amd.dist: the ranks disagree about the collective at hand: rank 0 is in call %d '%s', rank %d in %s "
                   "-- a collective entry point (factorize, update_w / update_h, frobenius_norm, update_s, the lazy init) was "
                   "called on some ranks only" % (_SEQ, tag, r, theirs))
            for q in range(1, w.size):
                _send(_PEERS[q], b"\x01" + msg.encode())
            raise CollectiveMismatch(msg)
        parts = [bytes(payload)] + [f[16:] for f in frames]
        for r in range(1, w.size):
            _send(_PEERS[r], b"\x00" + reply_for(parts, r))
        return parts, reply_for(parts, 0)
    _send(_PEERS[0], head + bytes(payload))
    blob = _recv(_PEERS[0])
    if blob[:1] != b"\x00":
        raise CollectiveMismatch(blob[1:].decode("utf-8", "replace"))
    return None, blob[1:]


def allgather_bytes(payload, tag=""):
    """list of every rank's payload, in rank order, on every rank."""
    w = _WORLD
    if w.size == 1:
        return [bytes(payload)]
    cache = {}

    def everything(parts, r):
        if "b" not in cache:
            cache["b"] = b"".join(struct.pack("<Q", len(p)) + p for p in parts)
        return cache["b"]
    parts, blob = _star(payload, tag, everything)
    if parts is not None:
        return parts
    parts, off = [], 0
    while off < len(blob):
        (n,) = struct.unpack_from("<Q", blob, off)
        parts.append(blob[off + 8:off + 8 + n])
        off += 8 + n
    return parts


def gather_bytes(payload, tag="gather"):
    """Every rank's payload on rank 0 (list in rank order); None on the other ranks -- a W block per rank need not travel to
    all of them."""
    if _WORLD.size == 1:
        return [bytes(payload)]
    parts, _ = _star(payload, tag, lambda parts, r: b"")
    return parts


def scatter_bytes(parts, tag="scatter"):
    """Rank 0 passes one payload per rank (`parts`, rank order; ignored elsewhere); rank r gets parts[r]."""
    w = _WORLD
    if w.size == 1:
        return bytes(parts[0])
    if w.rank == 0 and len(parts) != w.size:
        raise ValueError("scatter_bytes: %d payloads for %d ranks" % (len(parts), w.size))
    mine = parts if w.rank == 0 else None
    _, got = _star(b"", tag, lambda _parts, r: bytes(mine[r]))
    return got


def broadcast_bytes(payload, src=0, tag="bcast"):
    return allgather_bytes(payload if _WORLD.rank == src else b"", tag=tag)[src]


def barrier():
    if _WORLD.size > 1:
        allgather_bytes(b"", tag="barrier")


def allreduce_max(x, tag="max"):
    if _WORLD.size == 1:
        return float(x)
    return max(struct.unpack("<d", p)[0] for p in allgather_bytes(struct.pack("<d", float(x)), tag=tag))


def allgather_int(x):
    return [struct.unpack("<q", p)[0] for p in allgather_bytes(struct.pack("<q", int(x)), tag="ints")]


def allgather_float(x):
    return [struct.unpack("<d", p)[0] for p in allgather_bytes(struct.pack("<d", float(x)), tag="floats")]


def allreduce_sum_array(a, tag=None):
    """Sum of a float array over the ranks, added in rank order (the same bits on every rank).  The default tag carries
    dtype and size, so that two ranks summing DIFFERENT things at the same point fail loudly instead of mixing payloads."""
    a = np.ascontiguousarray(a)
    if _WORLD.size == 1:
        return a
    parts = allgather_bytes(a.tobytes(), tag=tag or ("s%s%d" % (a.dtype.char, a.size))[:8])
    out = np.zeros(a.shape, dtype=a.dtype)
    for p in parts:
        out += np.frombuffer(p, dtype=a.dtype).reshape(a.shape)
    return out


def broadcast_array(a, src=0):
    """rank `src`'s array (any shape/dtype) on every rank."""
    if _WORLD.size == 1:
        return np.asarray(a)
    if _WORLD.rank == src:
        a = np.ascontiguousarray(a)
        head = json.dumps({"dtype": a.dtype.str, "shape": list(a.shape)}).encode()
        blob = struct.pack("<I", len(head)) + head + a.tobytes()
    else:
        blob = b""
    blob = broadcast_bytes(blob, src, tag="array")
    (hl,) = struct.unpack_from("<I", blob, 0)
    head = json.loads(blob[4:4 + hl].decode())
    return np.frombuffer(blob[4 + hl:], dtype=np.dtype(head["dtype"])).reshape(head["shape"]).copy()


def share_rng_state(src=0):
    """Give every rank rank `src`'s state of NumPy's global legacy stream (np.random.random, the
    generator the reference's init_w / init_h draw from, nmf.py:116-120): after this call every rank
    draws the same numbers, so each can draw the GLOBAL W0 and keep its own rows, and H0 is
    identical everywhere without moving either matrix."""
    if _WORLD.size == 1:
        return
    if _WORLD.rank == src:
        name, keys, pos, has_gauss, cached = np.random.get_state()
        head = json.dumps({"name": name, "pos": int(pos), "has_gauss": int(has_gauss),
                           "cached": float(cached)}).encode()
        blob = struct.pack("<I", len(head)) + head + np.asarray(keys, dtype=np.uint32).tobytes()
    else:
        blob = b""
    blob = broadcast_bytes(blob, src, tag="rng")
    (hl,) = struct.unpack_from("<I", blob, 0)
    head = json.loads(blob[4:4 + hl].decode())
    keys = np.frombuffer(blob[4 + hl:], dtype=np.uint32).copy()
    np.random.set_state((head["name"], keys, head["pos"], head["has_gauss"], head["cached"]))


def shutdown():
    global _WORLD, _PEERS, _LISTENER, _SEQ
    if _PEERS is not None:
        try:
            barrier()                    # nobody closes while another rank still reads
        except Exception:
            pass
        for s in _PEERS.values():
            try:
                s.close()
            except OSError:
                pass
    if _LISTENER is not None:
        try:
            _LISTENER.close()
        except OSError:
            pass
    _PEERS, _LISTENER = None, None
    _WORLD = World()
    _SEQ = 0
