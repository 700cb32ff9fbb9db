"""Chain sharding across GPUs and the only two cross-shard exchanges of the path.

Chains are independent (src/metropolis.jl:303-307 maps mc_sweep! over chains with no shared
mutable state), so rank r of W owns a contiguous range of GLOBAL chain ids and sweeps need no
collective.  Information crosses chains in exactly two places, both tiny sums:
  * the callbacks (callback_energy particle_1d.jl:68-70, callback_acceptance metropolis.jl:319-321)
  * the GradientData `+` fold of the estimator (src/PolicyGuided/estimator.jl:113-129)
These become ONE all-reduce(sum, f64) of a few dozen bytes.  On GPUs it runs through the engine's own RCCL
communicator (amc_comm_init / amc_allreduce_sum: RCCL over xGMI, on a stream of the engine's own, no torch tensors involved);
a plain-socket key-value store beside the launcher's carries the 128-byte ncclUniqueId and the run's barriers (`StoreGroup`:
no process group, no torch in the worker, one RCCL instance in the process -- the system's).  A torch.distributed process group, if the script has one, is used the same way
("gloo" in CPU tests; "nccl" only to ship the unique id).  The Philox counter uses the global chain id, so results do
not depend on W (shard invariance is tested).
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import numpy as np


class SocketStore:
    """A key-value store over plain TCP sockets: what this path needs of the launcher's rendezvous (set / get / add / wait /
    delete), without torch.  Rank 0 serves it from a thread on MASTER_PORT + 1 (the launcher's own store sits on MASTER_PORT);
    every rank -- rank 0 included -- is a client.  `get` blocks on the server until the key exists.  A worker that uses this
    store never imports torch, so libamc.so binds the system's HIP runtime and RCCL (/opt/rocm) in every rank, as it does in
    a single process.

    On the wire: fixed binary frames (`_pack_request` / `_pack_reply`: an operation byte, a UTF-8 key, a value that is absent,
    raw bytes or an integer), at most MAX_FRAME bytes each -- nothing either side receives is ever unpickled or evaluated, and
    the server stores values as opaque bytes.  Both ends first prove that they belong to the same launch, by challenge and
    response: the server greets with a fresh nonce; the client answers with a nonce of its own and HMAC-SHA256(token, "client" |
    both nonces); the server checks it and answers HMAC(token, "server" | both nonces), which the client checks -- so a stale
    server of an earlier run, or a stranger who reaches the port, is neither served nor mistaken for the store, and nothing that
    is ever sent is a fixed function of the token.  The token: AMC_STORE_TOKEN, else what the launcher hands every worker of one
    launch (run id, restart count, world size, port -- and, under torch.distributed.run on one host, the launcher's pid).
    Where it listens: AMC_STORE_BIND if set; 127.0.0.1 when MASTER_ADDR is this host's loopback (one node is the whole machine
    for this path: nothing off the host can connect); every interface otherwise (ranks on other nodes must reach it) -- and
    then only with a secret: without AMC_STORE_TOKEN a store that strangers can reach refuses to start (the derived token is
    no secret: every part of it can be guessed)."""

    # Where the store may sit, relative to the port asked for: the first of these that rank 0 can bind.  Clients walk the same
    # list and know their server by its greeting, so a foreign service that happens to own MASTER_PORT + 1 neither stops the run
    # nor gets mistaken for the store.
    PORT_OFFSETS = (0, 100, 202, 1008, 2006)
    MAGIC = "amc-socket-store-3"
    NONCE_HEX = 32                       # characters of a nonce on the wire
    MAC_HEX = 64                         # ... of an HMAC-SHA256
    MAX_FRAME = 1 << 20                  # bytes: the store carries ids, flags and a few dozen doubles

    @staticmethod
    def launch_token(host: str, port: int) -> bytes:
        explicit = os.environ.get("AMC_STORE_TOKEN")
        if explicit:
            return explicit.encode()
        parts = [os.environ.get("TORCHELASTIC_RUN_ID", ""), os.environ.get("TORCHELASTIC_RESTART_COUNT", ""),
                 os.environ.get("WORLD_SIZE", ""), str(int(port))]
        if _is_loopback(host) and os.environ.get("TORCHELASTIC_RUN_ID"):
            parts.append(str(os.getppid()))      # the workers of one local launch share their launcher (ranks started by hand,
        return "/".join(parts).encode()          # from different shells, do not: no pid in their token)

    def __init__(self, host: str, port: int, is_master: bool, timeout_s: float = 600.0, token: Optional[bytes] = None):
        import errno
        import hashlib
        import socket
        import threading
        import time
        self._timeout = float(timeout_s)
        self._lock = threading.Lock()
        self._token = self.launch_token(host, port) if token is None else bytes(token)
        self._hello_head = f"{self.MAGIC} {int(port)} ".encode()          # + the connection's nonce + newline
        if is_master:
            self._data = {}
            self._cond = threading.Condition()
            bind_host = os.environ.get("AMC_STORE_BIND")
            if bind_host is None:
                bind_host = "127.0.0.1" if _is_loopback(host) else ""
            if (bind_host == "" or not _is_loopback(bind_host)) and not os.environ.get("AMC_STORE_TOKEN") and token is None:
                raise OSError(f"SocketStore: asked to listen on {bind_host or 'every interface'} (MASTER_ADDR={host}) without "
                              f"AMC_STORE_TOKEN: the token derived from the launch's environment is guessable, so a store that "
                              f"other hosts can reach needs a secret -- export AMC_STORE_TOKEN=<random string> on every rank, "
                              f"or use MASTER_ADDR=127.0.0.1 on one node")
            srv, last = None, None
            for off in self.PORT_OFFSETS:
                cand = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                cand.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                try:
                    cand.bind((bind_host, int(port) + off))
                    srv = cand
                    break
                except OSError as err:
                    cand.close()
                    if err.errno != errno.EADDRINUSE:     # not "taken": an address this host cannot listen on -- say so
                        raise OSError(f"SocketStore: cannot listen on {bind_host or '*'}:{int(port) + off} ({err}); "
                                      f"set AMC_STORE_BIND to an address of this host") from err
                    last = err                            # the port is taken: the next candidate
            if srv is None:
                raise OSError(f"SocketStore: the ports {[int(port) + o for o in self.PORT_OFFSETS]} on {bind_host or '*'} are all in use ({last})")
            srv.listen(256)
            srv.settimeout(0.2)
            self._srv = srv
            self._open = 0            # client connections open now
            # NOT a daemon: the serving process (rank 0) must outlive its clients' last reads -- the thread ends once the main
            # thread is done and every other client has hung up (or 30 s later)
            threading.Thread(target=self._serve, daemon=False).start()
        deadline = time.monotonic() + self._timeout
        self._sock = None
        while self._sock is None:                     # the server may come up after its clients
            for off in self.PORT_OFFSETS:
                try:
                    sk = socket.create_connection((host, int(port) + off), timeout=5.0)
                except OSError:
                    continue
                try:
                    sk.settimeout(3.0)
                    if self._handshake_as_client(sk):      # our store of THIS launch, not whoever else listens there
                        self._sock = sk
                        break
                except Exception:
                    pass
                sk.close()
            if self._sock is None:
                if time.monotonic() > deadline:
                    raise TimeoutError(
                        f"no SocketStore of this launch at {host}:{port} (+{list(self.PORT_OFFSETS)}) after {self._timeout:.0f} s: either rank 0 "
                        f"never served one, or one answers there that does not hold this rank's token.  The token is AMC_STORE_TOKEN if set "
                        f"({'set' if os.environ.get('AMC_STORE_TOKEN') else 'not set'} here), else TORCHELASTIC_RUN_ID / restart count / WORLD_SIZE / "
                        f"port (and the launcher's pid under torch.distributed.run): ranks started by hand must agree on all of them -- or "
                        f"export the same AMC_STORE_TOKEN on every rank")
                time.sleep(0.05)
        self._sock.settimeout(self._timeout)
        self._sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)

    def _mac(self, who: bytes, server_nonce: bytes, client_nonce: bytes) -> bytes:
        import hashlib
        import hmac
        return hmac.new(self._token, who + b"\0" + server_nonce + b"\0" + client_nonce, hashlib.sha256).hexdigest().encode()

    def _handshake_as_client(self, sk) -> bool:
        """Greeting with the server's nonce in; our nonce and HMAC out; the server's HMAC in.  True: this is our launch's store."""
        import hmac
        import secrets
        hello = _recv_exactly(sk, len(self._hello_head) + self.NONCE_HEX + 1)
        if hello is None or not hello.startswith(self._hello_head) or not hello.endswith(b"\n"):
            return False
        server_nonce = hello[len(self._hello_head):-1]
        client_nonce = secrets.token_hex(self.NONCE_HEX // 2).encode()
        sk.sendall(client_nonce + self._mac(b"client", server_nonce, client_nonce) + b"\n")
        proof = _recv_exactly(sk, self.MAC_HEX + 1)
        return proof is not None and hmac.compare_digest(proof, self._mac(b"server", server_nonce, client_nonce) + b"\n")

    # ---- server side (rank 0) ----
    def _serve(self) -> None:
        import socket
        import threading
        import time
        main_done_at = None
        while True:
            main_done = not threading.main_thread().is_alive()
            try:
                conn, _ = self._srv.accept()
                if main_done:                  # this launch is over: whoever connects now belongs to another one
                    conn.close()
                else:
                    with self._cond:
                        self._open += 1
                    threading.Thread(target=self._client, args=(conn,), daemon=True).start()
                continue
            except socket.timeout:
                pass
            except OSError:
                return
            if main_done:
                main_done_at = main_done_at or time.monotonic()
                with self._cond:
                    others_open = self._open - 1          # this process's own client connection stays open to the end
                if others_open <= 0 or time.monotonic() - main_done_at > 30.0:
                    self._srv.close()
                    return

    def _client(self, conn) -> None:
        import hmac
        import socket
        conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        try:
            import secrets
            conn.settimeout(10.0)
            server_nonce = secrets.token_hex(self.NONCE_HEX // 2).encode()
            conn.sendall(self._hello_head + server_nonce + b"\n")          # the challenge; nothing about the token yet
            answer = _recv_exactly(conn, self.NONCE_HEX + self.MAC_HEX + 1)
            if answer is None or not answer.endswith(b"\n"):
                return
            client_nonce, mac = answer[:self.NONCE_HEX], answer[self.NONCE_HEX:-1]
            if not hmac.compare_digest(mac, self._mac(b"client", server_nonce, client_nonce)):
                return                                # not a rank of this launch: nothing of it is parsed, nothing served
            conn.sendall(self._mac(b"server", server_nonce, client_nonce) + b"\n")        # ... and the proof that WE hold the token
            conn.settimeout(None)
            while True:
                req = _recv_frame(conn, self.MAX_FRAME)
                if req is None:
                    return
                op, key, val = _unpack_request(req)
                with self._cond:
                    if op == _OP_SET and isinstance(val, bytes):
                        self._data[key] = val
                        self._cond.notify_all()
                        rep = _pack_reply(True, 1)
                    elif op == _OP_ADD and isinstance(val, int):
                        cur = self._data.get(key, 0)
                        self._data[key] = (cur if isinstance(cur, int) else 0) + val
                        self._cond.notify_all()
                        rep = _pack_reply(True, self._data[key])
                    elif op == _OP_GET:
                        if not self._cond.wait_for(lambda: key in self._data, timeout=self._timeout):
                            rep = _pack_reply(False, f"timed out waiting for key {key!r}".encode())
                        else:
                            rep = _pack_reply(True, self._data[key])
                    elif op == _OP_DEL:
                        rep = _pack_reply(True, 1 if self._data.pop(key, None) is not None else 0)
                    elif op == _OP_LEN:
                        rep = _pack_reply(True, len(self._data))
                    else:
                        rep = _pack_reply(False, b"malformed request")
                _send_frame(conn, rep)
        except (OSError, ValueError):                 # a broken connection, a frame that does not parse: drop the client
            return
        finally:
            conn.close()
            with self._cond:
                self._open -= 1

    # ---- client side ----
    def _call(self, op: int, key: str, val=None):
        with self._lock:
            _send_frame(self._sock, _pack_request(op, key, val))
            rep = _recv_frame(self._sock, self.MAX_FRAME)
        if rep is None:
            raise ConnectionError("SocketStore: the server closed the connection")
        ok, value = _unpack_reply(rep)
        if not ok:
            raise TimeoutError(f"SocketStore: {value.decode(errors='replace') if isinstance(value, bytes) else value}")
        return value

    def set(self, key: str, value: bytes) -> None:
        self._call(_OP_SET, key, bytes(value))

    def get(self, key: str) -> bytes:
        return self._call(_OP_GET, key)

    def add(self, key: str, amount: int) -> int:
        return int(self._call(_OP_ADD, key, int(amount)))

    def wait(self, keys) -> None:
        for k in keys:
            self._call(_OP_GET, k)

    def delete_key(self, key: str) -> bool:
        return bool(self._call(_OP_DEL, key))

    def num_keys(self) -> int:
        return int(self._call(_OP_LEN, ""))


def _is_loopback(host: str) -> bool:
    return host in ("localhost", "::1", "") or host.startswith("127.")


# ---- the store's wire format: length-prefixed frames of plain fields (struct), nothing executable ----
_OP_SET, _OP_GET, _OP_ADD, _OP_DEL, _OP_LEN = 1, 2, 3, 4, 5
_V_NONE, _V_BYTES, _V_INT = 0, 1, 2


def _recv_exactly(sock, n: int):
    buf = b""
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            return None
        buf += chunk
    return buf


def _send_frame(sock, body: bytes) -> None:
    import struct
    sock.sendall(struct.pack("<I", len(body)) + body)


def _recv_frame(sock, limit: int):
    import struct
    head = _recv_exactly(sock, 4)
    if head is None:
        return None
    n = struct.unpack("<I", head)[0]
    if n > limit:
        raise ValueError(f"frame of {n} bytes exceeds the store's limit of {limit}")
    return _recv_exactly(sock, n)


def _pack_value(val) -> bytes:
    import struct
    if val is None:
        return bytes([_V_NONE])
    if isinstance(val, (bytes, bytearray)):
        return bytes([_V_BYTES]) + bytes(val)
    if isinstance(val, int):
        return bytes([_V_INT]) + struct.pack("<q", val)
    raise TypeError(f"the store carries bytes and integers, not {type(val).__name__}")


def _unpack_value(buf: bytes):
    import struct
    if not buf:
        raise ValueError("empty value field")
    if buf[0] == _V_NONE and len(buf) == 1:
        return None
    if buf[0] == _V_BYTES:
        return bytes(buf[1:])
    if buf[0] == _V_INT and len(buf) == 9:
        return struct.unpack("<q", buf[1:])[0]
    raise ValueError("malformed value field")


def _pack_request(op: int, key: str, val) -> bytes:
    import struct
    k = key.encode()
    return struct.pack("<BH", op, len(k)) + k + _pack_value(val)


def _unpack_request(buf: bytes):
    import struct
    if len(buf) < 3:
        raise ValueError("short request")
    op, nk = struct.unpack("<BH", buf[:3])
    if len(buf) < 3 + nk + 1:
        raise ValueError("short request")
    return op, buf[3:3 + nk].decode(), _unpack_value(buf[3 + nk:])


def _pack_reply(ok: bool, val) -> bytes:
    return bytes([1 if ok else 0]) + _pack_value(val)


def _unpack_reply(buf: bytes):
    if not buf:
        raise ValueError("empty reply")
    return buf[0] == 1, _unpack_value(buf[1:])


# ---- values the ranks exchange THROUGH the store (flags, reasons, the ncclUniqueId, a few doubles): a closed set of plain
# types in a tagged binary form -- a rank never unpickles what another process wrote ----
def _dumps(obj) -> bytes:
    import struct
    if obj is None:
        return b"N"
    if isinstance(obj, (bool, np.bool_)):
        return b"T" if obj else b"F"
    if isinstance(obj, (int, np.integer)):
        return b"i" + struct.pack("<q", int(obj))
    if isinstance(obj, (float, np.floating)):
        return b"d" + struct.pack("<d", float(obj))
    if isinstance(obj, str):
        raw = obj.encode()
        return b"s" + struct.pack("<I", len(raw)) + raw
    if isinstance(obj, (bytes, bytearray)):
        return b"b" + struct.pack("<I", len(obj)) + bytes(obj)
    if isinstance(obj, np.ndarray):
        arr = np.ascontiguousarray(obj)
        if arr.dtype.kind not in "fiub":
            raise TypeError(f"arrays of dtype {arr.dtype} do not travel over the store")
        dt = arr.dtype.str.encode()
        return (b"a" + struct.pack("<BB", len(dt), arr.ndim) + dt + struct.pack(f"<{arr.ndim}q", *arr.shape) + arr.tobytes())
    if isinstance(obj, (list, tuple)):
        return (b"l" if isinstance(obj, list) else b"t") + struct.pack("<I", len(obj)) + b"".join(_dumps(x) for x in obj)
    if isinstance(obj, dict):
        return b"m" + struct.pack("<I", len(obj)) + b"".join(_dumps(str(k)) + _dumps(v) for k, v in obj.items())
    raise TypeError(f"{type(obj).__name__} objects do not travel over the store")


def _loads(buf: bytes):
    obj, end = _load_at(memoryview(buf), 0)
    if end != len(buf):
        raise ValueError("trailing bytes after the value")
    return obj


def _load_at(buf, at: int):
    import struct
    tag = bytes(buf[at:at + 1])
    at += 1
    if tag == b"N":
        return None, at
    if tag in (b"T", b"F"):
        return tag == b"T", at
    if tag == b"i":
        return struct.unpack_from("<q", buf, at)[0], at + 8
    if tag == b"d":
        return struct.unpack_from("<d", buf, at)[0], at + 8
    if tag in (b"s", b"b"):
        n = struct.unpack_from("<I", buf, at)[0]
        raw = bytes(buf[at + 4:at + 4 + n])
        if len(raw) != n:
            raise ValueError("truncated value")
        return (raw.decode() if tag == b"s" else raw), at + 4 + n
    if tag == b"a":
        nd, ndim = struct.unpack_from("<BB", buf, at)
        dt = np.dtype(bytes(buf[at + 2:at + 2 + nd]).decode())
        if dt.kind not in "fiub":
            raise ValueError("array dtype not allowed")
        at += 2 + nd
        shape = struct.unpack_from(f"<{ndim}q", buf, at)
        at += 8 * ndim
        n = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
        raw = bytes(buf[at:at + n])
        if len(raw) != n or any(d < 0 for d in shape):
            raise ValueError("truncated array")
        return np.frombuffer(raw, dtype=dt).reshape(shape).copy(), at + n
    if tag in (b"l", b"t"):
        n = struct.unpack_from("<I", buf, at)[0]
        at += 4
        out = []
        for _ in range(n):
            x, at = _load_at(buf, at)
            out.append(x)
        return (out if tag == b"l" else tuple(out)), at
    if tag == b"m":
        n = struct.unpack_from("<I", buf, at)[0]
        at += 4
        out = {}
        for _ in range(n):
            k, at = _load_at(buf, at)
            v, at = _load_at(buf, at)
            out[k] = v
        return out, at
    raise ValueError(f"unknown value tag {tag!r}")


class StoreGroup:
    """The ranks of one launch, tied together by a key-value store only: barrier, all-gather of small Python objects
    (timings, the ncclUniqueId), and a deterministic host-side sum for engines without a communicator (CPU tests).  No
    init_process_group, hence no torch-side NCCL/RCCL communicator and no torch CUDA context.

    `python -m torch.distributed.run` hands every worker RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT.  By default the
    store is this module's own `SocketStore` on MASTER_PORT + 1 (rank 0 serves it): the worker does not import torch at
    all, so libamc.so binds the system's HIP runtime and RCCL exactly as in a single process.  `store="torch"` (or
    AMC_STORE=torch) uses the launcher's own TCPStore on MASTER_PORT instead (TORCHELASTIC_USE_AGENT_STORE=True; started by
    hand, rank 0 hosts it) -- torch then has to be imported first and its bundled ROCm is what libamc.so binds."""

    def __init__(self, rank: int, world_size: int, host: str, port: int, agent_store: bool, timeout_s: float = 600.0,
                 store: Optional[str] = None):
        self.rank, self.world_size = int(rank), int(world_size)
        kind = (store or os.environ.get("AMC_STORE", "socket")).lower()
        if kind == "torch":
            from datetime import timedelta
            from torch.distributed import TCPStore         # torch first, then libamc.so (one HIP runtime per process)
            self.store = TCPStore(host, int(port), self.world_size, is_master=(self.rank == 0 and not agent_store),
                                  timeout=timedelta(seconds=timeout_s), wait_for_workers=False)
        elif kind == "socket":
            self.store = SocketStore(host, int(port) + 1, is_master=(self.rank == 0), timeout_s=timeout_s)
        else:
            raise ValueError(f"unknown store kind {kind!r}: 'socket' or 'torch'")
        self.kind = kind
        self._n = 0
        self._old: List[str] = []        # rank 0: keys of finished rounds, deleted once every rank is known to be past them

    def _key(self, what: str) -> str:
        self._n += 1
        return f"amc/{what}/{self._n}"

    def _retire(self, keys: List[str], all_arrived: bool) -> None:
        """Rank 0 keeps the launcher's store from growing with the run (one round of keys per callback otherwise).  A round's
        keys may go once EVERY rank has finished it; rank 0 knows that of all earlier rounds when it completes a round in
        which it has seen every rank arrive (all-gather, barrier) -- a rank arrives at round n only after finishing n - 1."""
        if self.rank != 0:
            return
        if all_arrived:
            for k in self._old:
                try:
                    self.store.delete_key(k)
                except Exception:          # a store without delete_key: keep the keys, nothing else depends on it
                    pass
            self._old = []
        self._old.extend(keys)

    def barrier(self) -> None:
        key = self._key("barrier")
        if self.store.add(key, 1) == self.world_size:
            self.store.set(key + "/done", b"1")
        self.store.wait([key + "/done"])
        self._retire([key, key + "/done"], all_arrived=True)

    def allgather(self, obj) -> List:
        """[obj of rank 0, obj of rank 1, ...] on every rank (plain values: None, bool, int, float, str, bytes, numeric arrays,
        lists / tuples / dicts of those -- `_dumps`)."""
        key = self._key("gather")
        self.store.set(f"{key}/{self.rank}", _dumps(obj))
        out = [_loads(self.store.get(f"{key}/{r}")) for r in range(self.world_size)]
        self._retire([f"{key}/{r}" for r in range(self.world_size)], all_arrived=True)
        return out

    def broadcast(self, obj, src: int = 0):
        key = self._key("bcast")
        if self.rank == src:
            self.store.set(key, _dumps(obj))
        out = _loads(self.store.get(key))
        self._retire([key], all_arrived=False)       # the source does not learn here who has read the key
        return out

    def allreduce_sum(self, values: np.ndarray) -> np.ndarray:
        """Host-side sum in rank order (deterministic); for engines without a communicator of their own."""
        parts = self.allgather(np.ascontiguousarray(values, dtype=np.float64))
        total = parts[0].copy()
        for p in parts[1:]:
            total = total + p
        return total


_group: Optional[StoreGroup] = None


def init_store_group(rank: Optional[int] = None, world_size: Optional[int] = None, store: Optional[str] = None) -> StoreGroup:
    """Join the launch's ranks through a key-value store (environment of torch.distributed.run, or RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT set by hand): this module's socket store by default (no torch in the worker), the launcher's
    TCPStore with store="torch" / AMC_STORE=torch.  Call before building the Simulation, in place of
    torch.distributed.init_process_group."""
    global _group
    if _group is not None:
        return _group
    # RCCL between the ranks' processes: this pool's driver only supports dmabuf IPC (legacy mode: hipIpcGetMemHandle fails);
    # must be in the environment before the HIP runtime starts, i.e. before the first engine is created
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
    world_size = int(os.environ.get("WORLD_SIZE", "1")) if world_size is None else int(world_size)
    host = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(os.environ.get("MASTER_PORT", "29500"))
    agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "False") == "True"
    _group = StoreGroup(rank, world_size, host, port, agent, store=store)
    return _group


def group() -> Optional[StoreGroup]:
    return _group


def connect_engine(engine) -> bool:
    """Give the engine of this rank's shard an RCCL communicator over all ranks (amc_comm_init); the ncclUniqueId made on
    rank 0 travels over the store group or, failing that, over the script's torch.distributed process group.  A COLLECTIVE:
    every rank calls it, every rank consumes the same store rounds whatever fails where, and every rank gets the same
    answer -- True only when ALL engines are connected (their allreduce_sum / device-resident estimator then span the
    shards); otherwise every engine is left a single shard and the callers' sums go over the host."""
    if getattr(engine, "comm_connected", False):
        return True
    rank, size = world()
    capable = hasattr(engine, "comm_init") and hasattr(engine, "comm_unique_id")
    if _group is not None:
        bcast = _group.broadcast
        gather = _group.allgather
    else:
        import sys
        if "torch" not in sys.modules:
            return False
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_backend() != "nccl":
            return False

        def bcast(obj):
            box = [obj]
            dist.broadcast_object_list(box, src=0)
            return box[0]

        def gather(obj):
            out = [None] * size
            dist.all_gather_object(out, obj)
            return out
    # round 0: is every rank ready to enter ncclCommInitRank?  That call is itself a collective -- a rank whose engine cannot
    # do it, or whose librccl does not load, would leave the others blocked inside it --, so readiness is gathered first:
    # making a unique id loads the library and calls into it without touching any other rank (ids of ranks > 0 are dropped)
    uid, why_not = None, None
    if capable:
        try:
            uid = engine.comm_unique_id()
        except Exception as err:                       # librccl not loadable, ncclGetUniqueId failed
            why_not = str(err)
    else:
        why_not = "engine has no comm_init"
    ready = gather((uid is not None, why_not))
    if not all(f[0] for f in ready):
        engine.comm_connected = False
        if rank == 0:
            import sys
            reasons = "; ".join(f"rank {r}: {f[1]}" for r, f in enumerate(ready) if not f[0])
            print(f"[montecarlo_amd] no RCCL communicator over the shards ({reasons}): sums go over the host", file=sys.stderr)
        return False
    # round 1: rank 0's unique id
    record = bcast(("uid", uid) if rank == 0 else None)
    ok, why = False, None
    if record[0] == "uid" and capable:
        try:
            engine.comm_init(rank, size, record[1])
            ok = True
        except Exception as err:
            why = str(err)
    else:
        why = record[1] if record[0] == "error" else "engine has no comm_init"
    # round 2: all or none
    flags = gather((bool(ok), why))
    all_ok = all(f[0] for f in flags)
    if ok and not all_ok and hasattr(engine, "comm_destroy"):
        try:
            engine.comm_destroy()                      # connected here, not there: back to a single shard
        except Exception:
            pass
    engine.comm_connected = all_ok
    if not all_ok and rank == 0:
        import sys
        reasons = "; ".join(f"rank {r}: {f[1]}" for r, f in enumerate(flags) if not f[0])
        print(f"[montecarlo_amd] no RCCL communicator over the shards ({reasons}): sums go over the host", file=sys.stderr)
    return all_ok


def world() -> Tuple[int, int]:
    """(rank, world_size): of the store group if one was joined, else of the default torch process group, else (0, 1)."""
    if _group is not None:
        return _group.rank, _group.world_size
    # A process group can only exist if the caller has imported torch already; never import it from here:
    # torch bundles its own HIP runtime, and loading it AFTER libamc.so has bound the system one puts two
    # HIP runtimes in the process (the C-ABI RCCL path then fails).  Multi-GPU scripts import torch and call
    # init_process_group before building the Simulation, so torch is always first in that case.
    import sys
    if "torch" not in sys.modules:
        return 0, 1
    try:
        import torch.distributed as dist
    except Exception:
        return 0, 1
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n_global: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Half-open range [start, stop) of global chain ids owned by ``rank``.

    Boundaries fall on EVEN ids: two adjacent chains share one Philox Box-Muller draw
    (DESIGN.md §3), so a pair never straddles two shards.
    """
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    n_pairs = (n_global + 1) // 2
    base, rem = divmod(n_pairs, world_size)
    p0 = rank * base + min(rank, rem)
    p1 = p0 + base + (1 if rank < rem else 0)
    return min(2 * p0, n_global), min(2 * p1, n_global)


def allreduce_sum(values: np.ndarray, engine=None) -> np.ndarray:
    """Sum a small f64 vector over all ranks (no-op for a single process).  `engine`: this rank's engine; when it holds a
    communicator (connect_engine) the sum is ONE ncclAllReduce on the engine's communication stream."""
    rank, size = world()
    values = np.ascontiguousarray(values, dtype=np.float64)
    if engine is not None and getattr(engine, "comm_connected", False):
        return engine.allreduce_sum(values)
    if size == 1:
        return values
    if _group is not None:
        return _group.allreduce_sum(values)
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(values.copy())
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def allreduce_xsum(records: np.ndarray, engine=None) -> np.ndarray:
    """The merged records of all ranks (reproducible sums, include/amc.h): every rank ends with the same bits, those a single
    rank holding all the chains would have.  With a communicator: amc_allreduce_xsum (one RCCL all-reduce used as a gather).
    Otherwise the same gather over the host path -- each rank fills its own slot of a zeroed buffer, so the sum adds one value
    and zeros per word and is exact in any order -- followed by the integer merge of libamc.so (amc_xsum_merge)."""
    from ._capi import AMC_XSUM_WORDS, xsum_merge
    rank, size = world()
    rec = np.ascontiguousarray(records, dtype=np.float64)
    shape = rec.shape
    if engine is not None and getattr(engine, "comm_connected", False):
        return engine.allreduce_xsum(rec).reshape(shape)
    if size == 1:
        return rec
    flat = rec.reshape(-1)
    buf = np.zeros((size, flat.size))
    buf[rank] = flat
    tot = allreduce_sum(buf.reshape(-1), None).reshape(size, -1, AMC_XSUM_WORDS)
    merged = tot[0]
    for r in range(1, size):
        merged = xsum_merge(merged, tot[r])
    return merged.reshape(shape)


def all_ranks(flag: bool, engine=None) -> bool:
    """True on every rank iff every rank passed a true flag -- ONE small sum over the ranks (the engine's communicator when it
    has one).  For loops whose exit depends on something local, a clock say: every rank must leave after the same number of
    collectives, so the ranks decide together."""
    _, size = world()
    return bool(allreduce_sum(np.array([1.0 if flag else 0.0]), engine)[0] >= size)


def barrier() -> None:
    _, size = world()
    if _group is not None:
        _group.barrier()
    elif size > 1:
        import torch.distributed as dist
        dist.barrier()
