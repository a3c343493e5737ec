"""Corpus sharding and the one exchange step of the multi-GPU path.

The reference's inference is single-process (predict_memory.py:103); each issue report is scored
independently against a read-only anchor bank (model_memory.py:133-147) and the only cross-IR state is
the metric accumulator (custom_metric.py:61,72).  So: one process per GPU, contiguous corpus shards (the
reference's positives-first order is preserved inside the concatenation, reader_memory.py:150-152), no
data-path collective, and ONE all-gather of the per-rank ``(score fp32, label u8)`` sufficient statistics at the
end, after which every rank can run ``find_best_thres`` / ROC-AUC / AP on the concatenation (bit-identical to a
single-GPU run).

Transport (``init_transport``), torch-free:

1. every rank joins a small rendezvous hub on ``MASTER_ADDR : MASTER_PORT + 1`` (rank 0 listens; the handshake carries the
   rank and a run token, both checked);
2. over it the ranks AGREE on the data transport: each reports whether RCCL is usable in its process
   (``mv_comm_prepare``), rank 0 draws the ncclUniqueId (``mv_comm_unique_id``) and broadcasts it, every rank calls
   ``mv_comm_init`` and reports the outcome — only when ALL ranks succeeded does the run use RCCL bound inside
   libmemvul_hip.so (collective on the engine's stream, over xGMI); otherwise ALL ranks use the hub itself as the
   transport.  No rank decides alone, there is no id file in a shared temp directory and no single-node assumption.
   (A rank that dies INSIDE the collective ncclCommInitRank strands the others until RCCL's own timeout; the hub sockets
   themselves never block longer than HUB_IO_TIMEOUT_S during the agreement and HUB_DATA_TIMEOUT_S — the limit on rank skew — once the
   hub is the data transport, so a hung peer ends in a RuntimeError on every rank, not in a hang.)
3. ``shutdown()`` tears the transport down; a later ``init_transport`` starts from scratch.

The data path has no collective, only 8 B per issue report of statistics cross ranks, so a run on the hub is still a
whole-job measurement; bench.py reports which transport carried it.  ``torch.distributed`` appears only as the gloo
harness of the CPU tests (``init_process_group("gloo")``).
"""
from __future__ import annotations

import hashlib
import os
from typing import Optional, Tuple

import numpy as np

RENDEZVOUS_TIMEOUT_S = 180.0
# no hub socket ever blocks for ever.  Two limits: during the transport AGREEMENT (rendezvous, RCCL probe, unique-id broadcast,
# ncclCommInitRank outcome) a peer that hangs without closing its socket — or a rank stranded inside a collective ncclCommInitRank
# that another rank failed out of — surfaces as a RuntimeError on every waiting rank after HUB_IO_TIMEOUT_S; once the hub IS the data
# transport its collectives wait up to HUB_DATA_TIMEOUT_S, the hard limit on legitimate rank skew (a rank that finishes its shard
# early waits for the slowest one inside the barrier / all-gather; rank 0 concatenating part files between two barriers)
HUB_IO_TIMEOUT_S = float(os.environ.get("MEMVUL_HUB_TIMEOUT_S", "900"))
HUB_DATA_TIMEOUT_S = float(os.environ.get("MEMVUL_HUB_DATA_TIMEOUT_S", str(6 * 3600)))
EXIT_PORT_IN_USE = 98  # exit code of a rank 0 whose hub port was taken by somebody else between the launcher's probe and its own bind (bench.self_launch retries)


class HubPortInUse(RuntimeError):
    """rank 0 could not bind MASTER_PORT + 1: another process holds it."""


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [first, first+count) of an n-item corpus: ceil(n/world) per rank, the tail
    ranks possibly shorter/empty."""
    per = (n + world - 1) // world
    first = min(n, rank * per)
    return first, max(0, min(n, first + per) - first)


def env_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def run_token() -> bytes:
    """16 bytes every rank of one launch derives identically (launcher address, port and run id, or $MEMVUL_RUN_TOKEN):
    keeps a stray connection or a rank of ANOTHER job out of the hub.  Not a secret."""
    tag = os.environ.get("MEMVUL_RUN_TOKEN") or "|".join(os.environ.get(k, "") for k in ("MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"))
    return hashlib.sha256(tag.encode()).digest()[:16]


class _Hub:
    """Rank-0 socket hub: ``comm_allgather`` (every rank sends its block to rank 0, rank 0 returns the rank-ordered stack) and
    ``bcast`` (rank 0 -> all).  Same call surface as the Engine's RCCL methods (comm_world / comm_allgather / comm_destroy)."""

    def __init__(self, rank: int, world: int, addr: str, port: int, timeout_s: float = RENDEZVOUS_TIMEOUT_S):
        import socket
        import time

        self.rank, self.comm_world = rank, world
        self.peers = []
        self.sock = None
        if world <= 1:
            return
        token = run_token()
        if rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                srv.bind((addr, port))
            except OSError as e:
                import errno

                srv.close()
                if e.errno == errno.EADDRINUSE:
                    raise HubPortInUse(f"rendezvous hub: {addr}:{port} (MASTER_PORT + 1) is in use by another process") from e
                raise
            srv.listen(world + 8)
            deadline = time.time() + timeout_s
            by_rank = {}
            while len(by_rank) < world - 1:
                srv.settimeout(max(0.1, deadline - time.time()))
                try:
                    c, _ = srv.accept()
                except socket.timeout:
                    srv.close()
                    raise RuntimeError(f"rendezvous: only {len(by_rank) + 1} of {world} ranks reached {addr}:{port} within {timeout_s:.0f} s")
                try:
                    c.settimeout(10.0)
                    hello = self._recvn(c, 20)
                    r = int.from_bytes(hello[:4], "little")
                    if hello[4:] != token or not (0 < r < world) or r in by_rank:
                        raise ConnectionError("bad handshake")
                except (OSError, ConnectionError, RuntimeError):
                    c.close()  # not one of this run's ranks (or silent): drop it and keep listening
                    continue
                c.settimeout(HUB_IO_TIMEOUT_S)
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                by_rank[r] = c
            srv.close()
            self.peers = [by_rank[r] for r in range(1, world)]
        else:
            t0 = time.time()
            while True:
                try:
                    c = socket.create_connection((addr, port), timeout=5.0)
                    break
                except OSError:
                    if time.time() - t0 > timeout_s:
                        raise RuntimeError(f"rendezvous: rank 0 not reachable at {addr}:{port} within {timeout_s:.0f} s")
                    time.sleep(0.05)
            c.settimeout(HUB_IO_TIMEOUT_S)
            c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            c.sendall(int(rank).to_bytes(4, "little") + token)
            self.sock = c

    @staticmethod
    def _recvn(c, n: int) -> bytes:
        import socket

        buf = bytearray()
        while len(buf) < n:
            try:
                part = c.recv(min(1 << 20, n - len(buf)))
            except socket.timeout:
                raise RuntimeError(f"rendezvous hub: a peer stayed silent for {c.gettimeout() or 0:.0f} s (hung rank?) — giving up instead of waiting for ever")
            if not part:
                raise ConnectionError("peer closed the rendezvous socket")
            buf += part
        return bytes(buf)

    @staticmethod
    def _send(c, data: bytes):
        """sendall with the same failure surface as _recvn: a peer that stopped reading is a RuntimeError, not a bare socket.timeout."""
        import socket

        try:
            c.sendall(data)
        except socket.timeout:
            raise RuntimeError(f"rendezvous hub: a peer accepted no data for {c.gettimeout() or 0:.0f} s (hung rank?) — giving up instead of waiting for ever")

    def set_timeout(self, seconds: float):
        """The limit every later send / receive on the hub's sockets waits for (agreement: HUB_IO_TIMEOUT_S; data phase: HUB_DATA_TIMEOUT_S)."""
        for c in self.peers + ([self.sock] if self.sock is not None else []):
            c.settimeout(seconds)

    def comm_allgather(self, arr: np.ndarray) -> np.ndarray:
        arr = np.ascontiguousarray(arr)
        if self.comm_world <= 1:
            return arr[None].copy()
        nb = arr.nbytes
        if self.rank == 0:
            blocks = [arr.tobytes()] + [self._recvn(c, nb) for c in self.peers]
            whole = b"".join(blocks)
            for c in self.peers:
                self._send(c, whole)
        else:
            self._send(self.sock, arr.tobytes())
            whole = self._recvn(self.sock, nb * self.comm_world)
        return np.frombuffer(whole, arr.dtype).reshape((self.comm_world,) + arr.shape).copy()

    def bcast(self, payload: Optional[bytes], nbytes: int) -> bytes:
        """rank 0's `payload` (exactly nbytes) to every rank."""
        if self.comm_world <= 1:
            return payload
        if self.rank == 0:
            assert payload is not None and len(payload) == nbytes
            for c in self.peers:
                self._send(c, payload)
            return payload
        return self._recvn(self.sock, nbytes)

    def comm_destroy(self):
        for c in self.peers + ([self.sock] if self.sock is not None else []):
            try:
                c.close()
            except OSError:
                pass
        self.peers, self.sock = [], None


_comm = None        # the transport of all_gather_rows / barrier / all_reduce_max: an Engine (RCCL) or a _Hub
_comm_note = "none"


def init_transport(engine=None, rank: Optional[int] = None, world: Optional[int] = None, prefer: str = "rccl",
                   addr: Optional[str] = None, port: Optional[int] = None) -> str:
    """Set up the statistics exchange for this process (see the module docstring) and return a description of the transport
    that ALL ranks agreed on: "rccl ..." or "tcp hub ...".  prefer="tcp" skips RCCL.  world == 1: a no-op transport."""
    global _comm, _comm_note
    shutdown()
    r, _, w = env_world()
    rank = r if rank is None else rank
    world = w if world is None else world
    if world <= 1:
        if engine is not None:
            engine.comm_init(rank, 1)
            _comm = engine
        else:
            _comm = _Hub(0, 1, "", 0)
        _comm_note = "none (one rank)"
        return _comm_note
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = port if port is not None else int(os.environ.get("MASTER_PORT", "29500")) + 1
    hub = _Hub(rank, world, addr, port)
    try:
        return _agree_on_transport(hub, engine, rank, world, prefer, addr, port)
    except BaseException:  # a peer vanished mid-agreement (ConnectionError), Ctrl-C ...: leave no socket and no half-made communicator behind
        hub.comm_destroy()
        if engine is not None:
            try:
                engine.comm_destroy()
            except RuntimeError:
                pass
        _comm, _comm_note = None, "none"
        raise


def _agree_on_transport(hub, engine, rank: int, world: int, prefer: str, addr: str, port: int) -> str:
    global _comm, _comm_note
    why = "requested" if prefer != "rccl" else ""
    if prefer == "rccl" and engine is None:
        why = "no engine given"
    if prefer == "rccl" and engine is not None:
        # step 1: is RCCL usable in EVERY process?  (nobody enters the collective init unless all say yes)
        try:
            engine.comm_prepare()
            mine, err = 1, ""
        except RuntimeError as e:
            mine, err = 0, str(e)
        if int(hub.comm_allgather(np.array([mine], np.uint8)).min()) == 0:
            why = "RCCL not usable on every rank" + (f" (here: {err[:100]})" if err else "")
        else:
            # step 2: rank 0 draws the unique id and broadcasts [ok byte | 128 id bytes]
            msg = None
            if rank == 0:
                try:
                    msg = b"\x01" + engine.comm_unique_id().ljust(128, b"\0")
                except RuntimeError as e:
                    msg, err = b"\x00" + b"\0" * 128, str(e)
            msg = hub.bcast(msg, 129)
            if msg[0] != 1:
                why = "rank 0 could not draw a unique id" + (f" ({err[:100]})" if err else "")
            else:
                # step 3: the collective init; every rank reports its outcome, all must succeed
                try:
                    engine.comm_init(rank, world, msg[1:129])
                    ok, err = 1, ""
                except RuntimeError as e:
                    ok, err = 0, str(e)
                if int(hub.comm_allgather(np.array([ok], np.uint8)).min()) == 1:
                    hub.comm_destroy()
                    _comm = engine
                    _comm_note = "rccl (bound in libmemvul_hip.so, engine stream; unique id over the rendezvous socket)"
                    return _comm_note
                try:  # EVERY rank leaves the attempt behind (a rank whose own init failed too: its handle goes back to one rank)
                    engine.comm_destroy()
                except RuntimeError:
                    pass
                why = "ncclCommInitRank failed on a rank" + (f" (here: {err[:100]})" if err else "")
    _comm = hub
    hub.set_timeout(HUB_DATA_TIMEOUT_S)  # from here on the sockets carry the job's barriers and its one all-gather: the skew limit applies
    _comm_note = f"tcp hub on {addr}:{port} ({why or 'fallback'})"
    return _comm_note


def transport_note() -> str:
    return _comm_note


def shutdown():
    """Tear the transport down (idempotent).  Every driver calls it in a ``finally``: a second sharded call in the same
    process then starts from scratch instead of inheriting a stale communicator."""
    global _comm, _comm_note
    if _comm is not None:
        try:
            _comm.comm_destroy()
        finally:
            _comm, _comm_note = None, "none"


def transport_world() -> int:
    """Ranks of the ACTIVE transport (RCCL communicator or hub) — what all_gather_stats gathers over; 0 before init_transport."""
    return _world()


def _world() -> int:
    return getattr(_comm, "comm_world", 1) if _comm is not None else 0


def _comm_all_gather_rows(rows: np.ndarray) -> np.ndarray:
    world = _world()
    counts = _comm.comm_allgather(np.array([rows.shape[0]], np.int64)).reshape(world)
    n_max = max(int(counts.max()), 1)
    block = np.zeros((n_max, rows.shape[1]), np.float32)
    block[: rows.shape[0]] = rows
    out = _comm.comm_allgather(block)  # [world, n_max, k]
    return np.concatenate([out[r, : int(counts[r])] for r in range(world)])


def init_process_group(backend: str = "gloo"):
    """torch.distributed from the torchrun environment — the gloo harness of the CPU tests only (GPU runs use
    init_transport and never import torch)."""
    import torch.distributed as dist

    if backend != "gloo":
        raise ValueError("torch.distributed is the CPU test harness here (backend 'gloo'); GPU runs use init_transport")
    if dist.is_initialized():
        return dist
    rank, _, world = env_world()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    dist.init_process_group(backend, rank=rank, world_size=world)
    return dist


def all_gather_rows(rows: np.ndarray) -> np.ndarray:
    """All-gather per-rank fp32 row blocks ``[n_r, k]`` (n_r may differ per rank); returns the rank-ordered concatenation
    ``[sum n_r, k]``.  One collective on a zero-padded ``[n_max, k]`` block per rank plus a tiny count gather."""
    rows = np.ascontiguousarray(rows, np.float32)
    if rows.ndim != 2:
        raise ValueError("all_gather_rows expects [n, k]")
    if _comm is not None:
        return _comm_all_gather_rows(rows) if _world() > 1 else rows.copy()
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size() == 1:
        return rows.copy()
    world = dist.get_world_size()
    k = rows.shape[1]
    n = torch.tensor([rows.shape[0]], dtype=torch.int64)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    n_max = max(max(counts), 1)
    block = torch.zeros((n_max, k), dtype=torch.float32)
    block[: rows.shape[0]] = torch.from_numpy(rows)
    out = torch.empty((world * n_max, k), dtype=torch.float32)  # rank-major concatenation
    dist.all_gather_into_tensor(out, block)
    out = out.numpy().reshape(world, n_max, k)
    return np.concatenate([out[r, : counts[r]] for r in range(world)])


def all_gather_stats(scores: np.ndarray, labels: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """All-gather the per-rank ``(score, label)`` arrays; returns the rank-ordered concatenation, trimmed
    to the true per-rank counts (labels ride along as 0.0/1.0 in the same block: one collective)."""
    scores = np.ascontiguousarray(scores, np.float32)
    labels = np.ascontiguousarray(labels, np.uint8)
    out = all_gather_rows(np.stack([scores, labels.astype(np.float32)], 1) if len(scores) else np.zeros((0, 2), np.float32))
    return out[:, 0].copy(), out[:, 1].astype(np.uint8)


def barrier():
    if _comm is not None:
        if _world() > 1:
            _comm.comm_allgather(np.zeros(1, np.int32))  # returns when every rank has contributed
        return
    import torch.distributed as dist

    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def all_reduce_max(x: float) -> float:
    if _comm is not None:
        return float(_comm.comm_allgather(np.array([x], np.float64)).max()) if _world() > 1 else x
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size() == 1:
        return x
    t = torch.tensor([x], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
