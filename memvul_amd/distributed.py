"""Corpus sharding and the one exchange step of the multi-GPU path.

The reference's inference is single-process (predict_memory.py:103); each issue report is scored
independently against a read-only anchor bank (model_memory.py:133-147) and the only cross-IR state is
the metric accumulator (custom_metric.py:61,72).  So: one process per GPU, contiguous corpus shards (the
reference's positives-first order is preserved inside the concatenation, reader_memory.py:150-152), no
data-path collective, and ONE all-gather (RCCL over xGMI when the backend is "nccl") of the per-rank
``(score fp32, label u8)`` sufficient statistics at the end, after which every rank can run
``find_best_thres`` / ROC-AUC / AP on the concatenation (bit-identical to a single-GPU run).

Two transports behind the same four functions (all_gather_rows / all_gather_stats / barrier / all_reduce_max):

* RCCL bound directly inside libmemvul_hip.so (``init_rccl(engine)`` -> mv_comm_init / mv_comm_allgather): the GPU path.
  The collective runs on the engine's stream, the unique id travels through a file every rank of the node can reach,
  and the process never imports torch — so there is no second HIP runtime in the process and no load-order rule.
* torch.distributed (``init_process_group``): the gloo harness of the CPU tests (world size 2 here), imported lazily.

A third, ``init_tcp``, is the fallback of the RCCL one: a rank-0 socket hub that serves the same ``comm_allgather`` call with
the launcher's MASTER_ADDR / MASTER_PORT, so that a node whose RCCL cannot be initialised (library missing, IPC mode) still
gets its whole-job measurement — the data path has no collective, only the 8 B per issue report of statistics cross ranks.
bench.py says which transport carried a run.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import numpy as np


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [first, first+count) of an n-item corpus: ceil(n/world) per rank, the tail
    ranks possibly shorter/empty."""
    per = (n + world - 1) // world
    first = min(n, rank * per)
    return first, max(0, min(n, first + per) - first)


def env_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


_rccl_engine = None  # the Engine whose communicator carries the exchange (init_rccl)


def rccl_id_path() -> str:
    """A path every rank of this node derives identically: launcher port + the launcher's pid (all ranks are children of
    one torchrun / mpirun process); rank 0 publishes the RCCL unique id there and removes it when the communicator goes."""
    import tempfile

    tag = os.environ.get("MEMVUL_RCCL_ID_TAG") or f"{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'x')}_{os.getppid()}"
    return os.path.join(tempfile.gettempdir(), f"memvul_rccl_{tag}.id")


def init_rccl(engine, rank: Optional[int] = None, world: Optional[int] = None, id_path: Optional[str] = None):
    """Make `engine`'s RCCL communicator the transport of this module (GPU runs).  world == 1: a no-op transport."""
    global _rccl_engine
    r, _, w = env_world()
    rank = r if rank is None else rank
    world = w if world is None else world
    engine.comm_init(rank, world, id_path or (rccl_id_path() if world > 1 else None))
    _rccl_engine = engine
    return engine


class _TcpComm:
    """``comm_allgather`` over sockets: every rank sends its block to rank 0, rank 0 returns the rank-ordered stack.  Same call
    surface as the Engine's RCCL methods (comm_world / comm_allgather / comm_destroy), torch-free."""

    def __init__(self, rank: int, world: int, addr: str, port: int, timeout_s: float = 120.0):
        import socket
        import time

        self.rank, self.comm_world = rank, world
        self.peers = []
        self.sock = None
        if world <= 1:
            return
        if rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(world)
            srv.settimeout(timeout_s)
            by_rank = {}
            while len(by_rank) < world - 1:
                c, _ = srv.accept()
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                by_rank[int.from_bytes(self._recvn(c, 4), "little")] = c
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
                        raise
                    time.sleep(0.05)
            c.settimeout(None)
            c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            c.sendall(int(rank).to_bytes(4, "little"))
            self.sock = c

    @staticmethod
    def _recvn(c, n: int) -> bytes:
        buf = bytearray()
        while len(buf) < n:
            part = c.recv(min(1 << 20, n - len(buf)))
            if not part:
                raise ConnectionError("peer closed the statistics socket")
            buf += part
        return bytes(buf)

    def comm_allgather(self, arr: np.ndarray) -> np.ndarray:
        arr = np.ascontiguousarray(arr)
        if self.comm_world <= 1:
            return arr[None].copy()
        nb = arr.nbytes
        if self.rank == 0:
            blocks = [arr.tobytes()] + [self._recvn(c, nb) for c in self.peers]
            whole = b"".join(blocks)
            for c in self.peers:
                c.sendall(whole)
        else:
            self.sock.sendall(arr.tobytes())
            whole = self._recvn(self.sock, nb * self.comm_world)
        return np.frombuffer(whole, arr.dtype).reshape((self.comm_world,) + arr.shape).copy()

    def comm_destroy(self):
        for c in self.peers + ([self.sock] if self.sock is not None else []):
            try:
                c.close()
            except OSError:
                pass
        self.peers, self.sock = [], None


def init_tcp(rank: Optional[int] = None, world: Optional[int] = None, addr: Optional[str] = None, port: Optional[int] = None):
    """Fallback transport (see the module docstring): rank 0 listens on MASTER_ADDR : MASTER_PORT + 1."""
    global _rccl_engine
    r, _, w = env_world()
    rank = r if rank is None else rank
    world = w if world is None else world
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = port if port is not None else int(os.environ.get("MASTER_PORT", "29500")) + 1
    _rccl_engine = _TcpComm(rank, world, addr, port)
    return _rccl_engine


def shutdown_rccl():
    global _rccl_engine
    if _rccl_engine is not None:
        _rccl_engine.comm_destroy()
        _rccl_engine = None


def _rccl_world() -> int:
    return getattr(_rccl_engine, "comm_world", 1) if _rccl_engine is not None else 0


def _rccl_all_gather_rows(rows: np.ndarray) -> np.ndarray:
    world = _rccl_world()
    counts = _rccl_engine.comm_allgather(np.array([rows.shape[0]], np.int64)).reshape(world)
    n_max = max(int(counts.max()), 1)
    block = np.zeros((n_max, rows.shape[1]), np.float32)
    block[: rows.shape[0]] = rows
    out = _rccl_engine.comm_allgather(block)  # [world, n_max, k]
    return np.concatenate([out[r, : int(counts[r])] for r in range(world)])


def init_process_group(backend: Optional[str] = None):
    """Initialise torch.distributed from the torchrun environment (RANK/WORLD_SIZE/MASTER_*)."""
    import torch.distributed as dist

    if dist.is_initialized():
        return dist
    rank, local_rank, world = env_world()
    if backend is None:
        import torch

        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    if backend == "nccl":
        import torch

        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    return dist


def all_gather_rows(rows: np.ndarray, device=None) -> np.ndarray:
    """All-gather per-rank fp32 row blocks ``[n_r, k]`` (n_r may differ per rank); returns the rank-ordered concatenation
    ``[sum n_r, k]``.  One collective on a zero-padded ``[n_max, k]`` block per rank plus a tiny count gather."""
    rows = np.ascontiguousarray(rows, np.float32)
    if _rccl_engine is not None:
        if rows.ndim != 2:
            raise ValueError("all_gather_rows expects [n, k]")
        return _rccl_all_gather_rows(rows) if _rccl_world() > 1 else rows.copy()
    import torch
    import torch.distributed as dist

    if rows.ndim != 2:
        raise ValueError("all_gather_rows expects [n, k]")
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return rows.copy()
    world = dist.get_world_size()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    k = rows.shape[1]
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    n_max = max(max(counts), 1)
    block = torch.zeros((n_max, k), dtype=torch.float32)
    block[: rows.shape[0]] = torch.from_numpy(rows)
    block = block.to(device)
    out = torch.empty((world * n_max, k), dtype=torch.float32, device=device)  # rank-major concatenation
    dist.all_gather_into_tensor(out, block)
    out = out.cpu().numpy().reshape(world, n_max, k)
    return np.concatenate([out[r, : counts[r]] for r in range(world)])


def all_gather_stats(scores: np.ndarray, labels: np.ndarray, device=None) -> Tuple[np.ndarray, np.ndarray]:
    """All-gather the per-rank ``(score, label)`` arrays; returns the rank-ordered concatenation, trimmed
    to the true per-rank counts (labels ride along as 0.0/1.0 in the same block: one collective)."""
    scores = np.ascontiguousarray(scores, np.float32)
    labels = np.ascontiguousarray(labels, np.uint8)
    out = all_gather_rows(np.stack([scores, labels.astype(np.float32)], 1) if len(scores) else np.zeros((0, 2), np.float32), device)
    return out[:, 0].copy(), out[:, 1].astype(np.uint8)


def barrier():
    if _rccl_engine is not None:
        if _rccl_world() > 1:
            _rccl_engine.comm_allgather(np.zeros(1, np.int32))  # returns when every rank has contributed
        return
    import torch.distributed as dist

    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def all_reduce_max(x: float) -> float:
    if _rccl_engine is not None:
        return float(_rccl_engine.comm_allgather(np.array([x], np.float64)).max()) if _rccl_world() > 1 else x
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size() == 1:
        return x
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
