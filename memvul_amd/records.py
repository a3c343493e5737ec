"""JSON-lines records of the test branch (model_memory.py:169-191 -> predict_memory.py:111), built from arrays.

One line per batch: ``[{"Issue_Url": ..., "label": ..., "predict": {cwe: P(same), ...}}, ...]`` — byte for byte what
``json.dumps`` gives for the list ``make_output_human_readable`` returns.  The cost is Python's shortest-repr of
B x G doubles (~0.65 us each: ~11 k issue reports/s per process at G = 124), so the formatter lives in this import-light
module (json + numpy only, nothing that touches the GPU runtime) and can be fanned out over spawned worker processes.
"""
from __future__ import annotations

import json
from collections import deque
from typing import Deque, List, Optional, Sequence, Tuple

import numpy as np


def record_layout(golden_labels: Sequence[str]) -> Tuple[np.ndarray, str, List[str]]:
    """Anchor columns and the %-format of one record's ``predict`` object.  ``vote_num[name] = p`` in anchor order:
    a later anchor with the same label overwrites an earlier one but keeps its position (model_memory.py:181-183)."""
    order = {name: i for i, name in enumerate(golden_labels)}
    cols = np.fromiter(order.values(), dtype=np.int64, count=len(order))
    names = list(order.keys())
    fmt = ", ".join(json.dumps(name).replace("%", "%%") + ": %r" for name in names)
    return cols, fmt, names


def native_formatter():
    """``mv_format_records`` of libmemvul_hip.so (host-only code of the library: CPython's repr(float) restated in C++, ~40 ns per double instead of 0.65 us,
    and the GIL is released while it runs), or None where the library cannot be loaded — the Python formatter below gives the same bytes."""
    try:
        from .binding import load_library

        lib = load_library()
        return lib if hasattr(lib, "mv_format_records") else None
    except Exception:  # no library on this machine (a CPU-only checkout before the build): the Python formatter
        return None


def format_batch_native(lib, names: List[str], urls: Sequence[str], labels: Sequence[str], sel: np.ndarray) -> Optional[str]:
    """``format_batch`` through the library; None = this batch holds a non-finite value (or the call failed): format it in Python."""
    import ctypes as C

    sel = np.ascontiguousarray(sel, np.float64)
    rows, cols = sel.shape
    pre = [('{"Issue_Url": ' + json.dumps(u) + ', "label": ' + json.dumps(lab) + ', "predict": {').encode() for u, lab in zip(urls, labels)]
    pieces = [((", " if i else "") + json.dumps(name) + ": ").encode() for i, name in enumerate(names)]
    if len(pre) != rows or len(pieces) != cols:
        return None
    pre_off = np.zeros(rows + 1, np.int64)
    np.cumsum([len(b) for b in pre], out=pre_off[1:])
    piece_off = np.zeros(cols + 1, np.int64)
    np.cumsum([len(b) for b in pieces], out=piece_off[1:])
    pre_blob, piece_blob = b"".join(pre), b"".join(pieces)
    cap = int(pre_off[-1]) + rows * (int(piece_off[-1]) + 26 * cols + 8) + 16
    out = C.create_string_buffer(cap)
    n = C.c_int64(0)
    rc = lib.mv_format_records(pre_blob, pre_off.ctypes.data, rows, piece_blob, piece_off.ctypes.data, cols, b"}}", sel.ctypes.data, out, cap, C.byref(n))
    return out.raw[:n.value].decode() if rc == 0 else None


def format_batch(fmt: str, names: List[str], urls: Sequence[str], labels: Sequence[str], sel: np.ndarray, lib=None) -> str:
    """``sel``: float64 [B, len(names)] = P(same) per issue report and anchor label.  ``lib``: native_formatter() (same bytes, faster)."""
    sel = np.asarray(sel, np.float64)
    if lib is not None and sel.ndim == 2 and sel.shape[1] == len(names):
        s = format_batch_native(lib, names, urls, labels, sel)
        if s is not None:
            return s
    if not np.isfinite(sel).all():  # json spells these NaN / Infinity; a softmax never produces them
        return json.dumps([{"Issue_Url": u, "label": lab, "predict": dict(zip(names, row))} for u, lab, row in zip(urls, labels, sel.tolist())])
    out = ['{"Issue_Url": ' + json.dumps(u) + ', "label": ' + json.dumps(lab) + ', "predict": {' + fmt % tuple(row) + "}}"
           for u, lab, row in zip(urls, labels, sel.tolist())]
    return "[" + ", ".join(out) + "]"


class RecordWriter:
    """Writes batches as JSON lines in submission order; with ``workers > 0`` the formatting runs in that many spawned
    processes (spawn, not fork: the parent may hold an initialised GPU runtime) while this object only orders and writes."""

    def __init__(self, path: str, golden_labels: Sequence[str], workers: int = 0) -> None:
        self._cols, self._fmt, self._names = record_layout(golden_labels)
        self._lib = native_formatter() if not (workers and workers > 0) else None  # (worker processes keep the import-light Python formatter)
        self._f = open(path, "w")
        self._pool = None
        self._pending: Deque = deque()
        if workers and workers > 0:
            import multiprocessing as mp

            self._pool = mp.get_context("spawn").Pool(int(workers))
        self._depth = 2 * int(workers) if workers else 0

    def submit(self, urls: Sequence[str], labels: Sequence[str], p_same: np.ndarray) -> None:
        sel = np.asarray(p_same)[:, self._cols].astype(np.float64)
        if self._pool is None:
            self._f.write(format_batch(self._fmt, self._names, urls, labels, sel, self._lib) + "\n")
            return
        self._pending.append(self._pool.apply_async(format_batch, (self._fmt, self._names, list(urls), list(labels), sel)))
        while len(self._pending) > self._depth:
            self._f.write(self._pending.popleft().get() + "\n")

    def close(self) -> None:
        try:
            while self._pending:
                self._f.write(self._pending.popleft().get() + "\n")
        finally:
            if self._pool is not None:
                self._pool.close()
                self._pool.join()
                self._pool = None
            self._f.close()

    def __enter__(self) -> "RecordWriter":
        return self

    def __exit__(self, *exc) -> Optional[bool]:
        self.close()
        return None
