"""ctypes binding of libmemvul_hip.so (include/memvul_hip.h) and a thin ``Engine`` wrapper.

The product path has no CPU fallback: if the library cannot be loaded, or a call fails, a
``RuntimeError`` is raised.  Nothing here imports torch or anything under ``oracle/``.
"""
from __future__ import annotations

import ctypes as C
import os
import warnings
from typing import Dict, Iterable, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmemvul_hip.so")
LIB_PATH_DEV = os.path.join(_HERE, "lib", "libmemvul_hip_dev.so")  # the -DMEMVUL_DEV_SWITCHES build (memvul_amd/build.py): tests and A/B scripts only
DEV_SWITCHES = ("MEMVUL_GEMM_TILE", "MEMVUL_SHORT_VLO", "MEMVUL_RASTER", "MEMVUL_GN_MAX", "MEMVUL_NUM_CU")  # read by that build alone

MV_F32, MV_F16, MV_BF16, MV_I32, MV_I64 = 0, 1, 2, 3, 4
MV_F16X8 = 6  # compute dtype only ("precise"): fp16 MFMA sweep + one fp8 (e4m3) correction sweep per GEMM (include/memvul_hip.h)
COMPUTE_DTYPES = {"f16": MV_F16, "fast": MV_F16, "f16x8": MV_F16X8, "precise": MV_F16X8}
# The product's default is the compute dtype that holds the reference's 1e-3 logit tolerance on trained-like weights
# (model_memory.py:133-147 at config_memory.json:38's temperature): MV_F16X8.  MV_F16 ("fast") is an explicit opt-in:
# ~1.7x the rate, logits within 1e-3 only on small-logit models (measured 3.0-5.6e-3 at |logit| ~ 3).
DEFAULT_COMPUTE = "precise"


def default_compute() -> str:
    """$MEMVUL_COMPUTE (f16 | fast | f16x8 | precise) or the contract-holding default."""
    return os.environ.get("MEMVUL_COMPUTE", DEFAULT_COMPUTE)


def compute_dtype_of(name_or_code) -> int:
    """"f16" (alias "fast") | "f16x8" (alias "precise") or the numeric mv_dtype -> the code mv_finalize_weights takes; None = the
    default (default_compute()); anything else raises."""
    if name_or_code is None:
        name_or_code = default_compute()
    if isinstance(name_or_code, str):
        if name_or_code.lower() not in COMPUTE_DTYPES:
            raise ValueError(f"unknown compute dtype {name_or_code!r}: expected one of {sorted(COMPUTE_DTYPES)}")
        return COMPUTE_DTYPES[name_or_code.lower()]
    if int(name_or_code) not in (MV_F16, MV_F16X8):
        raise ValueError(f"unknown compute dtype code {name_or_code!r}: MV_F16 = {MV_F16} or MV_F16X8 = {MV_F16X8}")
    return int(name_or_code)


NUM_KERNEL_CLASSES = 14

# every symbol include/memvul_hip.h declares (tests check the .so exports all of them)
ABI_SYMBOLS = [
    "mv_create", "mv_destroy", "mv_last_error", "mv_sync", "mv_load_tensor", "mv_finalize_weights",
    "mv_anchor_reset", "mv_anchor_append", "mv_anchor_count", "mv_anchor_get", "mv_anchor_set",
    "mv_forward", "mv_forward_groups", "mv_forward_ragged", "mv_forward_ragged_begin", "mv_forward_ragged_end", "mv_encode", "mv_match", "mv_topk", "mv_corpus_upload", "mv_corpus_run", "mv_corpus_run_len",
    "mv_corpus_results", "mv_x8_saturation", "mv_attention_concentration", "mv_set_streams", "mv_profile_enable", "mv_profile_select", "mv_profile_read", "mv_kernel_class_name",
    "mv_debug_encode", "mv_debug_read", "mv_test_gemm", "mv_test_gemm_pp", "mv_test_e4m3", "mv_format_records", "mv_comm_prepare", "mv_comm_unique_id", "mv_comm_init", "mv_comm_allgather",
    "mv_comm_destroy", "mv_comm_info", "mv_device_count",
]


class MvConfig(C.Structure):
    _fields_ = [
        ("vocab_size", C.c_int32), ("hidden", C.c_int32), ("layers", C.c_int32), ("heads", C.c_int32),
        ("intermediate", C.c_int32), ("max_pos", C.c_int32), ("type_vocab", C.c_int32), ("proj_dim", C.c_int32),
        ("ln_eps", C.c_float), ("max_tokens", C.c_int32), ("max_batch", C.c_int32), ("max_anchors", C.c_int32),
        ("same_idx", C.c_int32),
    ]


_libs = {}


def load_library(path: Optional[str] = None, dev: bool = False):
    """dlopen the HIP library and declare the prototypes.  Raises RuntimeError if it is missing —
    there is deliberately no other implementation to fall back to.  dev: the development build (the same kernels, plus the
    A/B knobs of DEV_SWITCHES at mv_create); the product never asks for it."""
    path = path or (LIB_PATH_DEV if dev else (os.environ.get("MEMVUL_HIP_LIB") or LIB_PATH))
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise RuntimeError(
            f"libmemvul_hip.so not found at {path}: build it with `python -m memvul_amd.build` "
            "(hipcc --offload-arch=gfx950); the MemVul hot path has no CPU fallback"
        )
    try:
        # the product library as always (RTLD_GLOBAL); the development build, which a test process may load NEXT TO it, privately
        lib = C.CDLL(path, mode=C.RTLD_LOCAL if path == LIB_PATH_DEV else C.RTLD_GLOBAL)
    except OSError as e:  # pragma: no cover
        raise RuntimeError(f"cannot load {path}: {e}") from e
    vp, i32p, f32p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float)
    P = C.POINTER
    sig = {
        "mv_create": (C.c_int, [C.c_int, P(MvConfig), P(vp)]),
        "mv_destroy": (None, [vp]),
        "mv_last_error": (C.c_char_p, [vp]),
        "mv_sync": (C.c_int, [vp]),
        "mv_load_tensor": (C.c_int, [vp, C.c_char_p, vp, C.c_int, P(C.c_int64), C.c_int]),
        "mv_finalize_weights": (C.c_int, [vp, C.c_int]),
        "mv_anchor_reset": (C.c_int, [vp]),
        "mv_anchor_append": (C.c_int, [vp, vp, vp, C.c_int, C.c_int]),
        "mv_anchor_count": (C.c_int, [vp]),
        "mv_anchor_get": (C.c_int, [vp, vp]),
        "mv_anchor_set": (C.c_int, [vp, vp, C.c_int]),
        "mv_forward": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]),
        "mv_forward_groups": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp]),
        "mv_forward_ragged": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp]),
        "mv_forward_ragged_begin": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P(C.c_int)]),
        "mv_forward_ragged_end": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp]),
        "mv_encode": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, vp]),
        "mv_match": (C.c_int, [vp, vp, C.c_int, vp, vp, vp, vp]),
        "mv_topk": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp]),
        "mv_corpus_upload": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int]),
        "mv_corpus_run": (C.c_int, [vp, C.c_int64, C.c_int64, C.c_int, C.c_int]),
        "mv_corpus_run_len": (C.c_int, [vp, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int]),
        "mv_corpus_results": (C.c_int, [vp, C.c_int64, C.c_int64, vp, vp, vp]),
        "mv_x8_saturation": (C.c_int, [vp, C.POINTER(C.c_int64), C.c_int]),
        "mv_attention_concentration": (C.c_int, [vp, C.POINTER(C.c_float), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int]),
        "mv_set_streams": (C.c_int, [vp, C.c_int]),
        "mv_profile_enable": (C.c_int, [vp, C.c_int]),
        "mv_profile_select": (C.c_int, [vp, C.c_uint32]),
        "mv_profile_read": (C.c_int, [vp, P(C.c_double), P(C.c_int64), C.c_int]),
        "mv_kernel_class_name": (C.c_char_p, [C.c_int]),
        "mv_debug_encode": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int]),
        "mv_debug_read": (C.c_int, [vp, C.c_int, vp, C.c_int64]),
        "mv_test_gemm": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, P(C.c_float)]),
        "mv_test_gemm_pp": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_int, P(C.c_float)]),
        "mv_test_e4m3": (C.c_int, [vp, vp, C.c_int64]),
        "mv_format_records": (C.c_int, [vp, vp, C.c_int64, vp, vp, C.c_int64, C.c_char_p, vp, vp, C.c_int64, P(C.c_int64)]),
        "mv_comm_prepare": (C.c_int, [vp]),
        "mv_comm_unique_id": (C.c_int, [vp, vp, C.c_int]),
        "mv_comm_init": (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_int]),
        "mv_comm_allgather": (C.c_int, [vp, vp, vp, C.c_int64]),
        "mv_comm_destroy": (C.c_int, [vp]),
        "mv_comm_info": (C.c_int, [vp, P(C.c_int), C.c_int]),
        "mv_device_count": (C.c_int, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    # the three tiny getters the hot loop calls once per batch, bound a second time WITHOUT releasing the interpreter lock around the call (PyDLL): next to other
    # Python threads every release costs the scoring thread a wait for the lock that is far longer than these calls (profiles/r06_*_e2e_dropin.txt)
    quick = C.PyDLL(path)
    for name in ("mv_anchor_count", "mv_x8_saturation", "mv_attention_concentration"):
        fn = getattr(quick, name)
        fn.restype, fn.argtypes = sig[name]
    lib.quick = quick
    _libs[path] = lib
    return lib


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _as(a, dtype) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=dtype)


class Engine:
    """One handle <-> one GPU <-> one stream (see include/memvul_hip.h)."""

    def __init__(self, device: int = 0, *, vocab_size: int = 30522, layers: int = 12, max_pos: int = 512,
                 type_vocab: int = 2, ln_eps: float = 1e-12, max_tokens: int = 65536, max_batch: int = 512,
                 max_anchors: int = 1024, same_idx: int = 0, proj_dim: int = 512, dev: bool = False):
        """proj_dim: width of the embedding the matcher runs on — 512 (the header output: use_header=True, every reference
        config) or 768 (use_header=False: the pooler output, no ``_projector_single`` in the state dict).  dev: load the development
        build (tests / A/B scripts: the only one that reads DEV_SWITCHES)."""
        self._lib = load_library(dev=dev)
        self.P = int(proj_dim)
        self.cfg = MvConfig(vocab_size, 768, layers, 12, 3072, max_pos, type_vocab, self.P, ln_eps, max_tokens,
                            max_batch, max_anchors, same_idx)
        h = C.c_void_p()
        rc = self._lib.mv_create(device, C.byref(self.cfg), C.byref(h))
        if rc != 0:
            msg = self._lib.mv_last_error(None)
            raise RuntimeError(f"mv_create failed ({rc}): {msg.decode() if msg else ''}")
        self._h = h
        self.device = device
        self._tickets = []  # forward_by_length_begin: batches in flight, oldest first

    # -- plumbing
    def _check(self, rc: int, what: str):
        if rc != 0:
            msg = self._lib.mv_last_error(self._h)
            raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mv_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self._check(self._lib.mv_sync(self._h), "mv_sync")

    # -- weights
    def load_tensor(self, name: str, arr: np.ndarray):
        if arr.dtype == np.float16:
            dt = MV_F16
        elif arr.dtype in (np.int64, np.int32):
            dt = MV_I64 if arr.dtype == np.int64 else MV_I32
        else:
            arr = _as(arr, np.float32)
            dt = MV_F32
        arr = np.ascontiguousarray(arr)
        shape = (C.c_int64 * arr.ndim)(*arr.shape)
        self._check(self._lib.mv_load_tensor(self._h, name.encode(), _ptr(arr), dt, shape, arr.ndim), f"mv_load_tensor({name})")

    def load_state_dict(self, sd: Dict[str, np.ndarray], compute_dtype=None):
        """``sd``: reference ``state_dict`` keys -> arrays (torch tensors are converted by the caller)."""
        for k, v in sd.items():
            a = np.asarray(v)
            if a.ndim == 0:
                continue
            self.load_tensor(k, a)
        self._check(self._lib.mv_finalize_weights(self._h, compute_dtype_of(compute_dtype)), "mv_finalize_weights")
        self._precise = compute_dtype_of(compute_dtype) == MV_F16X8
        self._sat_warned = False

    def x8_saturation(self, reset: bool = False) -> int:
        """MV_F16X8: activation elements (raw stream, attention context, GELU output) that fell outside the +-112 range of the fp8
        correction planes since the engine was created / last reset (mv_x8_saturation; synchronises).  Such an element keeps fp16
        accuracy and loses its correction term."""
        n = C.c_int64(0)
        self._check(self._lib.quick.mv_x8_saturation(self._h, C.byref(n), int(bool(reset))), "mv_x8_saturation")
        return int(n.value)

    def attention_concentration(self, reset: bool = False):
        """MV_F16X8: (max_collision, items_over, items_total) of mv_attention_concentration — the largest sum_{j >= 2} p[CLS row][j]^2 over every (sequence,
        head, layer) processed so far, the number of them above 0.25 — more than half of a head's [CLS]-row attention on one token that is neither [CLS] nor
        [SEP]: the regime outside the measured envelope of the default form (include/memvul_hip.h) — and the number looked at."""
        m, n, t = C.c_float(0), C.c_int64(0), C.c_int64(0)
        self._check(self._lib.quick.mv_attention_concentration(self._h, C.byref(m), C.byref(n), C.byref(t), int(bool(reset))), "mv_attention_concentration")
        return float(m.value), int(n.value), int(t.value)

    def _check_saturation(self):
        """Called after the host-synchronous entry points of the precise mode: warn ONCE when the fp8 planes clamped anything."""
        if getattr(self, "_precise", False) and not getattr(self, "_conc_warned", False):
            m, n, t = self.attention_concentration()
            if t >= 100 and n > 0.02 * t:  # systematic, not the odd head of the odd sequence
                self._conc_warned = True
                warnings.warn(f"MV_F16X8: in {n} of {t} (sequence, head, layer) items the [CLS] row puts more than half of a head's attention on ONE ordinary token "
                              f"(collision mass up to {m:.2f}): the 1e-3 logit tolerance of the default form is backed by measurement for diffuse attention and "
                              "for attention sinks on [CLS] / [SEP] only (profiles/r06_n_sink_envelope.txt: 0.8 - 2.7e-3 for such a sink); "
                              "MEMVUL_CLS_ASIDE=0 MEMVUL_QKV_ASIDE=qkv is the most conservative form (include/memvul_hip.h mv_attention_concentration)",
                              RuntimeWarning, stacklevel=3)
        if getattr(self, "_precise", False) and not self._sat_warned:
            n = self.x8_saturation()
            if n:
                self._sat_warned = True
                warnings.warn(f"MV_F16X8: {n} activation element(s) exceeded the +-112 range of the fp8 correction planes and were "
                              "computed at fp16 accuracy; the 1e-3 logit tolerance is not backed by measurement for this model "
                              "(include/memvul_hip.h mv_x8_saturation; Engine.x8_saturation())", RuntimeWarning, stacklevel=3)

    # -- anchors
    def anchor_reset(self):
        self._check(self._lib.mv_anchor_reset(self._h), "mv_anchor_reset")

    def anchor_append(self, ids: np.ndarray, lens: np.ndarray):
        ids, lens = _as(ids, np.int32), _as(lens, np.int32)
        self._check(self._lib.mv_anchor_append(self._h, _ptr(ids), _ptr(lens), ids.shape[0], ids.shape[1]), "mv_anchor_append")
        self._check_saturation()

    @property
    def n_anchors(self) -> int:
        return int(self._lib.quick.mv_anchor_count(self._h))

    def anchor_get(self) -> np.ndarray:
        out = np.empty((self.n_anchors, self.P), np.float32)
        self._check(self._lib.mv_anchor_get(self._h, _ptr(out)), "mv_anchor_get")
        return out

    def _check_width(self, a: np.ndarray, what: str):
        if a.ndim != 2 or a.shape[1] != self.P:
            raise ValueError(f"{what}: expected [n, {self.P}] embeddings (mv_config.proj_dim), got {a.shape}")

    def anchor_set(self, v: np.ndarray):
        v = _as(v, np.float32)
        self._check_width(v, "anchor_set")
        self._check(self._lib.mv_anchor_set(self._h, _ptr(v), v.shape[0]), "mv_anchor_set")

    # -- hot loop
    def forward(self, ids: np.ndarray, lens: np.ndarray, want_logits=True, want_probs=True, want_embed=False):
        ids, lens = _as(ids, np.int32), _as(lens, np.int32)
        B, S = ids.shape
        G = self.n_anchors
        logits = np.empty((B, G, 2), np.float32) if want_logits else None
        probs = np.empty((B, G, 2), np.float32) if want_probs else None
        best = np.empty((B, 2), np.float32)
        idx = np.empty((B,), np.int32)
        embed = np.empty((B, self.P), np.float32) if want_embed else None
        self._check(self._lib.mv_forward(self._h, _ptr(ids), _ptr(lens), B, S, _ptr(logits), _ptr(probs), _ptr(best),
                                         _ptr(idx), _ptr(embed)), "mv_forward")
        self._check_saturation()
        return {"logits": logits, "probs": probs, "best": best, "best_idx": idx, "embed": embed}

    # tokens below which one pad-to-longest pass is kept as it is (a pass of a few thousand tokens leaves most of the 256 persistent workgroups idle)
    BY_LENGTH_MIN_TOKENS = 16384

    def forward_by_length(self, ids: np.ndarray, lens: np.ndarray, want_logits=True, want_probs=True, want_embed=False, min_tokens: Optional[int] = None):
        """``forward`` on a pad-to-longest batch of UNSORTED issue reports (the reference's collation, predict_memory.py:97-101) without paying for the
        padding: the rows are grouped by the padded length of their OWN token count (64 .. 256 in steps of 64, 384, 512: engine.hip padded_len), every
        group is one ``mv_forward`` at its own length, and the results go back to the rows' places.  A group of fewer than ``min_tokens`` tokens travels
        with the next longer one (a pass that small leaves most of the chip idle).  Same per-row arithmetic as ``forward`` at a different padded length
        (what Engine.bucketed_sweep does to a resident corpus); a batch whose rows share one padded length is ONE call, bit for bit ``forward``."""
        ids, lens = _as(ids, np.int32), _as(lens, np.int32)
        B, S = ids.shape
        if min_tokens is None:
            min_tokens = self.BY_LENGTH_MIN_TOKENS
        if B == 0 or B * S < 2 * min_tokens:
            return self.forward(ids, lens, want_logits, want_probs, want_embed)
        ragged = getattr(self, "_forward_ragged", None)  # (a stand-in engine of the tests has only `forward`)
        if ragged is not None:
            # everything below inside the library (mv_forward_ragged): ONE release of the interpreter lock per batch
            out = {"logits": np.empty((B, self.n_anchors, 2), np.float32) if want_logits else None, "probs": np.empty((B, self.n_anchors, 2), np.float32) if want_probs else None,
                   "best": np.empty((B, 2), np.float32), "best_idx": np.empty((B,), np.int32), "embed": np.empty((B, self.P), np.float32) if want_embed else None}
            if ragged(ids, lens, min_tokens, out):
                return out
        pl = np.where(lens <= 256, (np.maximum(lens, 1) + 63) // 64 * 64, (lens + 127) // 128 * 128).astype(np.int64)
        top = int(pl.max()) if B else 0
        if int(pl.min()) == top:  # one group: one call, at the group's own length
            return self.forward(np.ascontiguousarray(ids[:, :top]) if top < S else ids, lens, want_logits, want_probs, want_embed)
        # ONE gather into length order, the groups are then row slices of it and every pass writes its results straight into its slice of the
        # length-ordered outputs (no per-group fancy indexing: that was 8 ms of the scorer thread's 15 ms per 512-report batch), ONE gather back
        G = self.n_anchors
        order = np.argsort(pl, kind="stable")
        spl = pl[order]
        ids_s, lens_s = ids[order], lens[order]
        bufs = {"logits": np.empty((B, G, 2), np.float32) if want_logits else None, "probs": np.empty((B, G, 2), np.float32) if want_probs else None,
                "best": np.empty((B, 2), np.float32), "best_idx": np.empty((B,), np.int32), "embed": np.empty((B, self.P), np.float32) if want_embed else None}
        cuts = np.flatnonzero(np.diff(spl)) + 1  # group boundaries in the length order
        ends, widths, start = [], [], 0
        for end in list(cuts) + [B]:
            width = int(spl[end - 1])
            if end < B and (end - start) * width < min_tokens:
                continue  # too small a pass: these rows travel with the next longer group
            ends.append(int(end)); widths.append(min(S, width))
            start = end
        groups = getattr(self, "_forward_groups", None)  # (a stand-in engine of the tests has only `forward`)
        if groups is not None and groups(ids_s, lens_s, ends, widths, bufs):
            pass  # ONE library call (mv_forward_groups): the passes back to back on the stream, one synchronisation, the GIL released once per batch
        else:
            start = 0
            for end, width in zip(ends, widths):
                sub = self.forward(np.ascontiguousarray(ids_s[start:end, :width]), lens_s[start:end], want_logits, want_probs, want_embed)
                for k, v in sub.items():
                    if v is not None and bufs.get(k) is not None:
                        bufs[k][start:end] = v
                start = end
        inv = np.empty(B, np.int64)
        inv[order] = np.arange(B)
        return {k: (v[inv] if v is not None else None) for k, v in bufs.items()}

    def forward_by_length_begin(self, ids: np.ndarray, lens: np.ndarray, want_logits=True, want_probs=True, want_embed=False, min_tokens: Optional[int] = None):
        """``forward_by_length`` handed over without waiting for it (mv_forward_ragged_begin): returns a ticket for ``forward_by_length_end``.  One batch per
        workspace set may be in flight (two by default); collect in the order of the calls.  A batch the asynchronous entry cannot take (too small to be
        worth grouping, too large for one upload, every workspace set busy) is scored at once and its ticket holds the results."""
        ids, lens = _as(ids, np.int32), _as(lens, np.int32)
        B, S = ids.shape
        mt = self.BY_LENGTH_MIN_TOKENS if min_tokens is None else min_tokens
        if B > 0 and B * S >= 2 * mt and len(self._tickets) < 2:
            t = C.c_int(-1)
            rc = self._lib.mv_forward_ragged_begin(self._h, _ptr(ids), _ptr(lens), B, S, int(mt), int(want_logits), int(want_probs), int(want_embed), C.byref(t))
            if rc == 0:
                self._tickets.append(t.value)
                return ("pending", t.value, B, bool(want_logits), bool(want_probs), bool(want_embed))
            if rc != -5:  # (MV_ERR_CAPACITY: below)
                self._check(rc, "mv_forward_ragged_begin")
        return ("done", self.forward_by_length(ids, lens, want_logits, want_probs, want_embed, min_tokens))

    def forward_by_length_end(self, ticket):
        if ticket[0] == "done":
            return ticket[1]
        _, t, B, want_logits, want_probs, want_embed = ticket
        if not self._tickets or self._tickets[0] != t:
            raise RuntimeError("forward_by_length_end: tickets are collected in the order they were issued")
        G = self.n_anchors
        out = {"logits": np.empty((B, G, 2), np.float32) if want_logits else None, "probs": np.empty((B, G, 2), np.float32) if want_probs else None,
               "best": np.empty((B, 2), np.float32), "best_idx": np.empty((B,), np.int32), "embed": np.empty((B, self.P), np.float32) if want_embed else None}
        self._tickets.pop(0)
        self._check(self._lib.mv_forward_ragged_end(self._h, t, _ptr(out["logits"]), _ptr(out["probs"]), _ptr(out["best"]), _ptr(out["best_idx"]), _ptr(out["embed"])),
                    "mv_forward_ragged_end")
        if not self._tickets:  # (the counters are read after a synchronisation of EVERY stream: with the next batch in flight that would wait for it — holding the lock)
            self._check_saturation()
        return out

    def _forward_ragged(self, ids: np.ndarray, lens: np.ndarray, min_tokens: int, out: Dict[str, Optional[np.ndarray]]) -> bool:
        """mv_forward_ragged into the caller's arrays; False = the batch does not fit one upload (the caller walks the groups itself)."""
        B, S = ids.shape
        rc = self._lib.mv_forward_ragged(self._h, _ptr(ids), _ptr(lens), B, S, int(min_tokens), _ptr(out.get("logits")), _ptr(out.get("probs")),
                                         _ptr(out["best"]), _ptr(out["best_idx"]), _ptr(out.get("embed")))
        if rc == -5:  # MV_ERR_CAPACITY (checked before any GPU work)
            return False
        self._check(rc, "mv_forward_ragged")
        self._check_saturation()
        return True

    def _forward_groups(self, ids: np.ndarray, lens: np.ndarray, ends, widths, out: Dict[str, Optional[np.ndarray]]) -> bool:
        """mv_forward_groups into the caller's arrays (rows in length order); False = the batch does not fit one upload (max_batch / max_tokens): the caller
        walks the groups with ``forward``."""
        B, S = ids.shape
        ge, gw = np.asarray(ends, np.int32), np.asarray(widths, np.int32)
        rc = self._lib.mv_forward_groups(self._h, _ptr(ids), _ptr(lens), B, S, len(ge), _ptr(ge), _ptr(gw), _ptr(out.get("logits")), _ptr(out.get("probs")),
                                         _ptr(out["best"]), _ptr(out["best_idx"]), _ptr(out.get("embed")))
        if rc == -5:  # MV_ERR_CAPACITY (checked before any GPU work)
            return False
        self._check(rc, "mv_forward_groups")
        self._check_saturation()
        return True

    def encode(self, ids: np.ndarray, lens: np.ndarray) -> np.ndarray:
        ids, lens = _as(ids, np.int32), _as(lens, np.int32)
        out = np.empty((ids.shape[0], self.P), np.float32)
        self._check(self._lib.mv_encode(self._h, _ptr(ids), _ptr(lens), ids.shape[0], ids.shape[1], _ptr(out)), "mv_encode")
        self._check_saturation()
        return out

    def match(self, u: np.ndarray):
        u = _as(u, np.float32)
        self._check_width(u, "match")
        B, G = u.shape[0], self.n_anchors
        logits = np.empty((B, G, 2), np.float32)
        probs = np.empty((B, G, 2), np.float32)
        best = np.empty((B, 2), np.float32)
        idx = np.empty((B,), np.int32)
        self._check(self._lib.mv_match(self._h, _ptr(u), B, _ptr(logits), _ptr(probs), _ptr(best), _ptr(idx)), "mv_match")
        return {"logits": logits, "probs": probs, "best": best, "best_idx": idx}

    def topk(self, u: np.ndarray, k: int):
        u = _as(u, np.float32)
        self._check_width(u, "topk")
        p = np.empty((u.shape[0], k), np.float32)
        i = np.empty((u.shape[0], k), np.int32)
        self._check(self._lib.mv_topk(self._h, _ptr(u), u.shape[0], k, _ptr(p), _ptr(i)), "mv_topk")
        return p, i

    # -- resident corpus
    def corpus_upload(self, ids: np.ndarray, lens: np.ndarray):
        ids, lens = _as(ids, np.int32), _as(lens, np.int32)
        self._check(self._lib.mv_corpus_upload(self._h, _ptr(ids), _ptr(lens), ids.shape[0], ids.shape[1]), "mv_corpus_upload")
        self._corpus_n = ids.shape[0]

    def corpus_run(self, first: int, count: int, batch: int, keep_probs: bool = False, s_eff: int = 0):
        """Enqueue IRs [first, first+count) of the resident corpus in batches of `batch` (asynchronous).  s_eff > 0:
        process only the first s_eff tokens of each row (length-bucketed sweeps, see bucketed_sweep)."""
        self._check(self._lib.mv_corpus_run_len(self._h, first, count, batch, int(keep_probs), int(s_eff)), "mv_corpus_run_len")

    def bucketed_sweep(self, ids: np.ndarray, lens: np.ndarray, batch: int, with_probs: bool = False):
        """Score a ragged corpus with each batch padded to ITS longest member (the reference's pad-to-longest collation,
        predict_memory.py:97-101) instead of the corpus-wide S: rows are sorted by length, uploaded once, swept batch by
        batch at that batch's length (rounded up to 64 tokens), and the results are returned in the ORIGINAL order."""
        ids = np.ascontiguousarray(ids, np.int32)
        lens = np.ascontiguousarray(lens, np.int32)
        n = ids.shape[0]
        order = np.argsort(lens, kind="stable")
        self.corpus_upload(ids[order], lens[order])
        sl = lens[order]
        for s0 in range(0, n, batch):
            nb = min(batch, n - s0)
            self.corpus_run(s0, nb, nb, keep_probs=with_probs, s_eff=int(sl[s0 + nb - 1]))
        best, idx, ps = self.corpus_results(0, n, with_probs=with_probs)
        inv = np.empty(n, np.int64)
        inv[order] = np.arange(n)
        return best[inv], idx[inv], (ps[inv] if ps is not None else None)

    def corpus_results(self, first: int, count: int, with_probs: bool = False):
        best = np.empty((count, 2), np.float32)
        idx = np.empty((count,), np.int32)
        ps = np.empty((count, self.n_anchors), np.float32) if with_probs else None
        self._check(self._lib.mv_corpus_results(self._h, first, count, _ptr(best), _ptr(idx), _ptr(ps)), "mv_corpus_results")
        self._check_saturation()
        return best, idx, ps

    # -- multi-GPU exchange (RCCL bound inside the library; no torch in the process)
    def comm_prepare(self):
        """dlopen librccl.so and resolve its entry points (raises if RCCL is not usable in this process)."""
        self._check(self._lib.mv_comm_prepare(self._h), "mv_comm_prepare")

    def comm_unique_id(self) -> bytes:
        """Rank 0: the 128-byte ncclUniqueId every rank passes to comm_init."""
        buf = C.create_string_buffer(128)
        n = self._lib.mv_comm_unique_id(self._h, buf, 128)
        if n <= 0:
            self._check(n, "mv_comm_unique_id")
        return buf.raw[:n]

    def comm_init(self, rank: int, world: int, unique_id: Optional[bytes] = None):
        if unique_id is None:
            self._check(self._lib.mv_comm_init(self._h, int(rank), int(world), None, 0), "mv_comm_init")
        else:
            self._check(self._lib.mv_comm_init(self._h, int(rank), int(world), C.c_char_p(bytes(unique_id)), len(unique_id)), "mv_comm_init")
        self.comm_world = int(world)

    def comm_allgather(self, block: np.ndarray) -> np.ndarray:
        """Every rank contributes a same-shape array; returns ``[world, *block.shape]`` in rank order."""
        block = np.ascontiguousarray(block)
        world = getattr(self, "comm_world", 1)
        out = np.empty((world,) + block.shape, block.dtype)
        if block.nbytes:
            self._check(self._lib.mv_comm_allgather(self._h, _ptr(block), _ptr(out), block.nbytes), "mv_comm_allgather")
        return out

    def comm_destroy(self):
        self._check(self._lib.mv_comm_destroy(self._h), "mv_comm_destroy")
        self.comm_rank, self.comm_world = 0, 1

    def comm_info(self) -> Dict[str, int]:
        """What RCCL itself says about the live communicator: its rank count (0 = none), this rank, the RCCL version code, and
        the world mv_comm_allgather gathers over."""
        info = (C.c_int * 4)()
        self._check(self._lib.mv_comm_info(self._h, info, 4), "mv_comm_info")
        return {"rccl_ranks": int(info[0]), "rccl_rank": int(info[1]), "rccl_version": int(info[2]), "world": int(info[3])}

    # -- measurement / debug
    def set_streams(self, n: int):
        """Batches of the resident sweep in flight at once (1 or 2)."""
        self._check(self._lib.mv_set_streams(self._h, int(n)), "mv_set_streams")

    def profile_enable(self, on: bool = True):
        self._check(self._lib.mv_profile_enable(self._h, int(on)), "mv_profile_enable")

    def profile_select(self, names=None):
        """Restrict HIP-event recording to the named kernel classes (None = all)."""
        if names is None:
            mask = 0xFFFFFFFF
        else:
            all_names = [self._lib.mv_kernel_class_name(i).decode() for i in range(NUM_KERNEL_CLASSES)]
            mask = 0
            for nm in names:
                mask |= 1 << all_names.index(nm)
        self._check(self._lib.mv_profile_select(self._h, mask), "mv_profile_select")

    def profile_read(self) -> Dict[str, Tuple[float, int]]:
        ms = (C.c_double * NUM_KERNEL_CLASSES)()
        n = (C.c_int64 * NUM_KERNEL_CLASSES)()
        self._check(self._lib.mv_profile_read(self._h, ms, n, NUM_KERNEL_CLASSES), "mv_profile_read")
        return {self._lib.mv_kernel_class_name(i).decode(): (float(ms[i]), int(n[i])) for i in range(NUM_KERNEL_CLASSES)}

    def debug_encode(self, ids, lens, n_layers: int):
        ids, lens = _as(ids, np.int32), _as(lens, np.int32)
        self._check(self._lib.mv_debug_encode(self._h, _ptr(ids), _ptr(lens), ids.shape[0], ids.shape[1], n_layers), "mv_debug_encode")
        S = ids.shape[1]
        self._dbg = (ids.shape[0], (S + 63) // 64 * 64 if S <= 256 else (S + 127) // 128 * 128, lens.copy())  # engine.hip padded_len

    def debug_read(self, buffer: int) -> np.ndarray:
        B, Sp, lens = self._dbg
        shapes = {
            0: ((B, Sp, 768), np.float32), 1: ((B, Sp, 768), np.float16), 2: ((B, 12, Sp, 64), np.float16),
            3: ((B, 12, Sp, 64), np.float16), 4: ((B, 12, 64, Sp), np.float16), 5: ((B, Sp, 768), np.float16),
            6: ((B, Sp, 3072), np.float16), 7: ((B, self.P), np.float32),
        }
        shape, dt = shapes[buffer]
        out = np.empty(shape, dt)
        self._check(self._lib.mv_debug_read(self._h, buffer, _ptr(out), out.nbytes), "mv_debug_read")
        if getattr(self, "_precise", False) and buffer in self._TOKEN_AXIS:
            # MV_F16X8 keeps the LAST token of every sequence in row 1 and token 1 in the last token's row (the "special rows" of
            # misc_kernels.h embed_ln_kernel; only the [CLS] row's result leaves the encoder): hand the taps back in token order
            ax = self._TOKEN_AXIS[buffer]
            for b in range(B):
                n = int(lens[b])
                if n >= 3:
                    idx = [slice(None)] * out.ndim
                    idx[0] = b
                    i1, i2 = list(idx), list(idx)
                    i1[ax], i2[ax] = 1, n - 1
                    t = out[tuple(i1)].copy()
                    out[tuple(i1)] = out[tuple(i2)]
                    out[tuple(i2)] = t
        return out

    _TOKEN_AXIS = {0: 1, 1: 1, 2: 2, 3: 2, 4: 3, 5: 1, 6: 1}  # debug buffer -> its token axis

    def test_gemm(self, A16: np.ndarray, W16: np.ndarray, bias: Optional[np.ndarray], variant: int = 0, iters: int = 1):
        A16, W16 = _as(A16, np.float16), _as(W16, np.float16)
        M, K = A16.shape
        N = W16.shape[0]
        bias = None if bias is None else _as(bias, np.float32)
        out = np.empty((M, N), np.float32)
        ms = C.c_float(0)
        self._check(self._lib.mv_test_gemm(self._h, variant, M, N, K, _ptr(A16), _ptr(W16), _ptr(bias), _ptr(out), iters,
                                           C.byref(ms)), "mv_test_gemm")
        return out, float(ms.value)

    def test_gemm_pp(self, A: np.ndarray, W: np.ndarray, bias: np.ndarray, x8=False, iters: int = 1):
        """The persistent FFN-1 kernel on fp32 operands (unit row statistics): fp16 gelu(A W^T + bias) [M][N]; x8: the MV_F16X8
        build (True / 1: both correction terms, 2: the weight-side term A_hi8 W_lo8 only, the QKV projection's form), also returning
        the [lo8 | hi8] e4m3 planes of the output as uint8 [M][2 N]."""
        A, W, bias = _as(A, np.float32), _as(W, np.float32), _as(bias, np.float32)
        M, K = A.shape
        N = W.shape[0]
        out = np.empty((M, N), np.float16)
        out8 = np.empty((M, 2 * N), np.uint8) if x8 else None
        ms = C.c_float(0)
        self._check(self._lib.mv_test_gemm_pp(self._h, int(x8), M, N, K, _ptr(A), _ptr(W), _ptr(bias), _ptr(out), _ptr(out8), iters,
                                              C.byref(ms)), "mv_test_gemm_pp")
        return out, out8, float(ms.value)


def device_count() -> int:
    """GPUs visible to this process (0 without one)."""
    return max(0, int(load_library().mv_device_count()))


def e4m3_bits(x: np.ndarray) -> np.ndarray:
    """OCP e4m3fn bits of fp32 values through the library's host-side encoder (the one that builds the MV_F16X8 weight planes)."""
    lib = load_library()
    x = _as(x, np.float32)
    out = np.empty(x.shape, np.uint8)
    rc = lib.mv_test_e4m3(_ptr(x), _ptr(out), x.size)
    if rc != 0:
        raise RuntimeError("mv_test_e4m3 failed")
    return out


def e4m3_decode(b: np.ndarray) -> np.ndarray:
    """fp32 value of OCP e4m3fn bits (0x7f / 0xff = NaN)."""
    b = np.asarray(b, np.uint8).astype(np.int32)
    e, m = (b >> 3) & 15, b & 7
    v = np.where(e == 0, m * 2.0 ** -9, (8 + m) * 2.0 ** (e - 10))
    v = np.where((e == 15) & (m == 7), np.nan, v)
    return (np.where(b & 0x80, -v, v)).astype(np.float32)
