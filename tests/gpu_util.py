"""Helpers shared by the GPU parity tests (the HIP path is reached only through the C-ABI binding)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from memvul_amd import synth  # noqa: E402
from memvul_amd import binding  # noqa: E402
from memvul_amd.binding import Engine  # noqa: E402

DIAG_DIR = os.path.join(ROOT, "gpurun_out")
_engines = {}
_weights = {}


def record(name, **kv):
    """Append a diagnostics record (max errors, timings) to gpurun_out/diag.jsonl so one GPU call yields
    numbers even for passing tests."""
    os.makedirs(DIAG_DIR, exist_ok=True)
    with open(os.path.join(DIAG_DIR, "diag.jsonl"), "a") as f:
        f.write(json.dumps({"name": name, **{k: (float(v) if isinstance(v, (np.floating, float)) else v) for k, v in kv.items()}}) + "\n")


def weights_for(dims_kw: dict, w_kw: dict):
    key = (tuple(sorted(dims_kw.items())), tuple(sorted(w_kw.items())))
    if key not in _weights:
        dims = synth.BertDims(**dims_kw)
        _weights[key] = (dims, synth.make_weights(dims, **w_kw))
    return _weights[key]


BOTH_DTYPES = ("precise", "f16")  # the product default first (binding.DEFAULT_COMPUTE), then the explicit opt-in


def engine_for(dims_kw: dict, w_kw: dict, gemm_tile: int = 0, env: dict = None, compute_dtype=None, **eng_kw) -> Engine:
    """compute_dtype: None = the PRODUCT default (binding.default_compute(): MV_F16X8 unless $MEMVUL_COMPUTE says otherwise), "f16" / 1 =
    MV_F16, "precise" / 6 = MV_F16X8.  env: MEMVUL_* switches read at mv_create, e.g. {"MEMVUL_CLS_PRUNE": "0"}.  gemm_tile: 0 = the engine's
    own choice by pass size, 128 = the small-pass kernels, 512 = the persistent kernels forced (MEMVUL_GEMM_TILE) — like every switch of
    binding.DEV_SWITCHES that is a development knob the product library does not read: an engine asked for one is created on the development
    build of the same sources (libmemvul_hip_dev.so, memvul_amd/build.py)."""
    env = dict(env or {})
    if gemm_tile:
        env["MEMVUL_GEMM_TILE"] = str(gemm_tile)
    if compute_dtype is None:
        compute_dtype = binding.default_compute()
    dev = any(k in binding.DEV_SWITCHES for k in env)
    key = (tuple(sorted(dims_kw.items())), tuple(sorted(w_kw.items())), tuple(sorted(eng_kw.items())), tuple(sorted(env.items())), compute_dtype)
    if key not in _engines:
        if len(_engines) >= 2:  # keep HBM use bounded: drop the oldest engine
            k0 = next(iter(_engines))
            _engines.pop(k0).close()
        dims, w = weights_for(dims_kw, w_kw)
        kw = dict(max_tokens=16384, max_batch=64, max_anchors=64)
        kw.update(eng_kw)
        switches = ("MEMVUL_CLS_PRUNE", "MEMVUL_STREAMS", "MEMVUL_QKV_ASIDE", "MEMVUL_CLS_ASIDE", "MEMVUL_CLS_ASIDE_MIN_LEN") + binding.DEV_SWITCHES
        old = {k: os.environ.get(k) for k in switches}
        for k in switches:
            os.environ.pop(k, None)
        os.environ.update(env)
        try:
            e = Engine(0, vocab_size=dims.vocab_size, layers=dims.layers, max_pos=dims.max_pos, dev=dev, **kw)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        e.load_state_dict(w, compute_dtype)
        _engines[key] = e
    return _engines[key]
