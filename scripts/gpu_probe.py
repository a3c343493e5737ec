"""GPU probe: stand-alone GEMM variant timings at the production shapes + a per-kernel-class profile of
one hot-loop batch.  Writes gpurun_out/probe.json.  (Development aid; bench.py is the contract.)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from memvul_amd import synth  # noqa: E402
from memvul_amd.binding import Engine  # noqa: E402

out = {"gemm": [], "profile": {}}
dims = synth.BertDims(layers=2, vocab_size=4096)
w = synth.make_weights(dims)
eng = Engine(0, vocab_size=dims.vocab_size, layers=dims.layers, max_tokens=65536, max_batch=256, max_anchors=128)
eng.load_state_dict(w)

variants = [int(v) for v in os.environ.get("PROBE_VARIANTS", "10,21,22,23").split(",")]
M = int(os.environ.get("PROBE_M", 65536))
rng = np.random.default_rng(0)
for (N, K) in [(2304, 768), (768, 768), (3072, 768), (768, 3072)]:
    A = rng.standard_normal((M, K)).astype(np.float16)
    W = (rng.standard_normal((N, K)) * 0.05).astype(np.float16)
    for v in variants:
        try:
            _, ms = eng.test_gemm(A, W, None, variant=v, iters=10)
            tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
            out["gemm"].append({"variant": v, "M": M, "N": N, "K": K, "ms": ms, "tflops": tf})
            print(f"gemm v{v} M={M} N={N} K={K}: {ms:.4f} ms  {tf:.1f} TFLOP/s", flush=True)
        except Exception as e:  # noqa: BLE001
            print("gemm variant", v, "failed:", e, flush=True)

B, S, G = 256, 256, 124
ids, lens = synth.make_ids(B, S, dims.vocab_size)
eng.anchor_set(synth.make_anchor_bank(G))
eng.corpus_upload(ids, lens)
for _ in range(3):
    eng.corpus_run(0, B, B)
eng.sync()
eng.profile_enable(True)
eng.profile_read()
t0 = time.perf_counter()
for _ in range(5):
    eng.corpus_run(0, B, B)
eng.sync()
dt = (time.perf_counter() - t0) / 5
prof = eng.profile_read()
out["profile"] = {k: {"ms_total": v[0], "launches": v[1], "avg_us": (v[0] / v[1] * 1e3 if v[1] else 0)} for k, v in prof.items()}
out["ms_per_batch_2layers"] = dt * 1e3
print(json.dumps(out["profile"], indent=1))
print("2-layer batch ms:", dt * 1e3)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "probe.json"), "w"), indent=1)
