"""Probe: two engines on one GPU, each sized for half the CUs, running two batches concurrently on their own streams,
against one engine on the whole chip.  (The persistent GEMM workgroups of one launch run in lockstep, so their memory
phases hit HBM in bursts; two independent pipelines on disjoint CU sets interleave them.)  Prints IR/s of both set-ups."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from memvul_amd import synth  # noqa: E402
from memvul_amd.binding import Engine  # noqa: E402

B, S, G, STEPS, WARM = 256, 256, 124, 24, 4
dims = synth.BertDims(layers=12)
w = synth.make_weights(dims)
ids, lens = synth.make_ids(8 * B, S, dims.vocab_size)
anchors = synth.make_anchor_bank(G)


def make(ncu):
    if ncu:
        os.environ["MEMVUL_NUM_CU"] = str(ncu)
    else:
        os.environ.pop("MEMVUL_NUM_CU", None)
    e = Engine(0, vocab_size=dims.vocab_size, layers=dims.layers, max_tokens=B * S, max_batch=B, max_anchors=128, dev=True)  # MEMVUL_NUM_CU: a development switch
    e.load_state_dict(w)
    e.anchor_set(anchors)
    e.corpus_upload(ids, lens)
    return e


def run(engs, batch):
    per = B // batch if batch < B else 1
    for i in range(WARM):
        for e in engs:
            e.corpus_run((i % 8) * B, batch, batch)
    for e in engs:
        e.sync()
    t0 = time.perf_counter()
    for i in range(STEPS):
        for e in engs:
            e.corpus_run((i % 8) * B, batch, batch)
    for e in engs:
        e.sync()
    dt = time.perf_counter() - t0
    return STEPS * len(engs) * batch / dt


os.environ["MEMVUL_STREAMS"] = "1"  # each engine runs one batch at a time; the engines supply the concurrency
one = make(0)
print("one engine, 256 CUs, B=256:", round(run([one], 256), 1), "IR/s", flush=True)
one.close()
for n in (2, 3, 4):
    engs = [make(0) for _ in range(n)]
    print(f"{n} engines, full-size grids, B=256 each:", round(run(engs, 256), 1), "IR/s", flush=True)
    for e in engs:
        e.close()
for ncu in (128,):
    a, b = make(ncu), make(ncu)
    print(f"two engines, {ncu} CUs each, B=256 each:", round(run([a, b], 256), 1), "IR/s", flush=True)
    a.close(); b.close()
