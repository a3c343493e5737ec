"""Round 6: Engine.forward on pad-to-longest batches of 512 unsorted reports against Engine.forward_by_length at several `min_tokens`, engine time only
(12 layers, 124 anchors, precise), for two length distributions: the e2e probe's (20 .. 256 tokens, mean ~190) and a long-tailed one (16 .. 512, mean ~200).
Usage (GPU box): python scripts/r06_by_length_probe.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from memvul_amd import synth  # noqa: E402
from memvul_amd.binding import Engine  # noqa: E402

dims = synth.BertDims(layers=12)
w = synth.make_weights(dims, seed=1)
e = Engine(0, vocab_size=dims.vocab_size, layers=12, max_tokens=512 * 512, max_batch=512, max_anchors=128)
e.load_state_dict(w, "precise")
e.anchor_set(synth.make_anchor_bank(124))
rng = np.random.default_rng(3)
NB = 12
for name, S, draw in (("e2e-like 20..256", 256, lambda n: np.minimum(256, rng.integers(20, 330, n))), ("long tail 16..512", 512, lambda n: np.minimum(512, (16 + rng.gamma(2.0, 95.0, n)).astype(np.int64)))):
    batches = []
    for _ in range(NB):
        lens = draw(512).astype(np.int32)
        L = int(lens.max())
        ids = rng.integers(1000, dims.vocab_size, (512, L)).astype(np.int32) * (np.arange(L)[None, :] < lens[:, None])
        batches.append((ids.astype(np.int32), lens))
    mean = float(np.mean([b[1].mean() for b in batches]))
    res = []
    for label, f in [("padded forward", lambda i, l: e.forward(i, l, want_logits=False))] + [
            ("by length, min_tokens %d" % mt, (lambda mt: lambda i, l: e.forward_by_length(i, l, want_logits=False, min_tokens=mt))(mt)) for mt in (8192, 16384, 32768, 65536)]:
        f(*batches[0])
        t0 = time.perf_counter()
        for i, l in batches:
            f(i, l)
        dt = time.perf_counter() - t0
        res.append("%s: %.0f IR/s" % (label, NB * 512 / dt))
    print("%s (mean %.0f tokens, pad-to-longest %d): %s" % (name, mean, S, " | ".join(res)), flush=True)
