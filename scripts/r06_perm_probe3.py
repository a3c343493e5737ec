"""Round 6 diagnostic: how many elements of each stage tap differ between the copy at in-tile offset 0 and the one at offset 192 (S 192), per token block."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from memvul_amd import synth  # noqa: E402
import gpu_util as gu  # noqa: E402

dk, wk = dict(layers=2, vocab_size=2048), dict(qk_scale=2.0, match_scale=29.0, trained_like=True)
dims, w = gu.weights_for(dk, wk)
kw = dict(max_tokens=16384, max_batch=32, max_anchors=8)
S, L = 192, 192
for compute, env in (("precise", {}), ("precise", {"MEMVUL_QKV_ASIDE": "qkv"}), ("f16", {})):
    eng = gu.engine_for(dk, wk, compute_dtype=compute, env=env, **kw)
    ids1, _ = synth.make_ids(1, S, dims.vocab_size, seed=3 + S)
    lens = np.full((4,), L, np.int32)
    ids = np.repeat(ids1, 4, axis=0).astype(np.int32)
    eng.debug_encode(ids, lens, 1)
    for buf, name in ((2, "q"), (3, "k"), (5, "ctx"), (6, "h16"), (1, "x16")):
        t = eng.debug_read(buf).astype(np.float32)
        if buf in (2, 3):
            t = t.transpose(0, 2, 1, 3).reshape(4, S, 768)
        d = t[1] != t[0]
        per16 = d.reshape(S // 16, 16, -1).mean(axis=(1, 2))
        print("%s %s %-4s differing fraction %.2e; per 16-token block: %s; 2 vs 0: %d, 3 vs 1: %d" % (
            compute, env, name, d.mean(), " ".join("%.0e" % x for x in per16), int((t[2] != t[0]).sum()), int((t[3] != t[1]).sum())), flush=True)
