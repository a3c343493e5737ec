"""Round 6 diagnostic: is a row's embedding independent of its position in the pass at padded lengths 192 / 384 (a 256-row tile spans two sequences there)?"""
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
kw = dict(compute_dtype="precise", max_tokens=16384, max_batch=32, max_anchors=8)
for S in (192, 320, 128, 64):
    for form in ("1", "0"):
        eng = gu.engine_for(dk, wk, env={"MEMVUL_CLS_ASIDE": form}, **kw)
        for B in (7, 8):
            ids, lens = synth.make_ids(B, S, dims.vocab_size, seed=11 + S, ragged=True, min_len=max(S // 2 + 2, S - 60))
            ids = (ids * (np.arange(S)[None, :] < lens[:, None])).astype(np.int32)
            a = eng.encode(ids, lens)
            for layers in (0, 1, 2):
                pass
            perm = np.random.default_rng(S).permutation(B)
            p = eng.encode(ids[perm], lens[perm])
            d = np.abs(p - a[perm]).max(axis=1)
            print("S %d CLS_ASIDE %s B %d perm %s: rows that differ (position in the permuted batch: max diff) %s" % (
                S, form, B, perm.tolist(), {int(i): float("%.2e" % d[i]) for i in np.flatnonzero(d > 0)}), flush=True)
            one = np.stack([eng.encode(ids[i:i + 1], lens[i:i + 1])[0] for i in range(B)])
            d1 = np.abs(one - a).max(axis=1)
            print("   alone vs in the batch: %s" % {int(i): float("%.2e" % d1[i]) for i in np.flatnonzero(d1 > 0)}, flush=True)
