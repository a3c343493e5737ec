"""Round 6 diagnostic: the same sequence at positions 0..3 of one pass — which stage's tap differs between the copies?"""
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
names = {1: "x16 (stream hi)", 2: "q", 3: "k", 4: "vt", 5: "ctx", 6: "h16 (GELU out)", 0: "xres"}
for S, L in ((192, 170), (64, 50), (128, 100)):
    for form in ("1", "0"):
        eng = gu.engine_for(dk, wk, env={"MEMVUL_CLS_ASIDE": form}, **kw)
        ids1, _ = synth.make_ids(1, S, dims.vocab_size, seed=3 + S)
        lens = np.full((4,), L, np.int32)
        ids1[0, L:] = 0
        ids1[0, L - 1] = 102
        ids = np.repeat(ids1, 4, axis=0).astype(np.int32)
        for nl in (0, 1, 2):
            eng.debug_encode(ids, lens, nl)
            for buf in ((1,) if nl == 0 else (2, 3, 4, 5, 6, 1)):
                t = eng.debug_read(buf).astype(np.float32)
                ax = 3 if buf == 4 else (2 if buf in (2, 3) else 1)
                t = np.take(t, np.arange(L), axis=ax)  # the real tokens
                d = [float(np.abs(t[b] - t[0]).max()) for b in range(1, 4)]
                if any(d):
                    where = [np.unravel_index(int(np.argmax(np.abs(t[b] - t[0]))), t[0].shape) for b in range(1, 4)]
                    print("S %d form %s layers %d %-16s copies 1..3 vs copy 0: %s at %s" % (S, form, nl, names[buf], ["%.2e" % x for x in d], where), flush=True)
        print("S %d form %s done" % (S, form), flush=True)
