"""Round 4: how the trained-like logit error of the two compute dtypes is DISTRIBUTED over independent draws (weights, issue reports, anchors by seed):
the maximum over one case moves +-40 % with the draw of the fp16 roundings, so one case is one sample.  Six 12-layer trained-like models
(synth.make_weights(trained_like=True, qk_scale=2, match_scale=29, seed=s)), 8 issue reports x 256 tokens against 6 anchors of up to 512 tokens each;
checker = oracle/hf_reference.py (fp32 torch CPU: the reference's graph).  torch first, then the engine."""
import os
import sys

import numpy as np
import torch  # noqa: F401  (before the engine: tests/test_gpu_parity.py)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from memvul_amd import synth  # noqa: E402
from memvul_amd.binding import Engine  # noqa: E402
from oracle.hf_reference import HFReference  # noqa: E402

dims = synth.BertDims(layers=12)
rows = []
for seed in range(3001, 3007):
    w = synth.make_weights(dims, seed=seed, qk_scale=2.0, match_scale=29.0, trained_like=True)
    ids, lens = synth.make_ids(8, 256, dims.vocab_size, seed=seed + 11, ragged=(seed % 2 == 0), min_len=40)
    aids, alens = synth.make_ids(6, 512, dims.vocab_size, seed=seed + 23, ragged=True, min_len=32)
    LA = int(alens.max())
    ref = HFReference(w, dims.as_dict(), threads=min(os.cpu_count() or 1, 16))
    v = ref.instance_forward(aids[:, :LA].astype(np.int64), synth.mask_from_lens(alens, LA))
    u, lg, p, best, idx = ref.predict(ids.astype(np.int64), synth.mask_from_lens(lens, 256), v)
    errs = {}
    for mode in ("precise", "f16"):
        e = Engine(0, vocab_size=dims.vocab_size, layers=12, max_tokens=16 * 512, max_batch=16, max_anchors=16)
        e.load_state_dict(w, mode)
        e.anchor_append(aids[:, :LA], alens)
        o = e.forward(ids, lens)
        errs[mode] = float(np.abs(o["logits"] - lg).max())
        e.close()
    rows.append((seed, float(np.abs(lg).max()), errs["precise"], errs["f16"]))
    print("seed %d: max |logit| %.2f  precise %.2e  f16 %.2e" % rows[-1], flush=True)
pr = np.array([r[2] for r in rows]); f = np.array([r[3] for r in rows])
print("precise: min %.2e  median %.2e  max %.2e   |  f16: min %.2e  median %.2e  max %.2e" % (pr.min(), np.median(pr), pr.max(), f.min(), np.median(f), f.max()))
