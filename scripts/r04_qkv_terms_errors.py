"""Round-4 A/B on the GPU: trained-like logit error of the precise mode with the QKV projection sweeping both correction terms
(MEMVUL_QKV_ASIDE=qkv, round 3's form), the weight-side term only (none), or the A-side term in one block (q: shipped): the goldens l12_trained_*, the reference's own
12-layer run (ref12).  Checker = the committed fixtures (tests/golden)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import gpu_util as gu  # noqa: E402
import make_golden  # noqa: E402
import test_reference_pin as trp  # noqa: E402

for terms in ("qkv", "none", "q", "k", "v"):
    env = {"MEMVUL_QKV_ASIDE": terms}
    res = {}
    for name in ("l12_trained_s256", "l12_trained_ragged"):
        g = np.load(os.path.join(ROOT, "tests", "golden", f"{name}.npz"))
        dk, wk, B, S, ragged, G, SA = make_golden.CASES[name]
        eng = gu.engine_for(dk, wk, compute_dtype="precise", env=env, max_tokens=16384, max_batch=64, max_anchors=64)
        eng.anchor_reset()
        LA = int(g["anchor_lens"].max())
        eng.anchor_append(g["anchor_ids"][:, :LA], g["anchor_lens"])
        out = eng.forward(g["ids"], g["lens"])
        res[name] = float(np.abs(out["logits"] - g["logits"]).max())
    ref = trp.get_ref("ref12")
    aids, amask = trp._pad(ref["reader"]["golden"])
    ids, mask = trp._pad(ref["reader"]["test"])
    dk = dict(layers=ref["meta"]["layers"], vocab_size=ref["meta"]["vocab_size"])
    wk = dict(ref["meta"]["weight_kwargs"])
    eng = gu.engine_for(dk, wk, compute_dtype="precise", env=env, max_tokens=32 * 512, max_batch=32, max_anchors=16)
    eng.anchor_reset()
    eng.anchor_append(aids.astype(np.int32), amask.sum(1).astype(np.int32))
    out = eng.forward(ids.astype(np.int32), mask.sum(1).astype(np.int32))
    res["ref12"] = float(np.abs(out["logits"] - ref["logits"]).max())
    print("QKV blocks that sweep the A-side term too = %s:" % terms, {k: "%.2e" % v for k, v in res.items()}, flush=True)
