"""Round 5 (VERDICT r4 next #1 d): what each precision lever of the compliant mode costs and buys, on MANY draws instead of one.
Levers: which blocks of the QKV projection sweep the A-side correction term (MEMVUL_QKV_ASIDE = none / q / qv / qkv) x the residual stream's low part
(hi + lo fp16, or MEMVUL_STREAM_LO8=1: hi fp16 + lo8) x MEMVUL_CLS_ASIDE (the A-side term of the other GEMMs for the [CLS] rows alone).  For every combination: the trained-like logit error over the first N draws of
scripts/r05_make_refs.py (CPU references committed: tests/golden/r05_trained_like_refs.npz); rates come from bench.py (scripts/r05_visit5.sh).
Usage: python scripts/r05_margin_table.py [n_seeds] [out.json]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from memvul_amd import synth  # noqa: E402
from memvul_amd.binding import Engine  # noqa: E402
import r05_make_refs as mk  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
CONFIGS = [(a, s, "0") for s in ("0", "1") for a in ("none", "q", "qv", "qkv")]
if os.environ.get("R05_MARGIN_CONFIGS"):  # e.g. "v/0,v/1,qv/1,q/0": a second pass over other combinations; a third field = MEMVUL_CLS_ASIDE ("q/0/1")
    CONFIGS = [tuple((c.split("/") + ["0"])[:3]) for c in os.environ["R05_MARGIN_CONFIGS"].split(",")]
refs = np.load(mk.OUT)
errs = {c: [] for c in CONFIGS}
for seed in mk.SEEDS[:N]:
    lg = refs[f"seed_{seed}"]
    dims, ids, lens, aids, alens = mk.case_inputs(seed)
    w = synth.make_weights(dims, seed=seed, qk_scale=2.0, match_scale=29.0, trained_like=True)
    LA = int(alens.max())
    row = []
    for (aside, lo8, cls) in CONFIGS:
        os.environ["MEMVUL_QKV_ASIDE"] = aside
        os.environ["MEMVUL_STREAM_LO8"] = lo8
        os.environ["MEMVUL_CLS_ASIDE"] = cls
        e = Engine(0, vocab_size=dims.vocab_size, layers=12, max_tokens=16 * 512, max_batch=16, max_anchors=16)
        e.load_state_dict(w, "precise")
        e.anchor_append(aids[:, :LA], alens)
        o = e.forward(ids, lens)
        e.close()
        errs[(aside, lo8, cls)].append(float(np.abs(o["logits"] - lg).max()))
        row.append("%s/%s%s %.2e" % (aside, "lo8" if lo8 == "1" else "lo16", "/cls" if cls == "1" else "", errs[(aside, lo8, cls)][-1]))
    print("seed %d: " % seed + "  ".join(row), flush=True)
out = []
for c in CONFIGS:
    v = np.array(errs[c])
    out.append(dict(qkv_aside=c[0], stream="lo8" if c[1] == "1" else "lo16", cls_aside=c[2] == "1", draws=len(v), min=float(v.min()), median=float(np.median(v)),
                    p90=float(np.quantile(v, 0.9)), max=float(v.max()), rms_of_maxima=float(np.sqrt((v ** 2).mean()))))
    print("aside %-4s stream %-4s cls_aside %s over %d draws: min %.2e  median %.2e  p90 %.2e  max %.2e" % (c[0], out[-1]["stream"], c[2], len(v), v.min(), np.median(v), np.quantile(v, 0.9), v.max()))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
