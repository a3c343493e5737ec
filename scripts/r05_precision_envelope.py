"""Round 5 (VERDICT r4 next #5 i): WHERE the 1e-3 logit contract of model_memory.py:141 holds — trained-like logit error of both compute
dtypes against matcher scale {10, 29, 60, 100} x LayerNorm-outlier magnitude {1x, 3x, 10x}.

The match logits are LINEAR in the matcher weight (logit = W_m [u; v; |u - v|], bias-free: model_memory.py:76,141), and synth's matcher of
scale s is s / 29 times the scale-29 one, so the logit error at scale s is the embedding error seen through a matcher s / 29 times larger:
err(s) = err(29) s / 29 up to fp32 rounding.  The sweep therefore runs ONE engine pass per (outlier magnitude, compute dtype) — embeddings
u, v from the engine at scale 29 — and evaluates the scales on the host from those embeddings with the oracle's matcher arithmetic
(oracle/memvul_oracle.py match, fp32) against the CPU reference's embeddings (tests/golden/r05_trained_like_refs.npz); the engine's own
logits at scale 29 and — one direct check — at scale 100 (a second engine with the x100 matcher loaded) are printed next to it.
Also reports mv_x8_saturation per case (the 10x outliers are meant to reach the fp8 planes' +-112 range)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from memvul_amd import synth  # noqa: E402
from memvul_amd.binding import Engine  # noqa: E402
from oracle import memvul_oracle as orc  # noqa: E402  (checker only)
import r05_make_refs as mk  # noqa: E402

SCALES = (10, 29, 60, 100)
refs = np.load(mk.OUT)
out = []
for k in mk.OUTLIERS:
    if f"outlier_{k}_lg" not in refs:
        print("no reference for outlier x%d" % k)
        continue
    u_ref, v_ref, lg_ref = refs[f"outlier_{k}_u"], refs[f"outlier_{k}_v"], refs[f"outlier_{k}_lg"]
    dims, ids, lens, aids, alens = mk.case_inputs(mk.ENV_SEED)
    w = synth.make_weights(dims, seed=mk.ENV_SEED, qk_scale=2.0, match_scale=29.0, trained_like=True, outlier_scale=float(k))
    LA = int(alens.max())
    for mode in ("precise", "f16"):
        e = Engine(0, vocab_size=dims.vocab_size, layers=12, max_tokens=16 * 512, max_batch=16, max_anchors=16)
        e.load_state_dict(w, mode)
        e.anchor_append(aids[:, :LA], alens)
        o = e.forward(ids, lens, want_embed=True)
        u, v = o["embed"], e.anchor_get()
        sat = e.x8_saturation() if mode == "precise" else 0
        e.close()
        row = dict(outlier_scale=k, mode=mode, x8_saturated=sat, u_err=float(np.abs(u - u_ref).max()), v_err=float(np.abs(v - v_ref).max()),
                   engine_logit_err_at_29=float(np.abs(o["logits"] - lg_ref).max()), max_abs_u=float(np.abs(u_ref).max()), by_scale={})
        for s in SCALES:
            wm = w[synth.KEY_MATCH_W] * np.float32(s / 29.0)
            lg_g = orc.match(u, v, wm)[0]
            lg_r = orc.match(u_ref, v_ref, wm)[0]
            row["by_scale"][s] = dict(max_abs_logit=float(np.abs(lg_r).max()), err=float(np.abs(lg_g - lg_r).max()))
        if k == 1:  # the one direct check of the linearity argument: an engine with the x100 matcher loaded
            w100 = dict(w)
            w100[synth.KEY_MATCH_W] = w[synth.KEY_MATCH_W] * np.float32(100.0 / 29.0)
            e = Engine(0, vocab_size=dims.vocab_size, layers=12, max_tokens=16 * 512, max_batch=16, max_anchors=16)
            e.load_state_dict(w100, mode)
            e.anchor_append(aids[:, :LA], alens)
            o100 = e.forward(ids, lens)
            e.close()
            row["engine_logit_err_at_100_direct"] = float(np.abs(o100["logits"] - orc.match(u_ref, v_ref, w100[synth.KEY_MATCH_W])[0]).max())
        out.append(row)
        print("outliers x%-2d %-7s clamped %-8d |u err| %.1e |v err| %.1e  " % (k, mode, sat, row["u_err"], row["v_err"]) +
              "  ".join("s=%d: |logit|<=%.1f err %.2e%s" % (s, d["max_abs_logit"], d["err"], "" if d["err"] <= 1e-3 else " (>1e-3)") for s, d in row["by_scale"].items()) +
              ("  | engine @29 %.2e" % row["engine_logit_err_at_29"]) +
              ("  engine @100 direct %.2e" % row["engine_logit_err_at_100_direct"] if "engine_logit_err_at_100_direct" in row else ""), flush=True)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
