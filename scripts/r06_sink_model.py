"""Round 6 (VERDICT r5 next #1b), CPU only: the attention-concentration axis of the precision envelope in the float64 rounding model
(oracle/precision_model.py), BEFORE any GPU time is spent on it.  For every sink configuration (synth.apply_sink: which token collects the mass, for
which rows, how much of it) the trained-like logit error of the engine's forms:
  cls      the shipped default ([CLS]-row form: weight-side term everywhere, A-side term in the Q block + the [CLS] rows)
  both     MEMVUL_CLS_ASIDE=0 (both terms in every row; QKV: A-side in the Q block only)
  both+qkv MEMVUL_CLS_ASIDE=0 MEMVUL_QKV_ASIDE=qkv (both terms everywhere)
  cls+qkv  MEMVUL_QKV_ASIDE=qkv on the default form
  f16      MV_F16 (for scale)
and the floor the fp16 storage of Q / K / V / P leaves alone (GEMMs exact).
Usage: python scripts/r06_sink_model.py [--draws N] [--out profiles/r06_a_sink_model.txt]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from memvul_amd import synth  # noqa: E402
from oracle import precision_model as pm  # noqa: E402

KW = dict(qk_scale=2.0, match_scale=29.0, trained_like=True)
CONFIGS = [None] + [dict(token=t, rows=r, target=f) for t, r in (("sep", "cls"), ("sep", "all"), ("cls", "all")) for f in (0.5, 0.8, 0.95)]


def forms(L):
    return {
        "cls": (pm.engine_formats(L, "f16", **pm.X8_ENGINE_CLS), dict(cls_fix=True)),
        "both": (pm.engine_formats(L, "f16", **pm.X8_ENGINE), {}),
        "both+qkv": (pm.engine_formats(L, "f16", **dict(pm.X8_ENGINE, a_qkv="f16x8")), {}),
        "cls+qkv": (pm.engine_formats(L, "f16", **dict(pm.X8_ENGINE_CLS, a_qkv="f16x8")), dict(cls_fix=True)),
        "f16": (pm.engine_formats(L, "f16"), {}),
        "floor": (pm.engine_formats(L, "exact", qkv="f16", p="f16"), {}),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--draws", type=int, default=2)
    ap.add_argument("--out", default="")
    ap.add_argument("--forms", default="cls,both,both+qkv,cls+qkv,f16,floor")
    ap.add_argument("--configs", default="")
    args = ap.parse_args()
    dims = synth.BertDims(layers=12)
    out = open(args.out, "w") if args.out else None

    def emit(line):
        print(line, flush=True)
        if out:
            out.write(line + "\n")
            out.flush()

    emit("# float64 rounding model, 12 layers, trained-like weights (matcher x29), 3 issue reports x 256 tokens against 3 anchors of up to 512 tokens per draw")
    emit("# sink config | draw | achieved mass (IR / anchors, mean over layers) | eff. keys of the [CLS] row | max |logit| | max / rms logit error per form")
    want = [f for f in args.forms.split(",") if f]
    cfgs = CONFIGS if not args.configs else [CONFIGS[int(i)] for i in args.configs.split(",")]
    for cfg in cfgs:
        for d in range(args.draws):
            seed = 3001 + d
            ids, lens = synth.make_ids(3, 256, dims.vocab_size, seed=seed + 11)
            aids, alens = synth.make_ids(3, 512, dims.vocab_size, seed=seed + 23, ragged=True, min_len=32)
            LA = int(alens.max())
            aids = aids[:, :LA]
            sink = None
            tag = "none"
            rep = ""
            if cfg:
                g = synth.calibrate_sink(dims, seed, cfg["target"], cfg["token"], cfg["rows"], n=3, **KW)
                sink = dict(token=cfg["token"], rows=cfg["rows"], gains=g)
                tag = "%s/%s/%.2f" % (cfg["token"], cfg["rows"], cfg["target"])
            w = synth.make_weights(dims, seed=seed, sink=sink, **KW)
            m1, e1 = synth.sink_report(w, dims, ids, lens, *(sink["token"], sink["rows"]) if sink else ("sep", "cls"))
            m2, e2 = synth.sink_report(w, dims, aids, alens, *(sink["token"], sink["rows"]) if sink else ("sep", "cls"))
            rep = "mass %.2f / %.2f  eff keys %.1f / %.1f" % (m1.mean(), m2.mean(), e1.mean(), e2.mean())
            mask, amask = synth.mask_from_lens(lens, 256), synth.mask_from_lens(alens, LA)
            t0 = time.time()
            ref, _, _ = pm.logits(w, ids, mask, aids, amask, None)
            res = []
            for name in want:
                c, kw = forms(12)[name]
                lg, _, _ = pm.logits(w, ids, mask, aids, amask, c, **kw)
                e = lg - ref
                res.append("%s %.2e/%.2e" % (name, float(np.abs(e).max()), float(np.sqrt((e ** 2).mean()))))
            emit("%-14s seed %d | %s | max|logit| %.2f | %s | %.0fs" % (tag, seed, rep, float(np.abs(ref).max()), "  ".join(res), time.time() - t0))


if __name__ == "__main__":
    main()
