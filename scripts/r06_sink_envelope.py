"""Round 6 (VERDICT r5 next #1b), GPU side: the attention-concentration axis of the precision envelope.  For every case of tests/golden/r06_sink_refs.npz
(scripts/r06_make_sink_refs.py: trained-like 12-layer model with an attention sink of a given mass on the [SEP] / [CLS] / an ordinary token, CPU reference
logits) rebuild the weights from the stored gains, run the engine and report max |logit error| per (sink configuration, engine form):
  default        the shipped library (special rows: [CLS] + [SEP] row terms, V of those rows as hi + lo)
  cls_aside_0    MEMVUL_CLS_ASIDE=0 (both first-order terms in every row of the output projection / FFN GEMMs)
  qkv            MEMVUL_QKV_ASIDE=qkv (the A-side term in all three blocks of the QKV projection, every row)
  both_qkv       MEMVUL_CLS_ASIDE=0 MEMVUL_QKV_ASIDE=qkv (every term in every row: the most conservative form the library has)
Usage: python scripts/r06_sink_envelope.py [--alt-draws N] [--json out.json] [--only token_rows_pct]"""
import argparse
import json
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from memvul_amd.binding import Engine  # noqa: E402
import r06_make_sink_refs as mk6  # noqa: E402  (imports torch-free pieces only at call time... it imports torch: fine on the GPU box, before the engine)

FORMS = {"default": {}, "cls_aside_0": {"MEMVUL_CLS_ASIDE": "0"}, "qkv": {"MEMVUL_QKV_ASIDE": "qkv"},
         "both_qkv": {"MEMVUL_CLS_ASIDE": "0", "MEMVUL_QKV_ASIDE": "qkv"}}


def run(w, dims, ids, lens, aids, alens, env):
    saved = {k: os.environ.get(k) for k in ("MEMVUL_CLS_ASIDE", "MEMVUL_QKV_ASIDE")}
    for k in saved:
        os.environ.pop(k, None)
    os.environ.update(env)
    try:
        e = Engine(0, vocab_size=dims.vocab_size, layers=12, max_tokens=16 * 512, max_batch=16, max_anchors=16)
        e.load_state_dict(w, "precise")
        LA = int(alens.max())
        e.anchor_append(aids[:, :LA], alens)
        o = e.forward(ids, lens)
        sat = e.x8_saturation()
        e.close()
        return o["logits"], sat
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--alt-draws", type=int, default=3, help="draws per configuration that also run the non-default forms")
    ap.add_argument("--max-draws", type=int, default=24)
    ap.add_argument("--json", default="")
    ap.add_argument("--only", default="", help="restrict to configurations whose tag starts with this (e.g. sep_all_80)")
    ap.add_argument("--default-qkv", default="", help="run the 'default' column with MEMVUL_QKV_ASIDE set to this (an A/B of that switch over every draw)")
    args = ap.parse_args()
    if args.default_qkv:
        FORMS["default"] = {"MEMVUL_QKV_ASIDE": args.default_qkv}
    refs = np.load(mk6.OUT)
    cases = sorted(k[:-3] for k in refs.files if k.endswith("_lg"))
    rows = []
    pooled = {}  # the default form's signed per-logit errors, pooled over the draws with the sink on a delimiter token / on an ordinary token
    by_cfg = {}
    for t in cases:
        m = re.match(r"(\w+)_(\w+)_(\d+)_(\d+)$", t)
        token, rws, pct, seed = m.group(1), m.group(2), int(m.group(3)), int(m.group(4))
        cfg = "%s_%s_%02d" % (token, rws, pct)
        if args.only and not cfg.startswith(args.only):
            continue
        n = by_cfg.setdefault(cfg, 0)
        if n >= args.max_draws:
            continue
        by_cfg[cfg] = n + 1
        dims, w, ids, lens, aids, alens, _ = mk6.case(token, rws, pct / 100.0, seed, gains=refs[t + "_gains"])
        lg = refs[t + "_lg"]
        rec = dict(cfg=cfg, seed=seed, max_abs_logit=float(np.abs(lg).max()), stat=[float(x) for x in refs[t + "_stat"]])
        for form, env in FORMS.items():
            if form != "default" and n >= args.alt_draws:
                continue
            got, sat = run(w, dims, ids, lens, aids, alens, env)
            rec[form] = float(np.abs(got - lg).max())
            if form == "default":
                rec["x8_saturated"] = sat
                rec["rms"] = float(np.sqrt(((got - lg).astype(np.float64) ** 2).mean()))
                pooled.setdefault("ordinary" if token == "mid" else "delimiter", []).append((got - lg).ravel().astype(np.float64))
        rows.append(rec)
        print("%s seed %d: mass %.2f/%.2f eff keys %.1f/%.1f max|logit| %.2f | %s" % (
            cfg, seed, rec["stat"][0], rec["stat"][2], rec["stat"][1], rec["stat"][3], rec["max_abs_logit"],
            "  ".join("%s %.2e" % (f, rec[f]) for f in FORMS if f in rec)), flush=True)
    print("\n# per configuration: max |logit error| over the draws (count)")
    for cfg in sorted(by_cfg):
        line = [cfg]
        for form in FORMS:
            v = [r[form] for r in rows if r["cfg"] == cfg and form in r]
            if v:
                line.append("%s median %.2e max %.2e (%d)" % (form, float(np.median(v)), max(v), len(v)))
        print("  ".join(line))
    summary = {}
    for kind, v in sorted(pooled.items()):
        pe = np.concatenate(v)
        rms = float(np.sqrt((pe ** 2).mean()))
        qs = {q: float(np.quantile(np.abs(pe), q)) for q in (0.5, 0.9, 0.99, 0.999)}
        summary[kind] = dict(pooled_logits=len(pe), rms=rms, abs_quantiles=qs, max=float(np.abs(pe).max()))
        print("default form, sink on a %s token, all %d logits pooled: rms %.2e  |err| p50 %.2e  p90 %.2e  p99 %.2e  p99.9 %.2e  max %.2e  (1e-3 = %.1f rms)"
              % (kind, len(pe), rms, qs[0.5], qs[0.9], qs[0.99], qs[0.999], summary[kind]["max"], 1e-3 / rms))
    if args.json:
        json.dump(rows + [dict(pooled=summary)], open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
