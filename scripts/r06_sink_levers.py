"""Round 6, CPU only: which lever restores the 1e-3 contract under attention sinks (float64 rounding model, oracle/precision_model.py).  For a sink
configuration: the storage floor of Q / K / V / P one tensor at a time, then engine forms with the candidate fixes — a second plane of V / P / Q / K through
attention (fp16 lo "f16x2", fp8 lo "f16x8"), the A-side terms in all three QKV blocks, the row terms for the [CLS] AND the [SEP] token.
Usage: python scripts/r06_sink_levers.py <token> <rows> <target> [seed] [group ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from memvul_amd import synth  # noqa: E402
from oracle import precision_model as pm  # noqa: E402

KW = dict(qk_scale=2.0, match_scale=29.0, trained_like=True)
L = 12
E, C = pm.X8_ENGINE, pm.X8_ENGINE_CLS
FORMS = {
    "floor": [
        ("floor q,k,v,p f16", pm.engine_formats(L, "exact", qkv="f16", p="f16"), {}),
        ("floor v only", pm.engine_formats(L, "exact", v="f16"), {}),
        ("floor p only", pm.engine_formats(L, "exact", p="f16"), {}),
        ("floor q,k only", pm.engine_formats(L, "exact", q="f16", k="f16"), {}),
        ("floor v,p as hi+lo8", pm.engine_formats(L, "exact", q="f16", k="f16", v="f16x8", p="f16x8"), {}),
        ("floor all four as hi+lo8", pm.engine_formats(L, "exact", qkv="f16x8", p="f16x8"), {}),
    ],
    "engine": [
        ("cls (shipped default)", pm.engine_formats(L, "f16", **C), dict(cls_fix=True)),
        ("both, qkv=q (CLS_ASIDE=0)", pm.engine_formats(L, "f16", **E), {}),
        ("both, qkv=qkv", pm.engine_formats(L, "f16", **dict(E, a_qkv="f16x8")), {}),
        ("both, qkv=qkv, v,p hi+lo8", pm.engine_formats(L, "f16", **dict(E, a_qkv="f16x8", v="f16x8", p="f16x8")), {}),
        ("both, qkv=qkv, q,k,v,p hi+lo8", pm.engine_formats(L, "f16", **dict(E, a_qkv="f16x8", qkv="f16x8", p="f16x8")), {}),
        ("both, qkv=qkv, q,k,v,p hi+lo16", pm.engine_formats(L, "f16", **dict(E, a_qkv="f16x8", qkv="f16x2", p="f16x2")), {}),
        ("cls, rows cls+sep", pm.engine_formats(L, "f16", **C), dict(cls_fix=True, special="cls+sep")),
        ("cls, rows cls+sep, v,p hi+lo8", pm.engine_formats(L, "f16", **dict(C, v="f16x8", p="f16x8")), dict(cls_fix=True, special="cls+sep")),
        ("cls, rows cls+sep, q,k,v,p hi+lo8", pm.engine_formats(L, "f16", **dict(C, qkv="f16x8", p="f16x8")), dict(cls_fix=True, special="cls+sep")),
        ("cls, rows cls+sep, qkv=qkv, q,k,v,p hi+lo8", pm.engine_formats(L, "f16", **dict(C, a_qkv="f16x8", qkv="f16x8", p="f16x8")), dict(cls_fix=True, special="cls+sep")),
        ("cls, rows cls only, q,k,v,p hi+lo8", pm.engine_formats(L, "f16", **dict(C, qkv="f16x8", p="f16x8")), dict(cls_fix=True)),
    ],
    # the candidate of round 6: row terms for the [CLS] and the [SEP] row (in the K / V blocks of the QKV projection too), V of those two rows as hi + lo
    "special": [
        ("cls (shipped default of round 5)", pm.engine_formats(L, "f16", **C), dict(cls_fix=True)),
        ("cls, rows cls+sep, V lo of those rows", pm.engine_formats(L, "f16", **C), dict(cls_fix=True, special="cls+sep", special_v="f16x2")),
        ("both (CLS_ASIDE=0), rows cls+sep in K,V blocks, V lo of those rows", pm.engine_formats(L, "f16", **E), dict(cls_fix=True, special="cls+sep", special_v="f16x2")),
        ("both, qkv=qkv", pm.engine_formats(L, "f16", **dict(E, a_qkv="f16x8")), {}),
    ],
}


def main():
    token, rows, target = sys.argv[1], sys.argv[2], float(sys.argv[3])
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 3001
    groups = sys.argv[5:] or ["floor", "engine"]
    dims = synth.BertDims(layers=L)
    ids, lens = synth.make_ids(4, 256, dims.vocab_size, seed=seed + 11)
    aids, alens = synth.make_ids(4, 512, dims.vocab_size, seed=seed + 23, ragged=True, min_len=32)
    LA = int(alens.max())
    aids = aids[:, :LA]
    sink = None
    if target > 0:
        g = synth.calibrate_sink(dims, seed, target, token, rows, n=3, **KW)
        sink = dict(token=token, rows=rows, gains=g)
    w = synth.make_weights(dims, seed=seed, sink=sink, **KW)
    m1, e1 = synth.sink_report(w, dims, ids, lens, token, rows)
    mask, amask = synth.mask_from_lens(lens, 256), synth.mask_from_lens(alens, LA)
    ref, u, v = pm.logits(w, ids, mask, aids, amask, None)
    print("# sink %s/%s/%.2f seed %d: achieved mass %.2f, eff keys of the [CLS] row %.1f, max |logit| %.2f, 16 logits; max / rms logit error"
          % (token, rows, target, seed, m1.mean(), e1.mean(), float(np.abs(ref).max())), flush=True)
    for grp in groups:
        for name, cfg, kw in FORMS[grp]:
            t0 = time.time()
            lg, _, _ = pm.logits(w, ids, mask, aids, amask, cfg, **kw)
            e = lg - ref
            print("%-46s %.2e / %.2e   (%.0f s)" % (name, float(np.abs(e).max()), float(np.sqrt((e ** 2).mean())), time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
