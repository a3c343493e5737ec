"""Round 6, CPU only: is the Q block's A-side term (MEMVUL_QKV_ASIDE=q, swept for EVERY row: 22 us per layer) still worth anything now that the special rows get it
from the row term?  Float64 rounding model of the shipped form with a_qkv = f16x8q (shipped) / f16x8w (no block sweeps the A-side term: the special rows' row term only)
over diffuse draws and draws with a 50 % [SEP] sink on every row (the cell of the sink envelope nearest the contract).
Usage: python scripts/r06_qkv_model.py [--draws N]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from memvul_amd import synth  # noqa: E402
from oracle import precision_model as pm  # noqa: E402

KW = dict(qk_scale=2.0, match_scale=29.0, trained_like=True)
FORMS = [("q (shipped)", "f16x8q"), ("none", "f16x8w"), ("qkv", "f16x8")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--draws", type=int, default=4)
    args = ap.parse_args()
    dims = synth.BertDims(layers=12)
    res = {name: [] for name, _ in FORMS}
    for sink_cfg in (None, ("sep", "all", 0.5)):
        for d in range(args.draws):
            seed = 3001 + d
            ids, lens = synth.make_ids(2, 256, dims.vocab_size, seed=seed + 11)
            aids, alens = synth.make_ids(2, 512, dims.vocab_size, seed=seed + 23, ragged=True, min_len=200)
            LA = int(alens.max())
            sink = None
            if sink_cfg:
                g = synth.calibrate_sink(dims, seed, sink_cfg[2], sink_cfg[0], sink_cfg[1], n=2, **KW)
                sink = dict(token=sink_cfg[0], rows=sink_cfg[1], gains=g)
            w = synth.make_weights(dims, seed=seed, sink=sink, **KW)
            mask, amask = synth.mask_from_lens(lens, 256), synth.mask_from_lens(alens, LA)
            ref, _, _ = pm.logits(w, ids, mask, aids[:, :LA], amask, None)
            line = []
            for name, fmt in FORMS:
                cfg = pm.engine_formats(12, "f16", **dict(pm.X8_ENGINE_SHIPPED, a_qkv=fmt))
                lg, _, _ = pm.logits(w, ids, mask, aids[:, :LA], amask, cfg, **pm.SHIPPED_KW)
                e = lg - ref
                res[name].append((float(np.abs(e).max()), float(np.sqrt((e ** 2).mean()))))
                line.append("%s %.2e" % (name.split(" ")[0], res[name][-1][0]))
            print("%s seed %d (max |logit| %.2f): %s" % ("sink sep/all/0.5" if sink_cfg else "diffuse", seed, float(np.abs(ref).max()), "  ".join(line)), flush=True)
    print("\n# over %d draws: max of the maxima / mean of the maxima / mean rms" % (2 * args.draws))
    for name, _ in FORMS:
        v = np.array(res[name])
        print("%-14s %.2e / %.2e / %.2e" % (name, v[:, 0].max(), v[:, 0].mean(), v[:, 1].mean()))


if __name__ == "__main__":
    main()
