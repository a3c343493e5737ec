"""Round 6, CPU only: what the LOW PLANE of the residual stream is worth for the rows that are not special rows — before any kernel.  The residual GEMMs (output
projection, FFN-2) read and write the raw stream as two fp16 planes (hi + lo: 4 + 4 of the 9 bytes per element their epilogues move); the [CLS]-row argument (ordinary
rows reach the pooler only through attention, averaged over the keys) might let ordinary rows keep the hi plane alone.  Float64 rounding model of the shipped form with
  res        the stored stream of every row in that format: exact (= hi + lo, shipped) / f16x8 (hi + lo8: round 5's opt-in) / f16 (hi alone)
  special    the special rows ([CLS], [SEP]) keep hi + lo while the others follow `res`
over diffuse draws and draws with an attention sink (80 % of every row on [SEP]).
Usage: python scripts/r06_stream_model.py [--draws N]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from memvul_amd import synth  # noqa: E402
from oracle import precision_model as pm  # noqa: E402

KW = dict(qk_scale=2.0, match_scale=29.0, trained_like=True)
FORMS = [("shipped (hi + lo, every row)", "exact", None), ("hi + lo8, every row", "f16x8", None), ("hi + lo8, special rows hi + lo", "f16x8", "exact"),
         ("hi alone, every row", "f16", None), ("hi alone, special rows hi + lo", "f16", "exact")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--draws", type=int, default=4)
    args = ap.parse_args()
    dims = synth.BertDims(layers=12)
    res = {name: [] for name, _, _ in FORMS}
    for sink_cfg in (None, ("sep", "all", 0.8)):
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
            for name, fmt, sp in FORMS:
                cfg = pm.engine_formats(12, "f16", **pm.X8_ENGINE_CLS, res=fmt)
                lg, _, _ = pm.logits(w, ids, mask, aids[:, :LA], amask, cfg, res_special=sp, **pm.SHIPPED_KW)
                e = lg - ref
                res[name].append((float(np.abs(e).max()), float(np.sqrt((e ** 2).mean()))))
                line.append("%.2e" % res[name][-1][0])
            print("%s seed %d (max |logit| %.2f): %s" % ("sink sep/all/0.8" if sink_cfg else "diffuse", seed, float(np.abs(ref).max()), "  ".join(line)), flush=True)
    print("\n# over %d draws (%d diffuse + %d with the sink): max of the maxima / mean of the maxima / mean rms" % (2 * args.draws, args.draws, args.draws))
    for name, _, _ in FORMS:
        v = np.array(res[name])
        print("%-42s %.2e / %.2e / %.2e" % (name, v[:, 0].max(), v[:, 0].mean(), v[:, 1].mean()))


if __name__ == "__main__":
    main()
