"""Round 6, CPU only: one model-side experiment on the 300 us per layer every row still pays — the weight-side half-sweeps A_hi8 W_lo8^T —
BEFORE any kernel.  Float64 rounding model (oracle/precision_model.py) of the shipped form (special rows; weight-side term in every GEMM and every row) with the
weight-side term replaced by cheaper forms:
  fp4       both operands MX-fp4 (e2m1, a scale per 32 K-elements): 4x the fp16 matrix rate, half the bytes of the planes
  sparse    W_lo8 pruned 2:4 along K: the sparse matrix path at 2x the dense rate, half the W_lo8 bytes
  tophalf   the term only in the half of each matrix' 128-wide K-tiles that ranks highest in ||W_lo[:, tile]|| rms(A[.., tile]) (static choice per matrix)
  none      no weight-side term at all (scale);  none@w_1: none in FFN-1 only (what (c) — FFN-1's W'' as fp16 + a rank-r correction — has to recover)
over 4 diffuse draws and 4 draws with an attention sink (80 % of every row on [SEP]).  Also: how much of ||W_lo||_F^2 a rank-r approximation holds (c).
Usage: python scripts/r06_wside_model.py [--draws N]"""
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
FORMS = [("engine (e4m3 x e4m3, every K-tile)", None, None), ("fp4 (MX e2m1 both operands)", "fp4", None), ("sparse 2:4 W_lo8", "sparse", None),
         ("tophalf of the K-tiles", "tophalf", None), ("none", "none", None), ("none in FFN-1 only", "none", {"w_1"}), ("fp4 in FFN-1 and FFN-2 only", "fp4", {"w_1", "w_2"})]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--draws", type=int, default=4)
    args = ap.parse_args()
    dims = synth.BertDims(layers=12)
    cfg = pm.engine_formats(12, "f16", **pm.X8_ENGINE_CLS)
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
            pm.W_SIDE_MODE, pm.W_SIDE_ONLY = None, None
            ref, _, _ = pm.logits(w, ids, mask, aids[:, :LA], amask, None)
            line = []
            for name, mode, only in FORMS:
                pm.W_SIDE_MODE, pm.W_SIDE_ONLY = mode, only
                t0 = time.time()
                lg, _, _ = pm.logits(w, ids, mask, aids[:, :LA], amask, cfg, **pm.SHIPPED_KW)
                e = lg - ref
                res[name].append((float(np.abs(e).max()), float(np.sqrt((e ** 2).mean()))))
                line.append("%s %.2e" % (name.split(" ")[0] + ("@" + ",".join(sorted(only)) if only else ""), res[name][-1][0]))
            pm.W_SIDE_MODE, pm.W_SIDE_ONLY = None, None
            print("%s seed %d (max |logit| %.2f): %s" % ("sink sep/all/0.8" if sink_cfg else "diffuse", seed, float(np.abs(ref).max()), "  ".join(line)), flush=True)
            if d == 0 and sink_cfg is None:  # (c): is W_lo low-rank?
                W = w[pm.PFX + "encoder.layer.5.intermediate.dense.weight"].astype(np.float64)
                lo = W - W.astype(np.float16).astype(np.float64)
                sv = np.linalg.svd(lo, compute_uv=False)
                en = np.cumsum(sv ** 2) / (sv ** 2).sum()
                print("   (c) rank-r share of ||W_lo||_F^2 of FFN-1 layer 5 [3072 x 768]: r = 16: %.3f  64: %.3f  256: %.3f  (a flat spectrum: rounding noise has no low-rank part)" % (en[15], en[63], en[255]), flush=True)
    print("\n# over %d draws (%d diffuse + %d with the sink): max of the maxima / mean of the maxima / mean rms" % (2 * args.draws, args.draws, args.draws))
    for name, _, only in FORMS:
        v = np.array(res[name])
        print("%-42s %.2e / %.2e / %.2e" % (name, v[:, 0].max(), v[:, 0].mean(), v[:, 1].mean()))


if __name__ == "__main__":
    main()
