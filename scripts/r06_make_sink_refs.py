"""Round 6 (VERDICT r5 next #1b): CPU references for the attention-concentration axis of the precision envelope, computed ONCE on a CPU-only machine and
committed (tests/golden/r06_sink_refs.npz) so that the GPU box spends its minutes on the engine (scripts/r06_sink_envelope.py is the GPU side).

Every case is the trained-like 12-layer model of the round-5 error distribution (scripts/r05_make_refs.py: 8 issue reports x 256 tokens against 6 anchors of up
to 512 tokens, weights and inputs by seed) with an attention SINK written into the weights (synth.apply_sink): every head of every layer puts `target` of the
attention mass of the [CLS] row (rows = "cls") or of every row (rows = "all") on ONE token — the sequence's [SEP] (token = "sep"), its [CLS] ("cls"), or an
ordinary token in the middle of the sequence ("mid": the regime the engine's special rows do NOT cover).  Stored per case: the calibrated per-layer gains (the GPU
side rebuilds the very same weights from them), the achieved mass / effective number of keys of the [CLS] row on the CPU oracle, and the reference logits
(oracle/hf_reference.py: HF BertModel fp32 + the reference's head).
Usage: python scripts/r06_make_sink_refs.py [--draws N]"""
import argparse
import os
import sys

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from memvul_amd import synth  # noqa: E402
import r05_make_refs as mk  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "r06_sink_refs.npz")
KW = dict(qk_scale=2.0, match_scale=29.0, trained_like=True)
# (token, rows, target): the six cells the verdict names on [SEP], the [CLS]-token sink, and the ordinary-token sink
CONFIGS = [("sep", r, f) for r in ("cls", "all") for f in (0.5, 0.8, 0.95)] + [("cls", "all", 0.8), ("mid", "all", 0.5), ("mid", "all", 0.8), ("mid", "cls", 0.8)]
SEEDS = list(range(3001, 3025))
DRAWS = {("sep", "all", 0.8): 12}  # every other cell: --draws


def tag(token, rows, target, seed):
    return "%s_%s_%02d_%d" % (token, rows, round(target * 100), seed)


def case(token, rows, target, seed, gains=None):
    """(dims, weights, ids, lens, aids, alens, gains) of one case; gains None = calibrate here."""
    dims, ids, lens, aids, alens = mk.case_inputs(seed)
    if token == "mid":
        ids, aids = synth.mark_mid_token(ids, lens), synth.mark_mid_token(aids, alens)
    if gains is None:
        gains = synth.calibrate_sink(dims, seed, target, token, rows, n=3, **KW)
    w = synth.make_weights(dims, seed=seed, sink=dict(token=token, rows=rows, gains=list(gains)), **KW)
    return dims, w, ids, lens, aids, alens, gains


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--draws", type=int, default=8)
    args = ap.parse_args()
    have = dict(np.load(OUT)) if os.path.exists(OUT) else {}
    for token, rows, target in CONFIGS:
        for seed in SEEDS[:DRAWS.get((token, rows, target), args.draws)]:
            t = tag(token, rows, target, seed)
            if t + "_lg" in have:
                continue
            dims, w, ids, lens, aids, alens, gains = case(token, rows, target, seed)
            m1, e1 = synth.sink_report(w, dims, ids[:3], lens[:3], token, rows)
            m2, e2 = synth.sink_report(w, dims, aids[:2, :int(alens[:2].max())], alens[:2], token, rows)
            u, v, lg = mk.reference(w, dims, ids, lens, aids, alens)
            have[t + "_lg"], have[t + "_gains"] = lg, np.asarray(gains, np.float32)
            have[t + "_stat"] = np.array([m1.mean(), e1.mean(), m2.mean(), e2.mean()], np.float32)
            print("%s: mass %.2f / %.2f (issue reports / anchors)  eff. keys of the [CLS] row %.1f / %.1f  max |logit| %.2f" % (
                t, m1.mean(), m2.mean(), e1.mean(), e2.mean(), float(np.abs(lg).max())), flush=True)
            np.savez_compressed(OUT, **have)


if __name__ == "__main__":
    main()
