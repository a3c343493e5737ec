"""Round 5: CPU references (oracle/hf_reference.py, fp32 torch: the reference's graph) for the GPU-side error-distribution and
precision-envelope scripts (scripts/r05_error_distribution.py, scripts/r05_precision_envelope.py), computed ONCE on a CPU-only machine
and committed as a fixture (tests/golden/r05_trained_like_refs.npz) so that the GPU box spends its minutes on the engine, not on torch.

Cases (weights, issue reports, anchors all by seed; every one 12 layers, trained-like: LayerNorm outlier dims, peaked attention, matcher x29):
  seed_<s>          s = 3001 .. 3024: 8 issue reports x 256 tokens (every second seed ragged) against 6 anchors of up to 512 tokens
                    (3001 .. 3006 are round 4's six draws)                                     -> logits [8, 6, 2]
  outlier_<k>       k = 1, 3, 10: the same model family with the outlier offsets x k (seed 4001) -> u [8, 512], v [6, 512], logits
  len_<L>           L = 8, 16, 32, 64, 128, 256, 512: 16 full-length sequences of L tokens on the envelope model (seed 4001, outliers x1)
                    -> embeddings [16, 512]: the sequence-length axis of the envelope (short sequences average less in attention)
Usage: python scripts/r05_make_refs.py"""
import os
import sys

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from memvul_amd import synth  # noqa: E402
from oracle.hf_reference import HFReference  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "r05_trained_like_refs.npz")
SEEDS = list(range(3001, 3025))
OUTLIERS = (1, 3, 10)
ENV_SEED = 4001


def case_inputs(seed):
    dims = synth.BertDims(layers=12)
    ids, lens = synth.make_ids(8, 256, dims.vocab_size, seed=seed + 11, ragged=(seed % 2 == 0), min_len=40)
    aids, alens = synth.make_ids(6, 512, dims.vocab_size, seed=seed + 23, ragged=True, min_len=32)
    return dims, ids, lens, aids, alens


def reference(w, dims, ids, lens, aids, alens):
    LA = int(alens.max())
    ref = HFReference(w, dims.as_dict(), threads=min(os.cpu_count() or 1, 16))
    v = ref.instance_forward(aids[:, :LA].astype(np.int64), synth.mask_from_lens(alens, LA))
    u, lg, p, best, idx = ref.predict(ids.astype(np.int64), synth.mask_from_lens(lens, 256), v)
    return np.asarray(u, np.float32), np.asarray(v, np.float32), np.asarray(lg, np.float32)


def main():
    have = dict(np.load(OUT)) if os.path.exists(OUT) else {}
    for seed in SEEDS:
        if f"seed_{seed}" in have:
            continue
        dims, ids, lens, aids, alens = case_inputs(seed)
        w = synth.make_weights(dims, seed=seed, qk_scale=2.0, match_scale=29.0, trained_like=True)
        u, v, lg = reference(w, dims, ids, lens, aids, alens)
        have[f"seed_{seed}"] = lg
        print("seed %d: max |logit| %.2f" % (seed, float(np.abs(lg).max())), flush=True)
        np.savez_compressed(OUT, **have)
    for k in OUTLIERS:
        if f"outlier_{k}_lg" in have:
            continue
        dims, ids, lens, aids, alens = case_inputs(ENV_SEED)
        w = synth.make_weights(dims, seed=ENV_SEED, qk_scale=2.0, match_scale=29.0, trained_like=True, outlier_scale=float(k))
        u, v, lg = reference(w, dims, ids, lens, aids, alens)
        have[f"outlier_{k}_u"], have[f"outlier_{k}_v"], have[f"outlier_{k}_lg"] = u, v, lg
        print("outlier x%d: max |logit| %.2f  max |u| %.2f" % (k, float(np.abs(lg).max()), float(np.abs(u).max())), flush=True)
        np.savez_compressed(OUT, **have)


LENGTHS = (8, 16, 32, 64, 128, 192, 256, 384, 512)  # (192 and 384: round 6, the whole-pass [CLS]-row form)


def length_inputs(L):
    dims = synth.BertDims(layers=12)
    ids, lens = synth.make_ids(16, L, dims.vocab_size, seed=ENV_SEED + 100 + L)
    return dims, ids, lens


def main_lengths():
    have = dict(np.load(OUT))
    w = None
    for L in LENGTHS:
        if f"len_{L}" in have:
            continue
        dims, ids, lens = length_inputs(L)
        if w is None:
            w = synth.make_weights(dims, seed=ENV_SEED, qk_scale=2.0, match_scale=29.0, trained_like=True)
            ref = HFReference(w, dims.as_dict(), threads=min(os.cpu_count() or 1, 16))
        have[f"len_{L}"] = np.asarray(ref.instance_forward(ids.astype(np.int64), synth.mask_from_lens(lens, L)), np.float32)
        print("length %d: max |v| %.2f" % (L, float(np.abs(have[f"len_{L}"]).max())), flush=True)
        np.savez_compressed(OUT, **have)


if __name__ == "__main__":
    main()
    main_lengths()
