"""Generate the golden vectors under tests/golden/ (run in the build container; committed).

Golden outputs come from ``oracle/hf_reference.py`` — the HuggingFace/torch modules the
reference delegates its arithmetic to (model_memory.py:64,70,73; custom_PTM_embedder.py:99,228),
fp32, eager attention, CPU — on seeded synthetic weights/inputs from ``memvul_amd/synth.py``.
They pin ``oracle/memvul_oracle.py`` (tests/test_oracle_golden.py) and are the committed
fixtures the GPU parity tests compare the HIP path against.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from memvul_amd import synth  # noqa: E402
from oracle.hf_reference import HFReference  # noqa: E402

CASES = {
    # name: (dims kwargs, weight kwargs, B, S, ragged, G, S_anchor)
    "l2_peaky_full": (dict(layers=2, vocab_size=2048), dict(qk_scale=6.0), 4, 64, False, 8, 96),
    "l2_ragged": (dict(layers=2, vocab_size=2048), dict(qk_scale=3.0, match_scale=8.0), 6, 128, True, 5, 64),
    "l12_base_ragged": (dict(layers=12), dict(), 3, 128, True, 6, 160),
    "l12_base_s256": (dict(layers=12), dict(), 2, 256, False, 4, 256),
    # trained-like regime: 12 layers, peaked attention, stable outlier dimensions in every LayerNorm,
    # |u| = O(1), matcher scaled so that max |logit| ~ 3 (training temperature 0.1, config_memory.json:38)
    "l12_trained_s256": (dict(layers=12), dict(qk_scale=2.0, match_scale=29.0, trained_like=True), 4, 256, False, 8, 320),
    "l12_trained_ragged": (dict(layers=12), dict(qk_scale=2.0, match_scale=29.0, trained_like=True), 6, 256, True, 6, 512),
}


def case_inputs(name):
    dk, wk, B, S, ragged, G, SA = CASES[name]
    dims = synth.BertDims(**dk)
    w = synth.make_weights(dims, **wk)
    ids, lens = synth.make_ids(B, S, dims.vocab_size, ragged=ragged)
    aids, alens = synth.make_ids(G, SA, dims.vocab_size, seed=synth.SEED + 1, ragged=True, min_len=8)
    return dims, w, ids, lens, aids, alens


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    only = sys.argv[1:]
    for name in CASES:
        if only and name not in only:
            continue
        dims, w, ids, lens, aids, alens = case_inputs(name)
        ref = HFReference(w, dims.as_dict())
        # anchor bank: one chunk (<128 anchors), padded to its longest member (predict_memory.py:81)
        LA = int(alens.max())
        v = ref.instance_forward(aids[:, :LA], synth.mask_from_lens(alens, LA))
        mask = synth.mask_from_lens(lens, ids.shape[1])
        u, hidden = ref.instance_forward(ids, mask, all_hidden=True)
        logits, p, best, idx = ref.match(u, v, same_idx=0)
        cls_per_layer = np.stack([h[:, 0] for h in hidden], 0)  # [L+1, B, H]
        np.savez_compressed(
            os.path.join(out_dir, f"{name}.npz"),
            ids=ids, lens=lens, anchor_ids=aids, anchor_lens=alens,
            u=u, v=v, logits=logits, p=p, best=best, idx=idx,
            cls_per_layer=cls_per_layer, hidden_last_row0=hidden[-1][0],
        )
        print(name, "logits absmax", float(np.abs(logits).max()), "u absmax", float(np.abs(u).max()))


if __name__ == "__main__":
    main()
