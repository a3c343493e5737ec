"""Round 4: bench-scale determinism of the precise mode (the per-tile K-tile counts of the QKV sweep are new): the same 8 batches of 256 x 256 tokens
through mv_corpus_run five times with two batches in flight and once with one — every per-IR result must be bit-identical."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from memvul_amd import synth  # noqa: E402
from memvul_amd.binding import Engine  # noqa: E402

dims = synth.BertDims(layers=12)
w = synth.make_weights(dims, qk_scale=2.0, match_scale=29.0, trained_like=True)
B, S, G, NB = 256, 256, 124, 8
eng = Engine(0, vocab_size=dims.vocab_size, layers=12, max_tokens=B * S, max_batch=B, max_anchors=128)
eng.load_state_dict(w, "precise")
aids, alens = synth.make_ids(G, 512, dims.vocab_size, seed=synth.SEED + 1, ragged=True, min_len=32)
LA = int(alens.max())
eng.anchor_append(aids[:, :LA], alens)
ids, lens = synth.make_ids(NB * B, S, dims.vocab_size, seed=5)
eng.corpus_upload(ids, lens)
ref = None
for rep, streams in enumerate((2, 2, 2, 2, 2, 1)):
    eng.set_streams(streams)
    for i in range(NB):
        eng.corpus_run(i * B, B, B, keep_probs=True)
    best, idx, ps = eng.corpus_results(0, NB * B, with_probs=True)
    if ref is None:
        ref = (best.copy(), idx.copy(), ps.copy())
        print("first pass: best[0] =", best[0], "finite:", bool(np.isfinite(ps).all()))
    else:
        same = all(np.array_equal(a, b) for a, b in zip(ref, (best, idx, ps)))
        print(f"pass {rep} (streams {streams}): bit-identical = {same}")
        assert same
print("DETERMINISTIC_OK")
