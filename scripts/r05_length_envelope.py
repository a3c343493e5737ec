"""Round 5: the sequence-length axis of the precision envelope.  The full-batch trained-like test (tests/test_gpu_parity.py
::test_full_batch_trained_like_rows_against_the_oracle: 256-token issue reports against anchors of 8-64 tokens) reads 7.7e-4 in the precise mode where the
24 draws with anchors of 32-512 tokens read <= 4.8e-4: short sequences average the fp16 roundings of V and P over fewer keys.  This script measures it:
16 full-length sequences of L tokens, L = 8 .. 512, on the envelope model (seed 4001); per length the embedding error against the CPU reference
(tests/golden/r05_trained_like_refs.npz, scripts/r05_make_refs.py) and the logit error it causes at matcher scale 29 against 8 fixed issue-report embeddings
(the reference's on both sides, so that only the sequence under test contributes)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from memvul_amd import synth  # noqa: E402
from memvul_amd.binding import Engine  # noqa: E402
from oracle import memvul_oracle as orc  # noqa: E402  (checker only)
import r05_make_refs as mk  # noqa: E402

refs = np.load(mk.OUT)
dims = synth.BertDims(layers=12)
w = synth.make_weights(dims, seed=mk.ENV_SEED, qk_scale=2.0, match_scale=29.0, trained_like=True)
u_ref = refs["outlier_1_u"]
out = []
# (MEMVUL_SHORT_VLO is a development switch since round 6: only the -DMEMVUL_DEV_SWITCHES build reads it — Engine(dev=True))
for mode, env in (("precise", {}), ("precise, one V/P plane (MEMVUL_SHORT_VLO=0, development build)", {"MEMVUL_SHORT_VLO": "0"}), ("f16", {})):
    os.environ.pop("MEMVUL_SHORT_VLO", None)
    os.environ.update(env)
    e = Engine(0, vocab_size=dims.vocab_size, layers=12, max_tokens=16 * 512, max_batch=16, max_anchors=16, dev=bool(env))
    e.load_state_dict(w, "f16" if mode == "f16" else "precise")
    row = dict(mode=mode, by_length={})
    for L in mk.LENGTHS:
        if f"len_{L}" not in refs:
            continue
        _, ids, lens = mk.length_inputs(L)
        v = e.encode(ids, lens)
        v_ref = refs[f"len_{L}"]
        lg_g = orc.match(u_ref, v, w[synth.KEY_MATCH_W])[0]
        lg_r = orc.match(u_ref, v_ref, w[synth.KEY_MATCH_W])[0]
        row["by_length"][L] = dict(embed_err=float(np.abs(v - v_ref).max()), embed_err_rms=float(np.sqrt(((v - v_ref) ** 2).mean())),
                                   logit_err=float(np.abs(lg_g - lg_r).max()))
    e.close()
    out.append(row)
    print("%-46s " % mode + "  ".join("L=%d: embed %.1e logit %.2e" % (L, d["embed_err"], d["logit_err"]) for L, d in row["by_length"].items()), flush=True)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
