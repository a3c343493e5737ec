"""Round 5: two builds of the library must give the SAME BITS on the same inputs (the plane conversions without clamps / through v_fma_mix_f32 against
the clamped form: in range they are the same arithmetic).  Usage: python scripts/r05_bits_equal.py libA.so libB.so"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BODY = r'''
import sys, hashlib, numpy as np
sys.path.insert(0, %r)
from memvul_amd import synth
from memvul_amd.binding import Engine
dims = synth.BertDims(layers=12)
w = synth.make_weights(dims, qk_scale=2.0, match_scale=29.0, trained_like=True)
h = hashlib.sha256()
for (B, S, ragged) in ((16, 256, False), (7, 200, True), (4, 512, True)):
    ids, lens = synth.make_ids(B, S, dims.vocab_size, seed=5 + B, ragged=ragged, min_len=20)
    aids, alens = synth.make_ids(6, 128, dims.vocab_size, seed=9, ragged=True, min_len=8)
    e = Engine(0, vocab_size=dims.vocab_size, layers=12, max_tokens=16 * 512, max_batch=16, max_anchors=16)
    e.load_state_dict(w, "precise")
    e.anchor_append(aids, alens)
    o = e.forward(ids, lens, want_embed=True)
    h.update(o["logits"].tobytes()); h.update(o["embed"].tobytes())
    e.close()
print("DIGEST", h.hexdigest())
''' % ROOT
out = []
for lib in sys.argv[1:3]:
    env = dict(os.environ, MEMVUL_HIP_LIB=os.path.abspath(lib))
    r = subprocess.run([sys.executable, "-c", BODY], env=env, capture_output=True, text=True, timeout=600)
    d = [ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST")]
    if not d:
        print(lib, "FAILED", r.stderr[-1500:])
        sys.exit(1)
    out.append(d[0].split()[1])
    print(lib, out[-1][:16])
print("BITS EQUAL" if out[0] == out[1] else "BITS DIFFER")
sys.exit(0 if out[0] == out[1] else 2)
