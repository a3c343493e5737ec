"""INTEGRATION.md §B is EXECUTED, not prose: the ctypes stub a maintainer would paste into
MemVul/model_memory.py (model_memory.py:105-147 replaced by mv_anchor_append / mv_forward) is extracted from the markdown and

  * (CPU) compiled, its `_MvConfig` compared field for field with the binding's `struct mv_config`, and every `_mv.mv_*` entry
    point it calls looked up among the library's exports;
  * (GPU) run as written in a fresh process — torch first, `libmemvul_hip.so` found through LD_LIBRARY_PATH as the bare
    `C.CDLL("libmemvul_hip.so")` of the stub needs — behind a stand-in for AllenNLP's `Model` base that only provides
    `state_dict()` / `device`, on a synthetic state dict loaded through `mv_load_tensor` exactly as the stub does it, and
    checked against memvul_amd.binding.Engine on the same weights and inputs, bit for bit.
"""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stub_source() -> str:
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = md[md.index("## B. Patch the reference's own `ModelMemory`"):]
    m = re.search(r"```python\n(.*?)```", sec, re.S)
    assert m, "INTEGRATION.md §B lost its code block"
    return m.group(1)


def test_stub_compiles_and_matches_the_abi():
    import ctypes as C

    from memvul_amd import binding

    src = stub_source()
    compile(src, "INTEGRATION.md#B", "exec")
    used = set(re.findall(r"_mv\.(mv_\w+)", src))
    assert used and used <= set(binding.ABI_SYMBOLS), used - set(binding.ABI_SYMBOLS)
    lib = binding.load_library()
    for name in used:
        assert hasattr(lib, name)
    # the structure definition alone (everything up to the CDLL line) is executable without a GPU
    head = src[:src.index("_mv = C.CDLL")]
    ns = {}
    exec(head, ns)
    assert [(n, t) for n, t in ns["_MvConfig"]._fields_] == [(n, t) for n, t in binding.MvConfig._fields_]
    assert C.sizeof(ns["_MvConfig"]) == C.sizeof(binding.MvConfig)


DRIVER = r'''
import sys, os
import numpy as np
import torch                                   # BEFORE the engine library (INTEGRATION.md: load order)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from memvul_amd import synth
from test_integration_stub import stub_source

dims = synth.BertDims(layers=2, vocab_size=2048)
w = synth.make_weights(dims, qk_scale=3.0)

class Model:                                   # stand-in for allennlp.models.Model: what the stub touches of it
    def __init__(self):
        self.device = torch.device("cuda", 0)
        self._same_idx = 0
        self._golden_instances_labels = None
    def state_dict(self):
        return {k: torch.from_numpy(v.copy()) for k, v in w.items()}

ns = {"Model": Model, "torch": torch}
exec(stub_source(), ns)                        # the markdown block, verbatim
m = ns["ModelMemory"]()
m._mv_init()
aids, alens = synth.make_ids(5, 96, dims.vocab_size, seed=synth.SEED + 1, ragged=True, min_len=8)
ids, lens = synth.make_ids(7, 128, dims.vocab_size, ragged=True, min_len=5)
tf = lambda i, l: {"tokens": {"token_ids": torch.from_numpy(i.astype(np.int64)), "mask": torch.from_numpy(np.arange(i.shape[1])[None, :] < l[:, None])}}
m.forward_gold_instances(tf(aids, alens), [{"instance": [{"label": "CWE-%d" % g}]} for g in range(5)])
probs, best = m._match(tf(ids, lens))
assert m._golden_instances_labels == ["CWE-%d" % g for g in range(5)]

from memvul_amd.binding import Engine          # the same library through the repository's own binding
e = Engine(0, vocab_size=dims.vocab_size, layers=dims.layers, max_tokens=128 * 512, max_batch=512, max_anchors=1024)
e.load_state_dict(w)
e.anchor_append(aids, alens)
o = e.forward(ids, lens)
assert np.array_equal(probs, o["probs"]) and np.array_equal(best.numpy(), o["best"]), float(np.abs(probs - o["probs"]).max())
assert np.isfinite(probs).all() and abs(float(probs.sum(-1).max()) - 1.0) < 1e-6
print("STUB_OK", probs.shape, float(probs[0, 0, 0]))
'''


@pytest.mark.gpu
def test_stub_runs_as_written_and_equals_the_binding():
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "memvul_amd", "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    for k in ("MEMVUL_COMPUTE", "MEMVUL_QKV_ASIDE", "MEMVUL_CLS_ASIDE", "MEMVUL_CLS_ASIDE_MIN_LEN"):
        env.pop(k, None)  # the stub names its compute dtype itself; the binding it is compared with must run on the library's defaults too
    code = "ROOT = %r\n" % ROOT + DRIVER
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "STUB_OK" in r.stdout, r.stderr[-3000:]
