"""VERDICT r3 #6 — the two remaining callers of the golden-anchor path on the HIP engine (not on the oracle stand-in):

  * ``load_archive`` of a REAL ``model.tar.gz`` (``config.json`` + ``weights.th`` = ``torch.save(state_dict)`` + ``vocabulary/``;
    reference call predict_memory.py:62-70) -> ``mv_load_tensor`` / ``mv_finalize_weights``;
  * ``CustomValidation.on_epoch`` (MemVul/callbacks.py:41-53): ``_golden_instances_embeddings = None`` -> ``mv_anchor_reset``,
    then the anchor file through ``forward_on_instances`` in chunks of 128 -> ``mv_anchor_append`` — with MORE than 128 anchors,
    so the second chunk exists — twice (a second epoch REPLACES the bank, bit for bit).

torch writes / reads ``weights.th``, so the whole thing runs in a fresh process with torch imported FIRST (the supported
load order next to libmemvul_hip.so: tests/test_gpu_parity.py::test_engine_coexists_with_torch_hip_runtime).  The checker is
the same driver on the numpy-oracle engine (tests/plumbing_util.OracleEngine)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BODY = r"""
import json, os, sys, tarfile
import numpy as np
import torch                                   # FIRST: archive.py deserialises weights.th with it
sys.path.insert(0, os.path.join(ROOT, "tests"))
import plumbing_util as pu
from memvul_amd import callbacks, model_memory
from memvul_amd.archive import load_archive

N_ANCHORS = 140                                # > 128: callbacks.py:50-53 runs its second chunk
root, arch, golden, test_path, w, dims = pu.make_fixture(n_irs=4, n_anchors=N_ANCHORS, layers=2)
stage = os.path.join(root, "stage"); os.makedirs(os.path.join(stage, "vocabulary"))
for name in ("config.json", "vocabulary/labels.txt", "vocabulary/non_padded_namespaces.txt"):
    open(os.path.join(stage, name), "wb").write(open(os.path.join(arch, name), "rb").read())
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in w.items()}
sd["_text_field_embedder.token_embedder_tokens.transformer_model.embeddings.position_ids"] = torch.arange(512).unsqueeze(0)
torch.save(sd, os.path.join(stage, "weights.th"))
tar_path = os.path.join(root, "model.tar.gz")
with tarfile.open(tar_path, "w:gz") as tf:
    for name in ("config.json", "weights.th", "vocabulary"):
        tf.add(os.path.join(stage, name), arcname=name)

class Trainer: pass

def epochs(engine_cls):
    if engine_cls is not None:
        model_memory.Engine = engine_cls
    archive = load_archive(tar_path, overrides=pu.TEST_CONFIG, cuda_device=0,
                           engine_options=dict(max_tokens=128 * 64, max_batch=128, max_anchors=256))
    model = archive.model
    cb = callbacks.CustomValidation(anchor_path=golden, data_reader=archive.validation_dataset_reader)
    tr = Trainer(); tr.model = model
    banks = []
    for ep in range(2):
        cb.on_epoch(tr, {}, ep, True)
        assert len(model._golden_instances_labels) == N_ANCHORS
        banks.append(np.array(model._golden_instances_embeddings, dtype=np.float32, copy=True))
    return model, banks

hip_engine = model_memory.Engine
model, banks = epochs(None)
assert type(model._engine).__module__ == "memvul_amd.binding", type(model._engine)     # the HIP engine, not a stand-in
assert model._engine.n_anchors == N_ANCHORS
assert banks[0].shape == (N_ANCHORS, 512) and np.isfinite(banks[0]).all()
assert np.array_equal(banks[0], banks[1]), "second epoch must REPLACE the bank with identical rows"
assert np.array_equal(model._engine.anchor_get(), banks[1])                            # what mv_forward will match against
_, banks_o = epochs(pu.OracleEngine)
err = float(np.abs(banks[0] - banks_o[0]).max())
print("BANK_ERR", err, "max|v|", float(np.abs(banks_o[0]).max()))
assert err <= 1e-3, err
print("OK")
"""


@pytest.mark.gpu
def test_tar_gz_archive_and_anchor_refresh_callback_on_the_hip_engine():
    code = f"ROOT = {ROOT!r}\n" + BODY
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gpu_util as gu

    line = [ln for ln in r.stdout.splitlines() if ln.startswith("BANK_ERR")][-1].split()
    gu.record("archive_callback", bank_err=float(line[1]))
