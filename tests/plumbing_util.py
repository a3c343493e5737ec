"""Synthetic end-to-end fixture for the drop-in path (BASELINE.json configs[0] shape, scaled down):
an AllenNLP-style archive directory + golden anchor file + test file, and an oracle-backed stand-in for the
engine so the host plumbing can be exercised on a machine without a GPU (tests only)."""
import json
import os
import tempfile

import numpy as np

from memvul_amd import synth
from oracle import memvul_oracle as orc

CONFIG = {  # MemVul/config_memory.json, rendered (local variables substituted), trainer section dropped
    "random_seed": 2021, "numpy_seed": 2021, "pytorch_seed": 2021,
    "dataset_reader": {
        "type": "reader_memory", "sample_neg": 0.01, "train_iter": 1, "same_diff_ratio": {"diff": 16, "same": 16},
        "anchor_path": "CWE_anchor_golden_project.json",
        "tokenizer": {"type": "pretrained_transformer", "model_name": "bert-base-uncased", "add_special_tokens": True, "max_length": 256},
        "token_indexers": {"tokens": {"type": "pretrained_transformer", "model_name": "bert-base-uncased", "namespace": "tags"}},
    },
    "model": {
        "type": "model_memory", "label_namespace": "labels", "dropout": 0.1, "device": "cuda:0", "use_header": True,
        "PTM": "bert-base-uncased", "temperature": 0.1,
        "text_field_embedder": {"token_embedders": {"tokens": {
            "type": "custom_pretrained_transformer", "model_name": "bert-base-uncased", "train_parameters": True,
            "pretrained_model_path": "further_pretrain/out_wwm/"}}},
    },
    "data_loader": {"batch_size": 32, "shuffle": False},
    "validation_data_loader": {"batch_size": 512, "shuffle": False},
}
TEST_CONFIG = {  # test_config_memory.json verbatim
    "validation_dataset_reader": {
        "type": "reader_memory", "target": "Security_Issue_Full",
        "tokenizer": {"type": "pretrained_transformer", "model_name": "bert-base-uncased", "add_special_tokens": True, "max_length": 512},
        "token_indexers": {"tokens": {"type": "pretrained_transformer", "model_name": "bert-base-uncased", "namespace": "tags"}},
    },
    "model": {"device": "cuda:0"},
    "validation_data_loader": {"batch_size": 512, "shuffle": False},
}
WORDS = ("buffer overflow heap stack sql injection xss csrf auth bypass token leak race deadlock crash null pointer "
         "deref format string path traversal upload parser json yaml xml regex dos memory use after free double "
         "integer underflow privilege escalation sandbox escape cookie session header redirect ssrf").split()


def _text(rng, n):
    return " ".join(rng.choice(WORDS, size=n))


def make_fixture(n_irs=40, n_anchors=8, layers=2, seed=7, body_words=(5, 70), use_header=True):
    """Returns (dir, archive_dir, golden_path, test_path, weights, dims).  use_header=False: the archive of a model built
    without the 512-d header (config `use_header: false`, no _projector_single in the weights; model_memory.py:69-73)."""
    rng = np.random.default_rng(seed)
    root = tempfile.mkdtemp(prefix="mvplumb")  # no "test_"/"golden" in the directory name (reader dispatches on substrings)
    arch = os.path.join(root, "archive")
    os.makedirs(os.path.join(arch, "vocabulary"))
    dims = synth.BertDims(layers=layers)
    w = synth.make_weights(dims, qk_scale=2.0, match_scale=6.0, use_header=use_header)
    np.savez(os.path.join(arch, "weights.npz"), **w)
    config = json.loads(json.dumps(CONFIG))
    config["model"]["use_header"] = bool(use_header)
    json.dump(config, open(os.path.join(arch, "config.json"), "w"))
    open(os.path.join(arch, "vocabulary", "labels.txt"), "w").write("same\ndiff\n")
    open(os.path.join(arch, "vocabulary", "non_padded_namespaces.txt"), "w").write("*labels\n*tags\n")
    cwes = [f"CWE-{100 + i}" for i in range(n_anchors)]
    golden = os.path.join(root, "CWE_anchor_golden_project.json")
    json.dump({c: _text(rng, int(rng.integers(10, 60))) for c in cwes}, open(golden, "w"))
    recs = []
    for i in range(n_irs):
        pos = i % 7 == 3
        recs.append({"Issue_Title": _text(rng, 6), "Issue_Body": _text(rng, int(rng.integers(*body_words))),
                     "Security_Issue_Full": "1" if pos else "0", "Issue_Url": f"https://example.invalid/issues/{i}",
                     "CVE_ID": f"CVE-2020-{i}" if pos else None, "CWE_ID": str(rng.choice(cwes)) if pos else None})
    test_path = os.path.join(root, "test_project.json")
    json.dump(recs, open(test_path, "w"))
    os.makedirs(os.path.join(root, "test_results"))
    return root, arch, golden, test_path, w, dims


class OracleEngine:
    """Test-only stand-in with the binding.Engine surface ModelMemory uses, backed by the numpy oracle."""

    last_device = None  # device index the most recent instance was created for

    def __init__(self, device=0, **kw):
        OracleEngine.last_device = device
        self.same_idx = kw.get("same_idx", 0)
        self.P = kw.get("proj_dim", 512)
        self.v = np.zeros((0, self.P), np.float32)
        self.w = None

    def load_state_dict(self, sd, compute_dtype=1):
        self.w = {k: np.asarray(v) for k, v in sd.items()}

    def close(self):
        pass

    def anchor_reset(self):
        self.v = np.zeros((0, self.P), np.float32)

    @property
    def n_anchors(self):
        return self.v.shape[0]

    def anchor_get(self):
        return self.v.copy()

    def anchor_set(self, v):
        self.v = np.asarray(v, np.float32)

    def _mask(self, ids, lens):
        return np.arange(ids.shape[1])[None, :] < np.asarray(lens)[:, None]

    def encode(self, ids, lens):
        return orc.instance_forward(self.w, np.asarray(ids, np.int64), self._mask(ids, lens))

    def anchor_append(self, ids, lens):
        self.v = np.concatenate([self.v, self.encode(ids, lens)], 0)

    def forward(self, ids, lens, want_logits=True, want_probs=True, want_embed=False):
        u = self.encode(ids, lens)
        logits, p, best, idx = orc.match(u, self.v, self.w[synth.KEY_MATCH_W], self.same_idx)
        return {"logits": logits, "probs": p, "best": best, "best_idx": idx.astype(np.int32), "embed": u}

    # the product's own grouping logic (host code: it only needs forward / n_anchors / P of the engine it runs on)
    from memvul_amd.binding import Engine as _ProductEngine
    BY_LENGTH_MIN_TOKENS = _ProductEngine.BY_LENGTH_MIN_TOKENS
    forward_by_length = _ProductEngine.forward_by_length
    del _ProductEngine

    def bucketed_sweep(self, ids, lens, batch, with_probs=False):
        """binding.Engine.bucketed_sweep on the oracle: per-row results do not depend on the batching."""
        best, idx, ps = [], [], []
        for s0 in range(0, ids.shape[0], batch):
            L = int(np.asarray(lens[s0:s0 + batch]).max())
            o = self.forward(ids[s0:s0 + batch, :L], lens[s0:s0 + batch])
            best.append(o["best"]); idx.append(o["best_idx"]); ps.append(o["probs"][:, :, self.same_idx])
        return np.concatenate(best), np.concatenate(idx), (np.concatenate(ps) if with_probs else None)
