"""CPU ORACLE (test infrastructure — NOT the product path): the reference's CPU graph built
from the *real* third-party modules the reference delegates to.

The reference (``MemVul/model_memory.py``) owns no arithmetic of its own below the matcher: it
calls HuggingFace ``BertModel`` (custom_PTM_embedder.py:99,228), AllenNLP ``BertPooler``
(= HF ``BertPooler``: tanh(Linear(h[:,0])), model_memory.py:64,99), AllenNLP ``FeedForward``
(= Linear+ReLU, model_memory.py:70,102) and ``nn.Linear(1536,2,bias=False)`` (l.73).  AllenNLP
itself is not installed here, so this module assembles the same torch modules directly and runs
them in fp32 on the host cores with eager attention.  It serves two purposes:

1. pin ``oracle/memvul_oracle.py`` (the numpy restatement) — ``tests/golden/make_golden.py``;
2. be the CPU baseline timed by ``bench.py`` (``cpu_baseline.kind == "port"``).

transformers here is 5.x, the reference pins 4.1.0 (README.md:25-27); see the skew note in
``memvul_oracle.py``.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

PFX = "_text_field_embedder.token_embedder_tokens.transformer_model."


class HFReference:
    def __init__(self, weights: Dict[str, np.ndarray], dims: dict, threads: int | None = None):
        import torch
        from transformers import BertConfig, BertModel

        self.torch = torch
        if threads:
            torch.set_num_threads(threads)
        cfg = BertConfig(
            vocab_size=dims["vocab_size"],
            hidden_size=dims["hidden"],
            num_hidden_layers=dims["layers"],
            num_attention_heads=dims["heads"],
            intermediate_size=dims["intermediate"],
            max_position_embeddings=dims["max_pos"],
            type_vocab_size=dims["type_vocab"],
            layer_norm_eps=dims["ln_eps"],
            hidden_act="gelu",
            hidden_dropout_prob=0.1,
            attention_probs_dropout_prob=0.1,
        )
        cfg._attn_implementation = "eager"
        self.bert = BertModel(cfg, add_pooling_layer=False).eval()
        sd = {k[len(PFX):]: torch.from_numpy(np.ascontiguousarray(v)) for k, v in weights.items() if k.startswith(PFX)}
        missing, unexpected = self.bert.load_state_dict(sd, strict=False)
        missing = [m for m in missing if not m.endswith("position_ids")]  # custom_PTM_embedder.py:64
        assert not missing and not unexpected, (missing, unexpected)
        t = lambda k: torch.from_numpy(np.ascontiguousarray(weights[k]))  # noqa: E731
        self.pool_w, self.pool_b = t("_bert_pooler.pooler.dense.weight"), t("_bert_pooler.pooler.dense.bias")
        self.head_w, self.head_b = t("_projector_single._linear_layers.0.weight"), t("_projector_single._linear_layers.0.bias")
        self.match_w = t("_projector.weight")

    def instance_forward(self, ids: np.ndarray, mask: np.ndarray, all_hidden: bool = False):
        torch = self.torch
        with torch.no_grad():
            out = self.bert(
                input_ids=torch.from_numpy(ids.astype(np.int64)),
                attention_mask=torch.from_numpy(mask.astype(np.float32)),
                output_hidden_states=all_hidden,
            )
            h = out.last_hidden_state
            pooled = torch.tanh(h[:, 0] @ self.pool_w.T + self.pool_b)
            u = torch.relu(pooled @ self.head_w.T + self.head_b)
        if all_hidden:
            return u.numpy(), [x.numpy() for x in out.hidden_states]
        return u.numpy()

    def match(self, u: np.ndarray, v: np.ndarray, same_idx: int = 0):
        """model_memory.py:135-147 written with the same torch calls as the reference."""
        torch = self.torch
        with torch.no_grad():
            e1 = torch.from_numpy(u)
            g = torch.from_numpy(v)
            shape = e1.shape
            se = e1.view(shape[0], -1, shape[1]).expand(-1, g.shape[0], -1)
            ge = g.expand(shape[0], -1, -1)
            logits = torch.cat([se, ge, torch.abs(se - ge)], -1) @ self.match_w.T
            p = torch.nn.functional.softmax(logits, dim=-1)
            idx = torch.argmax(p, dim=1)[:, same_idx]
            best = torch.stack([p[i][idx[i]] for i in range(shape[0])])
        return logits.numpy(), p.numpy(), best.numpy(), idx.numpy()

    def predict(self, ids, mask, v, same_idx=0):
        u = self.instance_forward(ids, mask)
        return (u,) + self.match(u, v, same_idx)
