"""CPU ORACLE (test infrastructure — NOT the product path).

A numpy restatement of the arithmetic of MemVul's ``predict_memory.py`` hot loop.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; ``memvul_amd`` never does and fails loudly if its HIP library is missing.

What is restated (reference file:line, relative to the MemVul repo):

* ``encode``            custom_PTM_embedder.py:199-235 — token-type ids all zero -> None,
                        ``attention_mask = mask.float()``, returns ``last_hidden_state``.  The
                        BERT forward itself lives in the un-vendored dependency
                        ``transformers==4.1.0`` (README.md:25-27): ``BertEmbeddings``
                        (word+position+type, LayerNorm eps 1e-12), 12 x ``BertLayer``
                        (scores/sqrt(64) + additive mask (1-m)*-10000, softmax, exact-erf GELU,
                        post-LayerNorm residuals).  Its published algorithm is restated here.
* ``instance_forward``  model_memory.py:90-103 — BertPooler tanh(h[:,0] W_p^T + b_p) (l.99),
                        header FeedForward(768,1,[512],ReLU) (l.70,101-102); dropout = identity
                        in eval.
* ``match``             model_memory.py:135-147 — logits = W_m [u; v; |u-v|] (bias-free, l.73),
                        softmax over the size-2 axis (l.142), best anchor = argmax_g p[b,g,same]
                        (l.144-145), probs[b] = p[b, g*] (l.146-147).
* ``anchor bank``       model_memory.py:105-115 + predict_memory.py:81-83 (chunks of 128, each
                        padded to its own longest member).

Parity pinning: the reference ships no tests or golden vectors and cannot be
imported here (allennlp/overrides absent), so the BERT part is pinned against the installed
``transformers`` BertModel (eager attention, fp32) by ``tests/golden/make_golden.py``; the
resulting vectors are committed under ``tests/golden/`` and checked by
``tests/test_oracle_golden.py``.  Known skew: transformers 4.1.0 masks with -10000.0, 5.x with
finfo.min — identical for full-length rows, < 1e-6 for padded ones.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np

try:  # exact erf; scipy is in the image (here and on the GPU box)
    from scipy.special import erf as _erf
except Exception:  # pragma: no cover
    import math

    _erf = np.vectorize(math.erf)

PFX = "_text_field_embedder.token_embedder_tokens.transformer_model."
MASK_ADD = -10000.0  # transformers 4.1.0 get_extended_attention_mask


def _ln(x, g, b, eps):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * g + b


def _gelu(x):
    return x * 0.5 * (1.0 + _erf(x / np.sqrt(2.0).astype(x.dtype)))


def _softmax(x):
    m = x.max(-1, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(-1, keepdims=True)


def n_layers(w: Dict[str, np.ndarray]) -> int:
    n = 0
    while (PFX + f"encoder.layer.{n}.attention.self.query.weight") in w:
        n += 1
    return n


def encode(
    w: Dict[str, np.ndarray],
    ids: np.ndarray,
    mask: np.ndarray,
    heads: int = 12,
    eps: float = 1e-12,
    dtype=np.float32,
    taps: Optional[dict] = None,
    stream_round=None,
) -> np.ndarray:
    """BERT forward -> last_hidden_state ``[B,S,H]``.  ``taps`` (if a dict) receives the
    intermediate tensors the per-kernel GPU tests compare against.  ``stream_round`` (tests only): a function applied
    to every PRE-LayerNorm residual sum (embedding sum, attention-output + x, FFN output + x1) — the tensors the
    engine keeps in HBM between kernels — to model the storage format of that stream (tests/test_stream_precision.py)."""
    sr = (lambda t: t) if stream_round is None else (lambda t: stream_round(t).astype(dtype))  # noqa: E731
    W = lambda k: w[PFX + k].astype(dtype)  # noqa: E731
    B, S = ids.shape
    H = W("embeddings.word_embeddings.weight").shape[1]
    d = H // heads
    x = (
        W("embeddings.word_embeddings.weight")[ids]
        + W("embeddings.position_embeddings.weight")[np.arange(S)][None]
        + W("embeddings.token_type_embeddings.weight")[0][None, None]
    )
    x = _ln(sr(x), W("embeddings.LayerNorm.weight"), W("embeddings.LayerNorm.bias"), dtype(eps))
    if taps is not None:
        taps["embed"] = x.copy()
    addmask = ((1.0 - mask.astype(dtype)) * dtype(MASK_ADD))[:, None, None, :]
    for l in range(n_layers(w)):
        p = f"encoder.layer.{l}."
        q = x @ W(p + "attention.self.query.weight").T + W(p + "attention.self.query.bias")
        k = x @ W(p + "attention.self.key.weight").T + W(p + "attention.self.key.bias")
        v = x @ W(p + "attention.self.value.weight").T + W(p + "attention.self.value.bias")
        sp = lambda t: t.reshape(B, S, heads, d).transpose(0, 2, 1, 3)  # noqa: E731
        qh, kh, vh = sp(q), sp(k), sp(v)
        sc = qh @ kh.transpose(0, 1, 3, 2) / dtype(np.sqrt(d)) + addmask
        pr = _softmax(sc)
        ctx = (pr @ vh).transpose(0, 2, 1, 3).reshape(B, S, H)
        ao = ctx @ W(p + "attention.output.dense.weight").T + W(p + "attention.output.dense.bias")
        x1 = _ln(sr(ao + x), W(p + "attention.output.LayerNorm.weight"), W(p + "attention.output.LayerNorm.bias"), dtype(eps))
        h = _gelu(x1 @ W(p + "intermediate.dense.weight").T + W(p + "intermediate.dense.bias"))
        fo = h @ W(p + "output.dense.weight").T + W(p + "output.dense.bias")
        x = _ln(sr(fo + x1), W(p + "output.LayerNorm.weight"), W(p + "output.LayerNorm.bias"), dtype(eps))
        if taps is not None:
            if l == 0:
                taps["l0_q"], taps["l0_k"], taps["l0_v"] = qh.copy(), kh.copy(), vh.copy()
                taps["l0_ctx"] = ctx.copy()
                taps["l0_attn_ln"] = x1.copy()
                taps["l0_gelu"] = h.copy()
            taps[f"layer{l}"] = x.copy()
    return x


def instance_forward(w, ids, mask, heads=12, eps=1e-12, dtype=np.float32, taps=None, stream_round=None) -> np.ndarray:
    """model_memory.py:90-103 -> ``u [B,512]`` with use_header=True (every reference config); a state dict WITHOUT
    ``_projector_single`` is a model built with use_header=False (l.69-73): the embedding is then the pooler output [B,768]."""
    h = encode(w, ids, mask, heads, eps, dtype, taps, stream_round)
    cls = h[:, 0]
    pooled = np.tanh(cls @ w["_bert_pooler.pooler.dense.weight"].astype(dtype).T + w["_bert_pooler.pooler.dense.bias"].astype(dtype))
    if "_projector_single._linear_layers.0.weight" not in w:
        if taps is not None:
            taps["pooled"] = pooled.copy()
            taps["u"] = pooled.copy()
        return pooled
    u = np.maximum(
        pooled @ w["_projector_single._linear_layers.0.weight"].astype(dtype).T
        + w["_projector_single._linear_layers.0.bias"].astype(dtype),
        0,
    )
    if taps is not None:
        taps["pooled"] = pooled.copy()
        taps["u"] = u.copy()
    return u


def match(u: np.ndarray, v: np.ndarray, w_match: np.ndarray, same_idx: int = 0):
    """model_memory.py:135-147.  Returns logits [B,G,2], p [B,G,2], best [B,2], idx [B]."""
    dt = u.dtype
    B, G = u.shape[0], v.shape[0]
    ue = np.broadcast_to(u[:, None, :], (B, G, u.shape[1]))
    ve = np.broadcast_to(v[None, :, :], (B, G, v.shape[1]))
    feat = np.concatenate([ue, ve, np.abs(ue - ve)], -1)
    logits = feat @ w_match.astype(dt).T
    p = _softmax(logits)
    idx = np.argmax(p, axis=1)[:, same_idx]  # torch.argmax / np.argmax: first maximal index
    best = p[np.arange(B), idx]
    return logits, p, best, idx.astype(np.int64)


def topk_match(p_same: np.ndarray, k: int):
    """Top-k anchors per IR by P(same), ties -> lower anchor index first (BASELINE configs[4])."""
    order = np.argsort(-p_same, axis=1, kind="stable")[:, :k]
    return np.take_along_axis(p_same, order, 1), order


def build_anchor_bank(w, anchor_ids: List[np.ndarray], heads=12, eps=1e-12, dtype=np.float32, chunk: int = 128):
    """predict_memory.py:81-83 + model_memory.py:105-115: anchors forwarded in chunks of 128,
    each chunk padded to its own longest member, embeddings concatenated."""
    outs = []
    for s in range(0, len(anchor_ids), chunk):
        part = anchor_ids[s : s + chunk]
        L = max(len(a) for a in part)
        ids = np.zeros((len(part), L), np.int64)
        mask = np.zeros((len(part), L), bool)
        for i, a in enumerate(part):
            ids[i, : len(a)] = a
            mask[i, : len(a)] = True
        outs.append(instance_forward(w, ids, mask, heads, eps, dtype))
    return np.concatenate(outs, 0)


def predict(w, ids, mask, v, same_idx=0, heads=12, eps=1e-12, dtype=np.float32):
    """One hot-loop iteration (model_memory.py:133-147) on one batch."""
    u = instance_forward(w, ids, mask, heads, eps, dtype)
    return (u,) + match(u, v.astype(dtype), w["_projector.weight"], same_idx)


def classify_single(u: np.ndarray, w_cls: np.ndarray):
    """MemVul-m head (model_single.py:66, 92-94): logits = Linear(512, num_class, bias=False)(u), probs = softmax."""
    logits = u @ w_cls.astype(u.dtype).T
    return logits, _softmax(logits)
