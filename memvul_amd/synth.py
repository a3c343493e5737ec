"""Seeded synthetic weights and inputs for the MemVul inference hot path.

Everything here is numpy (PCG64) so the same bytes come out in this container and on
the GPU box, independent of the torch version.  Shapes and key names follow the
reference model's ``state_dict`` (attribute names at MemVul/model_memory.py:63,64,70,73
and MemVul/custom_PTM_embedder.py:99):

    _text_field_embedder.token_embedder_tokens.transformer_model.<HF BertModel keys>
    _bert_pooler.pooler.dense.{weight,bias}
    _projector_single._linear_layers.0.{weight,bias}
    _projector.weight

Inputs follow SURVEY.md §8(d): seed 2021, ``ids[b,0]=101`` ([CLS]), last real token
``102`` ([SEP]), interior uniform in [1000, vocab), no [PAD]=0 inside the real span.
"""
from __future__ import annotations

from dataclasses import dataclass, asdict
from typing import Dict, Optional, Tuple

import numpy as np

SEED = 2021  # the reference's seed (MemVul/config_memory.json:3)

PFX_BERT = "_text_field_embedder.token_embedder_tokens.transformer_model."
KEY_POOL_W = "_bert_pooler.pooler.dense.weight"
KEY_POOL_B = "_bert_pooler.pooler.dense.bias"
KEY_HEAD_W = "_projector_single._linear_layers.0.weight"
KEY_HEAD_B = "_projector_single._linear_layers.0.bias"
KEY_MATCH_W = "_projector.weight"


@dataclass
class BertDims:
    """bert-base-uncased geometry (HF BertConfig defaults); ``layers``/``vocab_size`` may be
    reduced for fast parity cases — the kernels are specialised to hidden=768, heads=12,
    intermediate=3072."""

    vocab_size: int = 30522
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    intermediate: int = 3072
    max_pos: int = 512
    type_vocab: int = 2
    proj_dim: int = 512  # FeedForward(768, 1, [512], ReLU) model_memory.py:70
    ln_eps: float = 1e-12

    def as_dict(self):
        return asdict(self)


def make_weights(
    dims: BertDims = BertDims(),
    seed: int = SEED,
    qk_scale: float = 1.0,
    match_scale: float = 1.0,
    plain_init: bool = False,
    ln_outliers: bool = False,
    trained_like: bool = False,
    use_header: bool = True,
    outlier_scale: float = 1.0,
) -> Dict[str, np.ndarray]:
    """Random-init weights of the reference architecture, fp32.

    HF init is N(0, 0.02) weights, zero biases, LayerNorm gamma=1 beta=0.  Zero biases and
    unit gammas would hide bias/affine bugs in a kernel, so unless ``plain_init`` the biases,
    gammas and betas are perturbed too.  ``qk_scale`` multiplies the query/key projections so
    attention is peaked rather than uniform (exercises the softmax); ``match_scale`` scales the
    matcher to produce "trained-like" |logit| ~ 3.  ``ln_outliers`` gives every
    LayerNorm two large-offset / large-gain dimensions, as trained BERT checkpoints have (a few hidden
    dims with |value| ~ 5-10 in every token): rows then have a visibly non-zero mean and a variance dominated by
    two dims, which is what the engine's folded-LayerNorm arithmetic has to survive.  (With gain 2.5 on a dimension
    whose offset has the same sign the outlier feeds on itself: fine for a few layers, but after 12 the stream collapses
    onto that dimension and every row encodes to the same vector.)  ``trained_like`` is the 12-layer form of it: the
    same two dimensions carry offsets -4 / +3 at gains 0.6 / 0.8 in every LayerNorm — a stable fixed point, |hidden| up
    to ~12 against a unit-variance bulk as in trained BERT checkpoints — so that, together with ``qk_scale`` >= 2 and a
    ``match_scale`` that puts |logit| near 3 (training temperature 0.1, config_memory.json:38), the 1e-3 logit tolerance
    is tested where it is hardest.  ``outlier_scale`` multiplies the two outlier offsets of ``trained_like``
    (1 = -4 / +3; the precision-envelope sweep of round 5 runs 1x, 3x, 10x: at 10x single hidden values pass 100 and reach the
    +-112 range of the MV_F16X8 fp8 planes, mv_x8_saturation).  ``use_header=False``: the state dict of a model built without the
    512-d header (no ``_projector_single``; ``_projector.weight`` is ``[2, 3 * 768]``).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    H, I, P = dims.hidden, dims.intermediate, dims.proj_dim
    w: Dict[str, np.ndarray] = {}

    def normal(shape, std=0.02):
        return (rng.standard_normal(shape, dtype=np.float32) * np.float32(std)).astype(np.float32)

    def bias(n, std=0.02):
        return np.zeros(n, np.float32) if plain_init else normal((n,), std)

    def gamma(n):
        return np.ones(n, np.float32) if plain_init else (1.0 + normal((n,), 0.1)).astype(np.float32)

    def beta(n):
        return np.zeros(n, np.float32) if plain_init else normal((n,), 0.05)

    def uniform_linear(out_f, in_f):
        k = 1.0 / np.sqrt(in_f)
        return rng.uniform(-k, k, size=(out_f, in_f)).astype(np.float32)

    e = PFX_BERT + "embeddings."
    w[e + "word_embeddings.weight"] = normal((dims.vocab_size, H))
    w[e + "position_embeddings.weight"] = normal((dims.max_pos, H))
    w[e + "token_type_embeddings.weight"] = normal((dims.type_vocab, H))
    w[e + "LayerNorm.weight"] = gamma(H)
    w[e + "LayerNorm.bias"] = beta(H)
    for l in range(dims.layers):
        p = PFX_BERT + f"encoder.layer.{l}."
        w[p + "attention.self.query.weight"] = normal((H, H)) * np.float32(qk_scale)
        w[p + "attention.self.query.bias"] = bias(H)
        w[p + "attention.self.key.weight"] = normal((H, H)) * np.float32(qk_scale)
        w[p + "attention.self.key.bias"] = bias(H)
        w[p + "attention.self.value.weight"] = normal((H, H))
        w[p + "attention.self.value.bias"] = bias(H)
        w[p + "attention.output.dense.weight"] = normal((H, H))
        w[p + "attention.output.dense.bias"] = bias(H)
        w[p + "attention.output.LayerNorm.weight"] = gamma(H)
        w[p + "attention.output.LayerNorm.bias"] = beta(H)
        w[p + "intermediate.dense.weight"] = normal((I, H))
        w[p + "intermediate.dense.bias"] = bias(I)
        w[p + "output.dense.weight"] = normal((H, I))
        w[p + "output.dense.bias"] = bias(H)
        w[p + "output.LayerNorm.weight"] = gamma(H)
        w[p + "output.LayerNorm.bias"] = beta(H)
    w[KEY_POOL_W] = normal((H, H))
    w[KEY_POOL_B] = bias(H)
    if use_header:
        w[KEY_HEAD_W] = uniform_linear(P, H)
        w[KEY_HEAD_B] = rng.uniform(-1 / np.sqrt(H), 1 / np.sqrt(H), size=(P,)).astype(np.float32)
    else:  # model_memory.py:69-73 with use_header=False: no _projector_single, the matcher runs on the 768-d pooler output
        P = H
    w[KEY_MATCH_W] = uniform_linear(2, 3 * P) * np.float32(match_scale)
    if ln_outliers:  # applied after generation: the random stream (and every golden vector) is unchanged without it
        for k in w:
            if k.endswith("LayerNorm.bias"):
                w[k] = w[k].copy()
                w[k][[H // 3 + 52, H // 2 - 3]] = [-6.0, 4.0]
            if k.endswith("LayerNorm.weight"):
                w[k] = w[k].copy()
                w[k][H // 3 + 52] = 2.5
    if trained_like:
        for k in w:
            if k.endswith("LayerNorm.bias"):
                w[k] = w[k].copy()
                w[k][[H // 3 + 52, H // 2 - 3]] = [-4.0 * outlier_scale, 3.0 * outlier_scale]
            if k.endswith("LayerNorm.weight"):
                w[k] = w[k].copy()
                w[k][[H // 3 + 52, H // 2 - 3]] = [0.6, 0.8]
    return w


def make_ids(
    n: int,
    seq_len: int,
    vocab_size: int = 30522,
    seed: int = SEED,
    ragged: bool = False,
    min_len: int = 4,
) -> Tuple[np.ndarray, np.ndarray]:
    """``ids int32[n, seq_len]`` (0-padded) and ``lens int32[n]``.

    Full-length rows unless ``ragged`` (then lengths are uniform in [min_len, seq_len] with at
    least one full-length row, mimicking pad-to-longest collation, predict_memory.py:97-101)."""
    rng = np.random.Generator(np.random.PCG64(seed + 7919 * seq_len + n))
    lo = min(1000, max(3, vocab_size // 2))
    ids = rng.integers(lo, vocab_size, size=(n, seq_len), dtype=np.int64).astype(np.int32)
    if ragged:
        lens = rng.integers(min_len, seq_len + 1, size=(n,), dtype=np.int64).astype(np.int32)
        lens[rng.integers(0, n)] = seq_len
    else:
        lens = np.full((n,), seq_len, np.int32)
    cls_id = 101 if vocab_size > 102 else 1
    sep_id = 102 if vocab_size > 102 else 2
    for b in range(n):
        L = int(lens[b])
        ids[b, 0] = cls_id
        ids[b, L - 1] = sep_id
        ids[b, L:] = 0
    return ids, lens


def make_labels(n: int, pos_rate: float = 3937.0 / 1221677.0, seed: int = SEED) -> np.ndarray:
    """Bernoulli CIR labels at the corpus' positive rate (README.md:8: 3,937 / 1,221,677)."""
    rng = np.random.Generator(np.random.PCG64(seed + 104729))
    return (rng.random(n) < pos_rate).astype(np.uint8)


def make_anchor_bank(g: int, proj_dim: int = 512, seed: int = SEED) -> np.ndarray:
    """A synthetic anchor-embedding bank ``v fp32[g, proj_dim]`` (post-ReLU statistics), for the
    matcher-only configuration (BASELINE.json configs[4])."""
    rng = np.random.Generator(np.random.PCG64(seed + 15485863))
    v = rng.standard_normal((g, proj_dim), dtype=np.float32) * np.float32(0.5)
    return np.maximum(v, 0).astype(np.float32)


def mask_from_lens(lens: np.ndarray, seq_len: int) -> np.ndarray:
    return (np.arange(seq_len)[None, :] < np.asarray(lens)[:, None])
