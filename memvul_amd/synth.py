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
    sink: Optional[dict] = None,
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
    ``sink`` (round 6: the attention-concentration axis of the precision envelope): ``dict(token="sep" | "cls", rows="cls" | "all", gains=[g_0 ..
    g_{L-1}])`` makes every head of every layer put a chosen share of its attention mass on ONE token — the [SEP] or the [CLS] token of each
    sequence, as trained BERT checkpoints do — for the [CLS] row alone or for every row; see ``apply_sink`` (the gains come from
    ``calibrate_sink``, which measures the achieved mass on the CPU oracle).
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
    if sink:
        apply_sink(w, dims, seed, **sink)
    return w


# ---- attention sinks (round 6) -------------------------------------------------------------------------------------------------------
# Trained BERT heads put most of their mass on [SEP] / [CLS] (1 - 4 effective keys of 256); the random-init family above spreads the [CLS] row over
# 67 - 149 keys.  What matters for the engine's precision argument is exactly that: the [CLS]-row form of the correction sweeps and the one-plane
# Q / K / V / P storage rest on "another row's rounding reaches the [CLS] row averaged over the keys".  Mechanism, in the weights only (the model
# stays a plain BERT state dict): hidden dimension SINK_DIM is a FLAG of the sink token (its word embedding carries +SINK_FLAG there, the LayerNorms
# keep the dimension alive with gamma = SINK_GAMMA, beta = 0), every head h of layer l gets a rank-1 key term  k_h += g_l x[SINK_DIM] dir_h  and
# every query a component along dir_h — a constant for all rows (rows = "all":  b_q += 8 dir_h) or proportional to a second flag that only the
# [CLS] token carries (rows = "cls":  W_q[:, CLSQ_DIM] += 8 dir_h / SINK_NOMINAL) — so the score of (query, sink key) is raised by ~ g_l x_sink.
SINK_DIM, CLSQ_DIM = 71, 167
SINK_FLAG, SINK_GAMMA, SINK_NOMINAL = 0.3, 1.1, 8.0
CLS_ID, SEP_ID, MID_ID = 101, 102, 1012  # token = "mid": an ORDINARY token (bert-base-uncased's "."), placed in the middle of each sequence by mark_mid_token


def sink_token_id(token: str) -> int:
    return {"sep": SEP_ID, "cls": CLS_ID, "mid": MID_ID}[token]


def sink_positions(lens, token: str) -> np.ndarray:
    lens = np.asarray(lens)
    return {"sep": lens - 1, "cls": np.zeros_like(lens), "mid": lens // 2}[token]


def mark_mid_token(ids, lens):
    """ids with MID_ID at position len // 2 of every row (the sink token of token = "mid"; rows of fewer than 4 tokens are left alone)."""
    ids = np.array(ids, copy=True)
    for b, n in enumerate(np.asarray(lens)):
        if n >= 4:
            ids[b, int(n) // 2] = MID_ID
    return ids


def _sink_dirs(dims: "BertDims", seed: int) -> np.ndarray:
    """One unit direction per (layer, head) in the head's 64-d space, by seed: [L, heads, 64]."""
    rng = np.random.Generator(np.random.PCG64(seed + 424243))
    dirs = rng.standard_normal((dims.layers, dims.heads, dims.hidden // dims.heads)).astype(np.float32)
    return dirs / np.linalg.norm(dirs, axis=-1, keepdims=True)


def apply_sink(w, dims: "BertDims", seed: int, token: str = "sep", rows: str = "cls", gains=None):
    """In place: the sink terms described above.  ``gains``: one g_l per layer (calibrate_sink); None = no key terms yet (what calibrate_sink starts from)."""
    assert token in ("sep", "cls", "mid") and rows in ("cls", "all")
    H = dims.hidden
    dirs = _sink_dirs(dims, seed)
    e = PFX_BERT + "embeddings."
    we = w[e + "word_embeddings.weight"] = w[e + "word_embeddings.weight"].copy()
    we[sink_token_id(token), SINK_DIM] += SINK_FLAG
    if rows == "cls":
        we[CLS_ID, CLSQ_DIM] += SINK_FLAG
    for k in list(w):
        if k.endswith("LayerNorm.weight"):
            w[k] = w[k].copy()
            w[k][[SINK_DIM, CLSQ_DIM]] = SINK_GAMMA
        if k.endswith("LayerNorm.bias"):
            w[k] = w[k].copy()
            w[k][[SINK_DIM, CLSQ_DIM]] = 0.0
    for l in range(dims.layers):
        p = PFX_BERT + f"encoder.layer.{l}."
        d = dirs[l].reshape(H)  # head-major: rows h * 64 .. of the projections
        if rows == "all":
            w[p + "attention.self.query.bias"] = (w[p + "attention.self.query.bias"] + 8.0 * d).astype(np.float32)
        else:
            wq = w[p + "attention.self.query.weight"] = w[p + "attention.self.query.weight"].copy()
            wq[:, CLSQ_DIM] += np.float32(8.0 / SINK_NOMINAL) * d
        if gains is not None:
            wk = w[p + "attention.self.key.weight"] = w[p + "attention.self.key.weight"].copy()
            wk[:, SINK_DIM] += np.float32(gains[l]) * d
    return w


def sink_report(w, dims: "BertDims", ids, lens, token: str = "sep", rows: str = "cls"):
    """Per layer, on an fp32 numpy restatement of the forward (kept here so that calibrate_sink needs nothing outside this module): the mean share of
    attention mass on the sink token — of the [CLS] row (rows = "cls") or of all valid rows (rows = "all") — and the [CLS] row's effective number of
    keys 1 / sum p^2, both averaged over sequences and heads.  Returns (mass [L], eff_keys [L])."""
    return _sink_forward(w, dims, ids, lens, token, rows, None, None)[:2]


def calibrate_sink(dims: "BertDims", seed: int, target: float, token: str = "sep", rows: str = "cls", n: int = 4, seq_len: int = 256, **weight_kw):
    """gains [L] such that the measured mass share (sink_report) is ``target`` in every layer, found layer by layer by bisection on the CPU (the
    stream into layer l depends on the gains of the layers before it).  Calibration batch: n full-length sequences of seq_len tokens."""
    ids, lens = make_ids(n, seq_len, dims.vocab_size, seed=seed + 31337)
    if token == "mid":
        ids = mark_mid_token(ids, lens)
    w = make_weights(dims, seed=seed, sink=dict(token=token, rows=rows, gains=None), **weight_kw)
    return _sink_forward(w, dims, ids, lens, token, rows, target, seed)[2]


def _sink_forward(w, dims, ids, lens, token, rows, target, seed):
    from scipy.special import erf as _erf

    H, nh = dims.hidden, dims.heads
    hd = H // nh
    B, S = ids.shape
    W = lambda k: w[PFX_BERT + k].astype(np.float32)  # noqa: E731
    dirs = _sink_dirs(dims, seed) if target is not None else None

    def ln(x, g, b):
        mu = x.mean(-1, keepdims=True)
        var = ((x - mu) ** 2).mean(-1, keepdims=True)
        return (x - mu) / np.sqrt(var + np.float32(dims.ln_eps)) * g + b

    def probs(scores):
        e = np.exp(scores - scores.max(-1, keepdims=True))
        return e / e.sum(-1, keepdims=True)

    x = W("embeddings.word_embeddings.weight")[ids] + W("embeddings.position_embeddings.weight")[np.arange(S)][None] + W("embeddings.token_type_embeddings.weight")[0][None, None]
    x = ln(x, W("embeddings.LayerNorm.weight"), W("embeddings.LayerNorm.bias"))
    mask = mask_from_lens(lens, S)
    addmask = ((1.0 - mask.astype(np.float32)) * np.float32(-10000.0))[:, None, None, :]
    spos = sink_positions(lens, token)
    bi = np.arange(B)

    def measure(pr):
        on_sink = pr[bi, :, :, spos]  # [B, nh, S(query)]
        if rows == "cls":
            mass = float(on_sink[:, :, 0].mean())
        else:
            mass = float((on_sink * mask[:, None, :]).sum() / (mask.sum() * nh))
        return mass, float((1.0 / (pr[:, :, 0, :] ** 2).sum(-1)).mean())

    masses, effs, gains = [], [], []
    for l in range(dims.layers):
        p = f"encoder.layer.{l}."
        sp = lambda t: t.reshape(B, S, nh, hd).transpose(0, 2, 1, 3)  # noqa: E731
        qh = sp(x @ W(p + "attention.self.query.weight").T + W(p + "attention.self.query.bias"))
        kh = sp(x @ W(p + "attention.self.key.weight").T + W(p + "attention.self.key.bias"))
        vh = sp(x @ W(p + "attention.self.value.weight").T + W(p + "attention.self.value.bias"))
        sc = qh @ kh.transpose(0, 1, 3, 2) / np.float32(8.0) + addmask  # [B, nh, S, S]
        if target is not None:
            # the rank-1 key term  k_h += g x[SINK_DIM] dir_h  adds  g x_j[SINK_DIM] (q_i . dir_h) / 8  to score (i, j): bisect g on the measured mass
            qd = (qh * dirs[l][None, :, None, :]).sum(-1) / np.float32(8.0)  # [B, nh, S]
            extra = qd[:, :, :, None] * x[:, None, None, :, SINK_DIM]        # [B, nh, S, S]
            lo, hi = 0.0, 64.0
            for _ in range(18):
                g = 0.5 * (lo + hi)
                if measure(probs(sc + np.float32(g) * extra))[0] < target:
                    lo = g
                else:
                    hi = g
            g = float(np.float32(0.5 * (lo + hi)))
            gains.append(g)
            # continue on the weights as apply_sink will write them (the term lands in W_k in fp32)
            wk = w[PFX_BERT + p + "attention.self.key.weight"] = w[PFX_BERT + p + "attention.self.key.weight"].copy()
            wk[:, SINK_DIM] += np.float32(g) * dirs[l].reshape(H)
            kh = sp(x @ wk.T + W(p + "attention.self.key.bias"))
            sc = qh @ kh.transpose(0, 1, 3, 2) / np.float32(8.0) + addmask
        pr = probs(sc)
        mass, eff = measure(pr)
        masses.append(mass)
        effs.append(eff)
        ctx = (pr @ vh).transpose(0, 2, 1, 3).reshape(B, S, H)
        x1 = ln(ctx @ W(p + "attention.output.dense.weight").T + W(p + "attention.output.dense.bias") + x, W(p + "attention.output.LayerNorm.weight"),
                W(p + "attention.output.LayerNorm.bias"))
        hh = x1 @ W(p + "intermediate.dense.weight").T + W(p + "intermediate.dense.bias")
        hh = (hh * np.float32(0.5) * (np.float32(1.0) + _erf(hh / np.float32(np.sqrt(2.0))))).astype(np.float32)
        x = ln(hh @ W(p + "output.dense.weight").T + W(p + "output.dense.bias") + x1, W(p + "output.LayerNorm.weight"), W(p + "output.LayerNorm.bias"))
    return np.array(masses), np.array(effs), gains


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
