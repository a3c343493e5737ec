"""Numerical safety of the virtual LayerNorm's one-pass statistics ("vstats", common.h ln_from_partials / gemm_pp.h): the
residual GEMMs emit per row three fp32 (sum, sum of squares) pairs — each the fixed-order sum of four 64-column wave shares —
and every consumer forms  mean = S1 / 768,  var = S2 / 768 - mean^2,  rstd = rsq(var + eps).  E[x^2] - mean^2 cancels when
|mean| >> std; this emulates the arithmetic in float32 on rows like the encoder's (O(1) values, trained-checkpoint-like outlier
dimensions, a large common offset) and bounds the error of (mean, rstd) against float64 two-pass statistics."""
import numpy as np
import pytest

H, EPS = 768, 1e-12


def vstats_f32(x):
    """x [rows, 768] float32 -> (mean, rstd) the way the kernels compute them."""
    x = x.astype(np.float32)
    parts = []
    for t in range(3):  # 256-column tile
        waves = []
        for w in range(4):  # 64-column wave share: per lane 32 values (two fragments), then the two half-waves are added
            blk = x[:, t * 256 + w * 64: t * 256 + (w + 1) * 64]
            s1 = np.zeros(len(x), np.float32)
            s2 = np.zeros(len(x), np.float32)
            for half in range(2):
                h1 = np.zeros(len(x), np.float32)
                h2 = np.zeros(len(x), np.float32)
                for c in range(32):
                    v = blk[:, half * 32 + c]
                    h1 = (h1 + v).astype(np.float32)
                    h2 = (v.astype(np.float64) * v + h2).astype(np.float32)  # fma: one rounding
                s1, s2 = (s1 + h1).astype(np.float32), (s2 + h2).astype(np.float32)
            waves.append((s1, s2))
        p1 = ((waves[0][0] + waves[1][0]).astype(np.float32) + waves[2][0]).astype(np.float32) + waves[3][0]
        p2 = ((waves[0][1] + waves[1][1]).astype(np.float32) + waves[2][1]).astype(np.float32) + waves[3][1]
        parts.append((p1.astype(np.float32), p2.astype(np.float32)))
    S1 = ((parts[0][0] + parts[1][0]).astype(np.float32) + parts[2][0]).astype(np.float32)
    S2 = ((parts[0][1] + parts[1][1]).astype(np.float32) + parts[2][1]).astype(np.float32)
    mean = (S1 * np.float32(1.0 / H)).astype(np.float32)
    var = np.maximum((S2 * np.float32(1.0 / H)).astype(np.float32) - (mean * mean).astype(np.float32), np.float32(0))
    rstd = (1.0 / np.sqrt(var.astype(np.float64) + EPS)).astype(np.float32)  # v_rsq_f32: 1 ulp, not modelled
    return mean, rstd


@pytest.mark.parametrize("kind,tol", [("unit", 2e-6), ("outliers", 2e-6), ("offset3", 2e-5)])
def test_one_pass_statistics_hold_on_encoder_like_rows(kind, tol):
    rng = np.random.default_rng(7)
    x = rng.standard_normal((512, H))
    if kind == "outliers":  # two dimensions carry +-20 sigma, as trained BERT checkpoints do (synth trained_like: -4 / +3 after LN)
        x[:, H // 3 + 52] += 20.0
        x[:, H // 2 - 3] -= 15.0
    if kind == "offset3":  # a common offset of 3 sigma: the cancellation case, an order of magnitude looser but still far below fp16's 5e-4
        x += 3.0
    mean, rstd = vstats_f32(x)
    x64 = x.astype(np.float32).astype(np.float64)
    m64 = x64.mean(1)
    r64 = 1.0 / np.sqrt(x64.var(1) + EPS)
    assert np.abs(mean - m64).max() <= tol * max(1.0, np.abs(m64).max())
    assert (np.abs(rstd - r64) / r64).max() <= tol


def test_two_class_softmax_from_the_logit_difference():
    """The fused matcher (match_topk.h) accumulates delta = logit_0 - logit_1 as ONE float32 chain with the class-difference
    weights and derives softmax_2 from it: against the reference's two separate sums (model_memory.py:141-142, float64) the
    probabilities agree to float32 rounding, also at |logit| ~ 3 where the verdict's 1e-3 budget is tight."""
    rng = np.random.default_rng(11)
    B, G, P = 64, 124, 512
    u = np.maximum(rng.standard_normal((B, P)), 0).astype(np.float32)
    v = np.maximum(rng.standard_normal((G, P)), 0).astype(np.float32)
    Wm = (rng.standard_normal((2, 3 * P)) * 0.1).astype(np.float32)  # [W_a | W_b | W_c] per class; |logit| up to ~5 here
    feat = np.concatenate([np.broadcast_to(u[:, None], (B, G, P)), np.broadcast_to(v[None], (B, G, P)),
                           np.abs(u[:, None] - v[None])], -1).astype(np.float64)
    logits = feat @ Wm.astype(np.float64).T
    ref = np.exp(logits - logits.max(-1, keepdims=True))
    ref /= ref.sum(-1, keepdims=True)
    wd = (Wm[0] - Wm[1]).astype(np.float32)  # rounded once, as the kernel's operand staging does
    delta = np.zeros((B, G), np.float32)
    for k in range(3 * P):  # one fma per feature, ascending
        delta = (feat[:, :, k] * np.float64(wd[k]) + delta).astype(np.float32)
    ed = np.exp(-np.abs(delta).astype(np.float64))
    p0 = np.where(delta >= 0, 1.0 / (1.0 + ed), ed / (1.0 + ed))
    assert np.abs(logits).max() > 3.0
    assert np.abs(p0 - ref[..., 0]).max() < 2e-6
    assert np.abs((logits[..., 0] - logits[..., 1]) - delta).max() < 2e-5


def _raster_pp(L, G, tm_count, tn_count, GN, mode):
    """gemm_pp.h raster_pp restated: logical tile L of a persistent grid of G workgroups -> (tile_m, tile_n)."""
    if mode == 1:
        ngroups = tn_count // GN
        w, r = divmod(L, G)
        b, g = divmod(w, ngroups)
        p = b * G + r
        tm = p // GN
        return tm, g * GN + (p - tm * GN)
    per_group = tm_count * GN
    g, r = divmod(L, per_group)
    tm = r // GN
    return tm, g * GN + (r - tm * GN)


def test_a_stationary_raster_is_a_bijection_and_keeps_tile_m_across_the_column_groups():
    """MEMVUL_RASTER=1 (gemm_pp.h raster_pp mode 1; the host enables it only when tm_count * GN % G == 0): every output tile is
    visited exactly once, and a workgroup's consecutive persistent iterations walk the column groups of the SAME tile_m before it
    moves on — what lets an XCD's A panels stay in its L2 across the whole N sweep."""
    for tm_count, tn_count, GN, G in [(256, 12, 4, 256), (256, 9, 3, 256), (64, 12, 4, 256), (128, 12, 4, 256)]:
        assert (tm_count * GN) % G == 0
        n = tm_count * tn_count
        for mode in (0, 1):
            seen = {_raster_pp(L, G, tm_count, tn_count, GN, mode) for L in range(n)}
            assert len(seen) == n and all(0 <= a < tm_count and 0 <= b < tn_count for a, b in seen), (tm_count, tn_count, mode)
        ngroups = tn_count // GN
        for slot in (0, 31, 32, 255):
            its = [_raster_pp(it * G + slot, G, tm_count, tn_count, GN, 1) for it in range(n // G)]
            for k in range(0, len(its) - ngroups + 1, ngroups):
                run = its[k:k + ngroups]
                assert len({tm for tm, _ in run}) == 1 and [tn // GN for _, tn in run] == list(range(ngroups)), (slot, run)
