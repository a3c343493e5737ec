"""Per-kernel parity of the HIP path against the oracle's intermediate tensors (all through the C ABI)."""
import numpy as np
import pytest

from memvul_amd import synth
from oracle import memvul_oracle as orc

pytestmark = pytest.mark.gpu

L2 = dict(layers=2, vocab_size=2048)
WK = dict(qk_scale=4.0)


@pytest.fixture(scope="module")
def gu():
    import gpu_util
    return gpu_util


@pytest.mark.parametrize("variant", [0, 1, 2, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 30])
@pytest.mark.parametrize("shape", [(128, 128, 64), (256, 384, 768), (384, 256, 3072), (256, 256, 64), (512, 768, 128), (1024, 2304, 768)])
def test_gemm_variants(gu, variant, shape):
    """0: 128^2 tile LDS-DMA, 1: 128^2 register-staged, 2: 256^2 tile, 10+: LDS-ring variants (tile / BK /
    stages, see mv_test_gemm in engine.hip; 19 = the 64^2-tile skinny-M path of the [CLS] tail), 30: persistent kernel.  All accumulate over K in the same
    order, so they must agree bit-for-bit with each other (checked against variant 0) and with fp32 numpy to
    rounding."""
    M, N, K = shape
    if variant >= 2 and variant != 19 and (M % 256 or N % 256):
        pytest.skip("256^2 tile needs M, N % 256 == 0")
    if variant == 30 and K % 128:
        pytest.skip("the ping-pong kernel walks K two 64-wide tiles at a time")
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((M, K)).astype(np.float16)
    W = (rng.standard_normal((N, K)) * 0.05).astype(np.float16)
    bias = rng.standard_normal(N).astype(np.float32)
    eng = gu.engine_for(L2, WK)
    out, ms = eng.test_gemm(A, W, bias, variant=variant, iters=1)
    ref = A.astype(np.float32) @ W.astype(np.float32).T + bias
    err = float(np.abs(out - ref).max())
    gu.record("gemm", variant=variant, M=M, N=N, K=K, max_err=err)
    assert err < 2e-4 * np.sqrt(K / 64.0), f"gemm variant {variant} {shape}: max err {err}"
    if variant != 0:
        out0, _ = eng.test_gemm(A, W, bias, variant=0, iters=1)
        assert np.array_equal(out, out0), "tile variants must be bit-identical"


def _taps(gu, B, S, ragged):
    dims, w = gu.weights_for(L2, WK)
    ids, lens = synth.make_ids(B, S, dims.vocab_size, ragged=ragged)
    taps = {}
    mask = synth.mask_from_lens(lens, S)
    u = orc.instance_forward(w, ids.astype(np.int64), mask, taps=taps)
    return dims, w, ids, lens, mask, taps, u


@pytest.mark.parametrize("B,S,ragged", [(3, 64, False), (2, 128, True), (2, 100, True)])
def test_embeddings_layernorm(gu, B, S, ragged):
    dims, w, ids, lens, mask, taps, _ = _taps(gu, B, S, ragged)
    eng = gu.engine_for(L2, WK)
    eng.debug_encode(ids, lens, 0)
    x = eng.debug_read(0)[:, :S]
    err = float(np.abs(x - taps["embed"]).max())
    gu.record("embed_ln", B=B, S=S, max_err=err)
    assert err < 2e-5
    x16 = eng.debug_read(1)[:, :S].astype(np.float32)
    assert np.abs(x16 - taps["embed"]).max() < 4e-3


@pytest.mark.parametrize("attn", ["1", "0"])
@pytest.mark.parametrize("gemm_tile", [0, 512])
@pytest.mark.parametrize("B,S,ragged", [(2, 64, False), (3, 128, True), (2, 192, True), (2, 256, True), (1, 320, True), (2, 100, True),
                                        (2, 384, True), (2, 512, True), (1, 500, True)])
def test_layer0_stages(gu, B, S, ragged, gemm_tile, attn):
    """QKV projection, attention, FFN and both LayerNorms of encoder layer 0 against the oracle taps.
    Tolerances are fp16-operand level (inputs rounded to fp16, fp32 accumulation)."""
    dims, w, ids, lens, mask, taps, _ = _taps(gu, B, S, ragged)
    # 512: every projection through the persistent ping-pong GEMM; attn 1: attention_v2.h for Sp <= 256 and (in chunks of 128 keys) Sp = 384 / 512, 0: attention.h
    eng = gu.engine_for(L2, WK, gemm_tile=gemm_tile, env={"MEMVUL_ATTN": attn})
    eng.debug_encode(ids, lens, 1)
    q = eng.debug_read(2)[:, :, :S].astype(np.float32) * 8.0  # engine folds 1/sqrt(64) into W_q
    k = eng.debug_read(3)[:, :, :S].astype(np.float32)
    vt = eng.debug_read(4)[:, :, :, :S].astype(np.float32)
    ctx = eng.debug_read(5)[:, :S].astype(np.float32)
    h16 = eng.debug_read(6)[:, :S].astype(np.float32)
    x = eng.debug_read(0)[:, :S]
    m = mask  # compare real tokens only (padded query rows are never consumed)
    errs = dict(
        q=np.abs(q - taps["l0_q"])[np.broadcast_to(m[:, None, :, None], q.shape)].max(),
        k=np.abs(k - taps["l0_k"])[np.broadcast_to(m[:, None, :, None], k.shape)].max(),
        v=np.abs(vt.transpose(0, 1, 3, 2) - taps["l0_v"])[np.broadcast_to(m[:, None, :, None], k.shape)].max(),
        ctx=np.abs(ctx - taps["l0_ctx"])[m].max(),
        gelu=np.abs(h16 - taps["l0_gelu"])[m].max(),
        layer0=np.abs(x - taps["layer0"])[m].max(),
    )
    gu.record("layer0", B=B, S=S, gemm_tile=gemm_tile, attn=attn, **{k_: float(v_) for k_, v_ in errs.items()})
    scale_q = float(np.abs(taps["l0_q"]).max())
    assert errs["q"] < 3e-3 * max(1.0, scale_q), errs
    assert errs["k"] < 3e-3 * max(1.0, float(np.abs(taps["l0_k"]).max())), errs
    assert errs["v"] < 3e-3, errs
    assert errs["ctx"] < 8e-3, errs  # peaked attention (qk_scale=4): fp16 rounding of P and V
    assert errs["gelu"] < 4e-3, errs
    assert errs["layer0"] < 1e-2, errs


@pytest.mark.parametrize("attn", ["1", "0"])
@pytest.mark.parametrize("B,S", [(48, 256), (70, 128), (40, 192), (26, 512), (30, 384)])
def test_attention_persistent_item_loop(gu, B, S, attn):
    """attention_v2 walks (batch row, head[, query block]) units with a 2-deep LDS ring: more units than resident
    workgroups, uneven tails, ragged lengths, one to four key chunks per unit; checked against the oracle's layer-0
    context (attn 0: the same inputs through attention.h, whose error is the yardstick: peaked attention, fp16 P and V)."""
    dims, w, ids, lens, mask, taps, _ = _taps(gu, B, S, True)
    eng = gu.engine_for(L2, WK, env={"MEMVUL_ATTN": attn}, max_tokens=B * S, max_batch=B)
    eng.debug_encode(ids, lens, 1)
    ctx = eng.debug_read(5)[:, :S].astype(np.float32)
    err = float(np.abs(ctx - taps["l0_ctx"])[mask].max())
    gu.record("attention_items", B=B, S=S, attn=attn, max_err=err)
    assert err < 1.2e-2


def test_match_and_topk_vs_oracle(gu):
    rng = np.random.default_rng(5)
    dims, w = gu.weights_for(L2, WK)
    eng = gu.engine_for(L2, WK, max_anchors=128)
    for B, G in [(1, 1), (5, 7), (37, 124), (64, 64)]:
        u = np.maximum(rng.standard_normal((B, 512)), 0).astype(np.float32)
        v = np.maximum(rng.standard_normal((G, 512)), 0).astype(np.float32)
        if G > 3:
            v[3] = v[1]  # exact tie between anchors 1 and 3: the lower index must win
        eng.anchor_set(v)
        out = eng.match(u)
        logits, p, best, idx = orc.match(u, v, w[synth.KEY_MATCH_W], same_idx=0)
        e = float(np.abs(out["logits"] - logits).max())
        gu.record("match", B=B, G=G, max_err=e)
        assert e < 2e-5
        assert np.abs(out["probs"] - p).max() < 1e-5
        # GPU-side consistency (bit-exact): best is the row of probs at best_idx, best_idx the first arg-max
        ps = out["probs"][:, :, 0]
        assert np.array_equal(out["best_idx"], np.argmax(ps, axis=1).astype(np.int32))
        assert np.array_equal(out["best"], out["probs"][np.arange(B), out["best_idx"]])
        k = min(5, G)
        tp, ti = eng.topk(u, k)
        rp, ri = orc.topk_match(ps, k)
        assert np.array_equal(ti, ri.astype(np.int32)) and np.array_equal(tp, rp)
    eng.anchor_reset()


@pytest.mark.parametrize("B,G", [(256, 1000), (512, 1000), (37, 257), (300, 124), (3, 1024), (5, 128), (9, 129), (4, 256), (66, 513)])
def test_fused_match_topk_at_configs4_size(gu, B, G):
    """BASELINE.json configs[4]: 1000-anchor synthetic bank, fused match + top-k (k = 1, 5, 10): the k best anchors per
    issue report against the oracle (argsort of the oracle's own P(same), ties to the lower index), the best-anchor
    outputs of mv_match against the same kernel's full probabilities, exact duplicates in the bank (ties across chunk
    boundaries) and a NaN row (ranks first, index stays valid)."""
    rng = np.random.default_rng(G + B)
    dims, w = gu.weights_for(L2, WK)
    eng = gu.engine_for(L2, WK, max_anchors=1024, max_batch=512)
    u = np.maximum(rng.standard_normal((B, 512)), 0).astype(np.float32) * np.float32(0.5)
    v = synth.make_anchor_bank(G)
    if G > 300:
        v[290] = v[7]; v[G - 1] = v[7]   # the same anchor in three different 256-anchor chunks
    eng.anchor_set(v)
    out = eng.match(u)
    logits, p, best, idx = orc.match(u, v, w[synth.KEY_MATCH_W], same_idx=0)
    e = float(np.abs(out["logits"] - logits).max())
    gu.record("match_topk", B=B, G=G, max_logit_err=e)
    assert e < 2e-5 and np.abs(out["probs"] - p).max() < 1e-5
    ps = out["probs"][:, :, 0]
    assert np.array_equal(out["best_idx"], np.argmax(ps, axis=1).astype(np.int32))
    assert np.array_equal(out["best"], out["probs"][np.arange(B), out["best_idx"]])
    for k in (1, 5, 10, 64):  # 64 = MK_KMAX (chunks x k <= 1024, match_dev: four chunks x 64 = one merge pass of four candidates per lane)
        if k > G or ((G + 255) // 256) * k > 1024:
            continue
        tp, ti = eng.topk(u, k)
        rp, ri = orc.topk_match(ps, k)          # the GPU's own probabilities: the selection must be exact
        assert np.array_equal(ti, ri.astype(np.int32)) and np.array_equal(tp, rp), k
        op, oi = orc.topk_match(p[:, :, 0], k)  # and against the oracle's probabilities wherever the margins are clear
        srt = -np.sort(-p[:, :, 0], axis=1)[:, :k + 1]
        clear = (np.abs(np.diff(srt, axis=1)) > 1e-5).all(1) if G > k else np.ones(B, bool)
        assert np.array_equal(ti[clear], oi[clear].astype(np.int32))
    if G > 300:
        assert (np.diff(np.stack([ps[:, 7], ps[:, 290], ps[:, G - 1]]), axis=0) == 0).all()  # duplicates score identically
    # a NaN issue report: every score NaN -> torch.argmax semantics (index 0), no out-of-range index
    un = u[:2].copy()
    un[0, 5] = np.nan
    o2 = eng.match(un)
    assert o2["best_idx"][0] == 0 and np.isnan(o2["best"][0]).all() and o2["best_idx"][1] == out["best_idx"][1]
    eng.anchor_reset()
