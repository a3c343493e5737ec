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


@pytest.mark.parametrize("variant", [0, 19])
@pytest.mark.parametrize("shape", [(128, 128, 64), (256, 384, 768), (384, 256, 3072), (256, 256, 64), (512, 768, 128), (1024, 2304, 768)])
def test_gemm_variants(gu, variant, shape):
    """0: the 128^2-tile LDS-DMA kernel of small passes, 19: the 64^2-tile ring kernel of the [CLS] tail (mv_test_gemm).  Both
    accumulate over K in the same order, so they must agree bit-for-bit with each other and with fp32 numpy to rounding.
    (The ring-geometry sweep, the 256^2 one-tile kernel and the register-staged form of rounds 1-2 are retired (git history).)"""
    M, N, K = shape
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


def _gelu64(x):
    from math import erf
    return 0.5 * x * (1.0 + np.vectorize(erf)(x / np.sqrt(2.0)))


@pytest.mark.parametrize("shape", [(256, 256, 256), (512, 768, 768), (256, 3072, 768), (768, 768, 3072), (16384, 768, 768)])
def test_persistent_gemm_fp16_and_fp8_correction_sweep(gu, shape):
    """gemm_pp_kernel<PP_GELU> on caller data (mv_test_gemm_pp), plain and as the MV_F16X8 build.  The reference is the
    float64 product of the UNROUNDED operands; what is checked is the operand precision each build delivers:
      * MV_F16: pre-activation error at the fp16-operand level (2^-12 per operand, ~sqrt(K) accumulation);
      * MV_F16X8: the fp8 correction sweep (v_mfma_scale_f32_32x32x64_f8f6f4, A_lo8 W_hi8 + A_hi8 W_lo8 over a virtual K of
        2 K) must remove most of it — this is also the test of the instruction's operand layout as the kernel uses it
        (fragment chunks (2 ks + hi) of a 128-byte row, uniform E8M0 scales): a wrong byte order or scale shows up as NO
        improvement or as garbage — and the [lo8 | hi8] planes of the output must decode to the output itself."""
    from memvul_amd.binding import e4m3_decode
    from oracle import precision_model as pm

    M, N, K = shape
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    A = (rng.standard_normal((M, K)) * (1.0 + 3.0 * (rng.random((1, K)) < 0.01))).astype(np.float32)  # a few outlier columns
    W = (rng.standard_normal((N, K)) * 0.04).astype(np.float32)
    bias = (rng.standard_normal(N) * 0.1).astype(np.float32)
    eng = gu.engine_for(L2, WK)
    pre = A.astype(np.float64) @ W.astype(np.float64).T + bias
    ref = _gelu64(pre)
    scale = float(np.sqrt((pre ** 2).mean()))
    half_ulp16 = np.abs(ref) * 2.0 ** -11 + 2.0 ** -25  # the fp16 rounding of the stored output itself
    out, _, ms = eng.test_gemm_pp(A, W, bias, x8=False)
    e16 = np.abs(out.astype(np.float64) - ref) - half_ulp16
    out_x, planes, ms8 = eng.test_gemm_pp(A, W, bias, x8=True)
    # the stored output decoded from its planes: hi16 + lo8 / 2^(11 + s)  (s = MV_X8_ACT_SHIFT = 2)
    lo8 = e4m3_decode(planes[:, :N]).astype(np.float64) / 2.0 ** 13
    hi8 = e4m3_decode(planes[:, N:]).astype(np.float64) / 4.0
    full = out_x.astype(np.float64) + lo8
    ex = np.abs(full - ref)
    # the same three-term product formed on the CPU (oracle/precision_model.py: fp16 / e4m3 planes, exact accumulation)
    emul = _gelu64(pm._mm("f16x8", "f16x8", A.astype(np.float64), W.astype(np.float64)) + bias)
    em = np.abs(full - emul)
    lo8_res = np.abs(ref) * 2.0 ** -15 + 2.0 ** -22       # what the 4-bit lo8 plane leaves of the output's own rounding
    gu.record("gemm_pp", M=M, N=N, K=K, err_f16=float(e16.max()), err_x8=float(ex.max()), x8_vs_cpu_emulation=float(em.max()),
              emulation_vs_exact=float(np.abs(emul - ref).max()), rms_pre=scale, ms_f16=ms, ms_x8=ms8)
    assert np.isfinite(full).all()
    assert e16.max() < 1.5e-3 * scale, float(e16.max())                 # fp16 operands: 2^-12 each, ~5 sigma over M N outputs
    assert (em <= lo8_res + 3e-5 * scale).all(), float(em.max())        # the kernel forms exactly the modelled product (fp32 sums)
    assert (ex <= lo8_res + 2.5e-4 * scale).all(), float(ex.max())      # ... which is ~2^-15.5 operands
    assert ex.max() < 0.5 * max(e16.max(), 1e-4 * scale)                # several times below the fp16 build on the same data
    assert np.abs(hi8 - out_x.astype(np.float64)).max() <= np.abs(out_x.astype(np.float64)).max() * 2.0 ** -4 + 2.0 ** -11
    # x8_terms = 1 (the QKV projection's form since round 4): the fp8 sweep covers only the weight-side term A_hi8 W_lo8 — the second
    # halves of both operand rows, K / 128 K-tiles.  Against the SAME product formed on the CPU: agreement at the fp32-accumulation
    # level; a wrong half / K-tile count / byte offset shows up as the fp16 build's error or as garbage.
    if K % 256 == 0:
        out_w, _, ms_w = eng.test_gemm_pp(A, W, bias, x8=2)
        emul_w = _gelu64(pm._mm("f16x8w", "f16x8", A.astype(np.float64), W.astype(np.float64)) + bias)
        ew = np.abs(out_w.astype(np.float64) - emul_w) - (np.abs(emul_w) * 2.0 ** -11 + 2.0 ** -25)
        both = _gelu64(pm._mm("f16x8", "f16x8", A.astype(np.float64), W.astype(np.float64)) + bias)
        gu.record("gemm_pp_wside", M=M, N=N, K=K, vs_cpu_emulation=float(ew.max()), wside_vs_both=float(np.abs(emul_w - both).max()), ms=ms_w, ms_both=ms8)
        assert ew.max() <= 3e-5 * scale, float(ew.max())


def test_a_stationary_raster_gives_identical_bits(gu):
    """MEMVUL_RASTER=1 / MEMVUL_GN_MAX (the raster A/B switches of round 4): which workgroup computes a tile, and when, must not
    change a single bit of it — the FFN-1 kernel at a shape where the A-stationary raster applies (tm_count * GN % grid == 0)."""
    M, N, K = 16384, 3072, 768
    rng = np.random.default_rng(99)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) * 0.04).astype(np.float32)
    bias = (rng.standard_normal(N) * 0.1).astype(np.float32)
    base, _, _ = gu.engine_for(L2, WK).test_gemm_pp(A, W, bias, x8=False)
    base8, planes8, _ = gu.engine_for(L2, WK).test_gemm_pp(A, W, bias, x8=True)
    for env in ({"MEMVUL_RASTER": "1"}, {"MEMVUL_GN_MAX": "12"}, {"MEMVUL_RASTER": "1", "MEMVUL_GN_MAX": "6"}):
        eng = gu.engine_for(L2, WK, env=env)
        out, _, _ = eng.test_gemm_pp(A, W, bias, x8=False)
        assert np.array_equal(out.view(np.uint16), base.view(np.uint16)), env
        out8, p8, _ = eng.test_gemm_pp(A, W, bias, x8=True)
        assert np.array_equal(out8.view(np.uint16), base8.view(np.uint16)) and np.array_equal(p8, planes8), env


def _taps(gu, B, S, ragged):
    dims, w = gu.weights_for(L2, WK)
    ids, lens = synth.make_ids(B, S, dims.vocab_size, ragged=ragged)
    taps = {}
    mask = synth.mask_from_lens(lens, S)
    u = orc.instance_forward(w, ids.astype(np.int64), mask, taps=taps)
    return dims, w, ids, lens, mask, taps, u


@pytest.mark.parametrize("compute", ["precise", "f16"])
@pytest.mark.parametrize("B,S,ragged", [(3, 64, False), (2, 128, True), (2, 100, True)])
def test_embeddings_layernorm(gu, B, S, ragged, compute):
    """(precise: the raw two-plane stream + vstats of the persistent path, normalised by the final LayerNorm kernel; f16 at this size: the fused embedding LayerNorm)"""
    dims, w, ids, lens, mask, taps, _ = _taps(gu, B, S, ragged)
    eng = gu.engine_for(L2, WK, compute_dtype=compute)
    eng.debug_encode(ids, lens, 0)
    x = eng.debug_read(0)[:, :S]
    e_row = np.abs(x - taps["embed"]).max(-1)
    err = float(e_row.max())
    gu.record("embed_ln", B=B, S=S, compute=compute, max_err=err)
    if compute == "precise":
        # MV_F16X8 keeps the raw stream as hi fp16 + the lo8 plane of its fp8 planes (2^-15 of the element; gemm.h GemmArgs::out16b) — except the special rows (the [CLS]
        # and the last token of every sequence), which keep hi + lo: their stream reaches the pooler un-averaged
        sp = np.zeros(e_row.shape, bool)
        for b_ in range(B):
            sp[b_, 0] = sp[b_, lens[b_] - 1] = True
        assert float(e_row[sp].max()) < 2e-5 and err < 1.5e-4, (float(e_row[sp].max()), err)  # measured 5.2 - 5.7e-5
    else:
        assert err < 2e-5
    x16 = eng.debug_read(1)[:, :S].astype(np.float32)
    assert np.abs(x16 - taps["embed"]).max() < 4e-3


@pytest.mark.parametrize("gemm_tile,compute", [(0, "f16"), (512, "f16"), (0, "precise")])
@pytest.mark.parametrize("B,S,ragged", [(2, 64, False), (3, 128, True), (2, 192, True), (2, 256, True), (1, 320, True), (2, 100, True),
                                        (2, 384, True), (2, 512, True), (1, 500, True)])
def test_layer0_stages(gu, B, S, ragged, gemm_tile, compute):
    """QKV projection, attention, FFN and both LayerNorms of encoder layer 0 against the oracle taps.
    Tolerances are fp16-operand level (inputs rounded to fp16, fp32 accumulation)."""
    dims, w, ids, lens, mask, taps, _ = _taps(gu, B, S, ragged)
    # gemm_tile 0: these passes are small -> the small-pass kernels; 512: the persistent kernels (virtual LayerNorm, two-plane
    # stream) forced; "precise": MV_F16X8, always persistent.  Attention: attention_v2 for every padded length (64-key blocks
    # up to 256, 128-key chunks at 384 / 512; 320 and 500 run padded to 384 / 512)
    eng = gu.engine_for(L2, WK, gemm_tile=gemm_tile, compute_dtype=compute)
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
    gu.record("layer0", B=B, S=S, gemm_tile=gemm_tile, compute=compute, **{k_: float(v_) for k_, v_ in errs.items()})
    scale_q = float(np.abs(taps["l0_q"]).max())
    assert errs["q"] < 3e-3 * max(1.0, scale_q), errs
    assert errs["k"] < 3e-3 * max(1.0, float(np.abs(taps["l0_k"]).max())), errs
    # bounds = ~1.5x the largest error measured over all nine shapes and both GEMM paths (profiles/r04_f_parity_records.jsonl:
    # v 1.5e-3, ctx 6.8e-3 (MV_F16) / 4.0e-3 (precise), gelu 1.6e-3, layer0 1.6e-3): a regression of a few fp16 ulps in one stage shows here
    assert errs["v"] < 2.5e-3, errs
    assert errs["ctx"] < (6e-3 if compute == "precise" else 9e-3), errs  # peaked attention (qk_scale=4): fp16 rounding of P and V
    assert errs["gelu"] < 2.5e-3, errs
    assert errs["layer0"] < 2.5e-3, errs


@pytest.mark.parametrize("compute", ["f16", "precise"])
@pytest.mark.parametrize("B,S", [(48, 256), (70, 128), (40, 192), (26, 512), (30, 384), (21, 320)])
def test_attention_persistent_item_loop(gu, B, S, compute):
    """attention_v2 walks (batch row, head[, query block]) units with a 2-deep LDS ring: more units than resident
    workgroups, uneven tails, ragged lengths, one to four key chunks per unit; checked against the oracle's layer-0
    context ("precise": the instantiations that also write the fp8 planes of the context)."""
    dims, w, ids, lens, mask, taps, _ = _taps(gu, B, S, True)
    Sp = (S + 63) // 64 * 64 if S <= 256 else (S + 127) // 128 * 128
    eng = gu.engine_for(L2, WK, compute_dtype=compute, max_tokens=B * Sp, max_batch=B)
    eng.debug_encode(ids, lens, 1)
    ctx = eng.debug_read(5)[:, :S].astype(np.float32)
    err = float(np.abs(ctx - taps["l0_ctx"])[mask].max())
    gu.record("attention_items", B=B, S=S, compute=compute, max_err=err)
    assert err < (8e-3 if compute == "precise" else 1.2e-2)  # measured 5.4e-3 / 9.4e-3 (profiles/r04_f_parity_records.jsonl)


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
