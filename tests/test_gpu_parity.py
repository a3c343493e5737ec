"""End-to-end parity of the HIP path (C ABI) with the committed golden vectors and the oracle, plus
size-independent properties at BASELINE.json's full batch size."""
import os

import numpy as np
import pytest

from memvul_amd import synth
from oracle import memvul_oracle as orc

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-3  # north_star: logits within 1e-3 of the CPU reference
# Trained-like regime (|logit| ~ 3 through a matcher of norm ~ 17): what fp16 MFMA operands deliver, MEASURED — 3.3e-3 ..
# 5.7e-3 on the MI355X (profiles/r02_a_trained_like.txt), 2.5e-3 .. 3.2e-3 in the float64 rounding model
# (tests/test_precision_model.py), i.e. the 1e-3 budget is NOT met by MV_F16 here.  The compute dtype that meets it is
# MV_F16X8 ("precise": + one fp8 correction sweep per GEMM): test_precise_mode_holds_1e3_in_the_trained_like_regime
# asserts LOGIT_TOL for it.  The probabilities — what thresholds, decisions and every metric of the path consume — stay
# within 1e-4 in either mode because softmax_2 is flat at |logit| ~ 3.
TRAINED_LIKE_LOGIT_BOUND = 8e-3  # MV_F16 only (measured bound, not the target)
TRAINED_LIKE_P_TOL = 2e-4


@pytest.fixture(scope="module")
def gu():
    import gpu_util
    return gpu_util


PATHS = [(0, "precise"), (0, "f16"), (512, "f16")]  # (MEMVUL_GEMM_TILE, compute dtype): the product default (always the persistent kernels), then the opt-in
                                                     # MV_F16 on the path its pass size selects and with the bench-scale kernels forced at test sizes


@pytest.mark.parametrize("gemm_tile,compute", PATHS)
@pytest.mark.parametrize("name", ["l2_peaky_full", "l2_ragged", "l12_base_ragged", "l12_base_s256"])
def test_golden_logits(gu, golden_dir, name, gemm_tile, compute):
    import make_golden
    g = np.load(os.path.join(golden_dir, f"{name}.npz"))
    dk, wk, B, S, ragged, G, SA = make_golden.CASES[name]
    eng = gu.engine_for(dk, wk, gemm_tile=gemm_tile, compute_dtype=compute)
    eng.anchor_reset()
    LA = int(g["anchor_lens"].max())
    eng.anchor_append(g["anchor_ids"][:, :LA], g["anchor_lens"])  # one chunk padded to its longest (predict_memory.py:81)
    v = eng.anchor_get()
    out = eng.forward(g["ids"], g["lens"], want_embed=True)
    errs = dict(
        v=float(np.abs(v - g["v"]).max()), u=float(np.abs(out["embed"] - g["u"]).max()),
        logits=float(np.abs(out["logits"] - g["logits"]).max()), p=float(np.abs(out["probs"] - g["p"]).max()),
        logit_scale=float(np.abs(g["logits"]).max()),
    )
    gu.record("golden", case=name, gemm_tile=gemm_tile, compute=compute, **errs)
    assert errs["logits"] <= LOGIT_TOL, errs
    assert errs["p"] <= LOGIT_TOL, errs
    # decisions: best-anchor index agrees wherever the reference's top-2 margin exceeds the tolerance
    ps = g["p"][:, :, 0]
    srt = np.sort(ps, axis=1)
    clear = (srt[:, -1] - srt[:, -2]) > 2 * LOGIT_TOL if ps.shape[1] > 1 else np.ones(len(ps), bool)
    assert np.array_equal(out["best_idx"][clear], g["idx"][clear].astype(np.int32))
    assert np.abs(out["best"] - g["best"])[clear].max() <= LOGIT_TOL if clear.any() else True
    eng.anchor_reset()


@pytest.mark.parametrize("gemm_tile", [0, 512])
@pytest.mark.parametrize("name", ["l12_trained_s256", "l12_trained_ragged"])
def test_trained_like_logits(gu, golden_dir, name, gemm_tile):
    """The 1e-3 logit tolerance where it is hardest: 12 layers, peaked attention, outlier
    dimensions in every LayerNorm, |u| = O(1) and a matcher scaled so that max |logit| ~ 3 (the regime a checkpoint
    trained at temperature 0.1, config_memory.json:38, lives in).  Goldens: HF BertModel fp32 + the reference's head
    (tests/golden/make_golden.py); issue reports of 256 tokens (cfg-2 shape) against anchors of up to 320 / 512 tokens,
    on both GEMM paths."""
    import make_golden
    g = np.load(os.path.join(golden_dir, f"{name}.npz"))
    dk, wk, B, S, ragged, G, SA = make_golden.CASES[name]
    eng = gu.engine_for(dk, wk, gemm_tile=gemm_tile, compute_dtype="f16", max_tokens=16384, max_batch=64, max_anchors=64)  # the opt-in mode and its measured bound
    eng.anchor_reset()
    LA = int(g["anchor_lens"].max())
    eng.anchor_append(g["anchor_ids"][:, :LA], g["anchor_lens"])
    v = eng.anchor_get()
    out = eng.forward(g["ids"], g["lens"], want_embed=True)
    errs = dict(
        v=float(np.abs(v - g["v"]).max()), u=float(np.abs(out["embed"] - g["u"]).max()),
        logits=float(np.abs(out["logits"] - g["logits"]).max()), p=float(np.abs(out["probs"] - g["p"]).max()),
        logit_scale=float(np.abs(g["logits"]).max()), u_scale=float(np.abs(g["u"]).max()),
    )
    gu.record("trained_like", case=name, gemm_tile=gemm_tile, **errs)
    assert errs["logit_scale"] > 2.5
    assert errs["logits"] <= TRAINED_LIKE_LOGIT_BOUND, errs  # NOT the 1e-3 target: the measured level, see above
    assert errs["p"] <= TRAINED_LIKE_P_TOL, errs
    assert errs["u"] <= 1e-3 and errs["v"] <= 1e-3, errs     # embeddings of magnitude 0.8: 6e-4 relative
    eng.anchor_reset()


PRECISE_TRAINED_LIKE_REGRESSION_BOUND = 6.2e-4  # the shipped default on the trained-like goldens measures 5.75 / 3.5e-4 (round 6, after every fp16 store of the GEMM epilogues
                                                 # rounds the fp32 value it names: gemm_pp.h pin_f32x4; before that re-draw of the roundings 4.5 / 3.5e-4, rounds 5 - 6a 3.6 - 4.0 /
                                                 # 3.5 - 5.0e-4 — ONE golden is one draw: the 24-draw distribution did not move, median 3.40 -> 3.38e-4, pooled rms 1.32 -> 1.28e-4,
                                                 # profiles/r06_*_error_distribution.txt): a bound that catches erosion of the margin to the 1e-3 contract, not the contract itself


@pytest.mark.parametrize("name", ["l12_trained_s256", "l12_trained_ragged", "l12_base_ragged", "l12_base_s256", "l2_peaky_full", "l2_ragged"])
def test_precise_mode_holds_1e3_in_the_trained_like_regime(gu, golden_dir, name):
    """MV_F16X8 (compute dtype "precise"): every GEMM of the encoder adds ONE correction sweep on the fp8 matrix path —
    A_lo8 W_hi8 + A_hi8 W_lo8, the first-order terms of the split-operand product in OCP e4m3 (in the default [CLS]-row form: the second term in every
    row, the first for the [CLS] rows alone: test_cls_row_aside_form) — to its fp16 sweep
    (gemm_pp.h X8; Q / K / V and P stay fp16): the engine change that meets the 1e-3 logit tolerance where plain fp16
    operands measure 3 - 6e-3 (test_trained_like_logits).  oracle/precision_model.py predicts 3.4e-4 for this configuration
    (tests/test_precision_model.py::test_fp8_correction_sweeps_hold_the_budget); the bench line's `precise` object carries its
    rate.  The mode always runs the persistent kernels (both GEMM paths of the other tests collapse into one here)."""
    import make_golden
    from memvul_amd.binding import MV_F16X8

    g = np.load(os.path.join(golden_dir, f"{name}.npz"))
    dk, wk, B, S, ragged, G, SA = make_golden.CASES[name]
    eng = gu.engine_for(dk, wk, compute_dtype=MV_F16X8, max_tokens=16384, max_batch=64, max_anchors=64)
    eng.anchor_reset()
    LA = int(g["anchor_lens"].max())
    eng.anchor_append(g["anchor_ids"][:, :LA], g["anchor_lens"])
    v = eng.anchor_get()
    out = eng.forward(g["ids"], g["lens"], want_embed=True)
    errs = dict(v=float(np.abs(v - g["v"]).max()), u=float(np.abs(out["embed"] - g["u"]).max()),
                logits=float(np.abs(out["logits"] - g["logits"]).max()), p=float(np.abs(out["probs"] - g["p"]).max()),
                logit_scale=float(np.abs(g["logits"]).max()))
    gu.record("precise_mode", case=name, **errs)
    assert errs["logits"] <= (PRECISE_TRAINED_LIKE_REGRESSION_BOUND if "trained" in name else LOGIT_TOL), errs
    assert errs["p"] <= 1e-4, errs
    eng.anchor_reset()


@pytest.mark.parametrize("qkv_aside", ["q", "none"])
@pytest.mark.parametrize("name", ["l12_trained_s256", "l12_trained_ragged"])
def test_cls_row_aside_form(gu, golden_dir, name, qkv_aside):
    """MEMVUL_CLS_ASIDE (round 5; 1 is the default): sequences of >= 128 tokens in passes of padded length 256 / 512 sweep the weight-side correction term only and get
    the A-side term A_lo W_hi^T for their [CLS] row alone — cls_lo_gather_kernel + a skinny fp16 GEMM over the B rows in front of the persistent
    GEMMs, added to those rows' accumulators (gemm_pp.h GemmArgs::cls_corr) — because only that row reaches the pooler un-averaged
    (oracle/precision_model.py knob `cls_fix`; tests/test_precision_model.py::test_cls_row_aside_is_priced_by_the_model).  With the term missing or
    misplaced the logits would sit at the weight-side-only level (2.7e-3, profiles/r04_a2_*): the contract bound below is the functional test.
    `qkv_aside` = "q": the Q block of the QKV projection keeps its A-side term for every row (the default of rounds 4 - 6a), "none": no block does (the default
    since: the special rows take the term from the QKV projection's row term in all three blocks)."""
    import make_golden

    g = np.load(os.path.join(golden_dir, f"{name}.npz"))
    dk, wk, B, S, ragged, G, SA = make_golden.CASES[name]
    env = {"MEMVUL_CLS_ASIDE": "1", "MEMVUL_QKV_ASIDE": qkv_aside}
    eng = gu.engine_for(dk, wk, compute_dtype="precise", env=env, max_tokens=16384, max_batch=64, max_anchors=64)
    eng.anchor_reset()
    LA = int(g["anchor_lens"].max())
    eng.anchor_append(g["anchor_ids"][:, :LA], g["anchor_lens"])
    out = eng.forward(g["ids"], g["lens"], want_embed=True)
    errs = dict(u=float(np.abs(out["embed"] - g["u"]).max()), logits=float(np.abs(out["logits"] - g["logits"]).max()),
                p=float(np.abs(out["probs"] - g["p"]).max()), logit_scale=float(np.abs(g["logits"]).max()))
    gu.record("precise_mode_cls_aside", case=name, qkv_aside=qkv_aside, **errs)
    assert errs["logits"] <= LOGIT_TOL and errs["p"] <= 2e-4, errs
    assert eng.x8_saturation() == 0
    # the same model with both terms in every row: the two differ, by less than the contract
    ref = gu.engine_for(dk, wk, compute_dtype="precise", env={"MEMVUL_CLS_ASIDE": "0"}, max_tokens=16384, max_batch=64, max_anchors=64)
    ref.anchor_reset()
    ref.anchor_append(g["anchor_ids"][:, :LA], g["anchor_lens"])
    o2 = ref.forward(g["ids"], g["lens"])
    d = float(np.abs(o2["logits"] - out["logits"]).max())
    assert 0 < d <= LOGIT_TOL, d
    eng.anchor_reset(); ref.anchor_reset()


SINK_BOUND = 6e-4  # delimiter sinks on the shipped default: measured 1.9 .. 4.2e-4 on the five committed draws below (1.7 .. 8.9e-4 over all 36: profiles/r06_n_sink_envelope.txt); the contract is LOGIT_TOL


@pytest.mark.parametrize("case", ["sep_all_80_3001", "sep_all_95_3001", "sep_cls_80_3002", "cls_all_80_3001", "sep_all_50_3003"])
def test_attention_sinks_on_the_delimiter_tokens_hold_the_contract(gu, golden_dir, case):
    """Round 6 (VERDICT r5 next #1b): the attention-concentration axis of the precision envelope.  Trained BERT heads put most of their mass on [SEP] / [CLS] (1 - 4
    effective keys of 256); the random-init family of the other tests spreads the [CLS] row over 67 - 149.  synth.apply_sink writes such a sink into the weights —
    every head of every layer puts 50 / 80 / 95 % of the mass of every row ("all") or of the [CLS] row ("cls") on the sequence's [SEP] or [CLS] token, measured
    on the CPU oracle — and tests/golden/r06_sink_refs.npz holds the CPU reference logits (scripts/r06_make_sink_refs.py; 8 issue reports x 256 tokens against 6
    anchors of up to 512 tokens, trained-like, |logit| up to 5).  Round 5's default measured 1.8 - 3.4e-3 here in the float64 model (the sink row's A-side
    roundings and the fp16 storage of its V reach every row un-averaged); the special rows of round 6 (gemm_pp.h / attention_v2.h: rows 0 and 1 of every sequence
    hold [CLS] and [SEP], take the row terms in every GEMM and carry V as hi + lo) hold the contract with margin."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import r06_make_sink_refs as mk6
    from memvul_amd.binding import Engine

    refs = np.load(os.path.join(golden_dir, "r06_sink_refs.npz"))
    if case + "_lg" not in refs.files:
        pytest.skip("no CPU reference for this case in tests/golden/r06_sink_refs.npz")
    token, rows, pct, seed = case.split("_")
    dims, w, ids, lens, aids, alens, _ = mk6.case(token, rows, int(pct) / 100.0, int(seed), gains=refs[case + "_gains"])
    eng = Engine(0, vocab_size=dims.vocab_size, layers=12, max_tokens=16 * 512, max_batch=16, max_anchors=16)
    try:
        eng.load_state_dict(w)  # the product default
        LA = int(alens.max())
        eng.anchor_append(aids[:, :LA], alens)
        out = eng.forward(ids, lens)
        e = float(np.abs(out["logits"] - refs[case + "_lg"]).max())
        gu.record("attention_sink", case=case, logits_err=e, logit_scale=float(np.abs(refs[case + "_lg"]).max()),
                  mass=float(refs[case + "_stat"][0]), eff_keys=float(refs[case + "_stat"][1]))
        assert refs[case + "_stat"][1] < 12            # the [CLS] row looks at a handful of keys
        assert e <= SINK_BOUND, e
        assert eng.x8_saturation() == 0
    finally:
        eng.close()


def test_concentration_monitor_tells_ordinary_token_sinks_from_delimiter_sinks(gu, golden_dir):
    """mv_attention_concentration (round 6): the special rows cover attention sinks on [CLS] / [SEP]; a head whose [CLS] row concentrates on an ORDINARY token is
    outside the measured envelope of the default form (profiles/r06_n_sink_envelope.txt: 0.8 - 2.7e-3) — so the attention kernel keeps the largest collision mass
    sum_{j >= 2} p[CLS row][j]^2 it has seen and counts the (sequence, head, layer) items above 0.25, and the Python wrapper warns once.  A [SEP] sink (80 % of
    every row's mass) must NOT trip it, the same sink on a token in the middle of the sequence must, the diffuse model reads ~1 / (effective keys)."""
    import sys
    import warnings
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import r06_make_sink_refs as mk6
    from memvul_amd.binding import Engine

    refs = np.load(os.path.join(golden_dir, "r06_sink_refs.npz"))
    seen = {}
    for case in ("sep_all_80_3001", "mid_all_80_3001"):
        token, rows, pct, seed = case.split("_")
        dims, w, ids, lens, aids, alens, _ = mk6.case(token, rows, int(pct) / 100.0, int(seed), gains=refs[case + "_gains"])
        eng = Engine(0, vocab_size=dims.vocab_size, layers=12, max_tokens=16 * 512, max_batch=16, max_anchors=16)
        try:
            eng.load_state_dict(w)
            with warnings.catch_warnings(record=True) as rec:
                warnings.simplefilter("always")
                eng.encode(ids, lens)
                eng.encode(ids, lens)
            seen[case] = eng.attention_concentration() + (sum("ONE ordinary token" in str(r.message) for r in rec),)
            assert eng.attention_concentration(reset=True)[1] == seen[case][1] and eng.attention_concentration() == (0.0, 0, 0)
        finally:
            eng.close()
    gu.record("concentration_monitor", **{k: list(v) for k, v in seen.items()})
    m_sep, n_sep, t_sep, warned_sep = seen["sep_all_80_3001"]
    m_mid, n_mid, t_mid, warned_mid = seen["mid_all_80_3001"]
    assert t_sep == t_mid == 2 * 8 * 12 * 11                               # two passes of 8 sequences x 12 heads x 11 layers looked at (the pruned last layer runs the single-query tail)
    assert n_sep <= 0.02 * t_sep and warned_sep == 0, seen                 # the sink sits on a special row: covered, silent
    assert n_mid >= 0.5 * t_mid and m_mid > 0.4 and warned_mid == 1, seen  # 80 % on an ordinary token: most (sequence, head, layer) items, warned once
    # the diffuse family of the other tests: far below the threshold
    dk, wk = dict(layers=2), dict(qk_scale=2.0, match_scale=29.0, trained_like=True)
    dims, w = gu.weights_for(dk, wk)
    eng = gu.engine_for(dk, wk, compute_dtype="precise")
    eng.attention_concentration(reset=True)
    ids, lens = synth.make_ids(4, 256, dims.vocab_size, seed=9)
    eng.encode(ids, lens)
    m, n, t = eng.attention_concentration()
    assert n == 0 and t == 4 * 12 and 0.0 < m < 0.1, (m, n, t)


def test_cls_row_aside_is_decided_per_sequence(gu, golden_dir):
    """The other rows' A-side rounding reaches the [CLS] row averaged over the keys (tests/test_precision_model.py::test_cls_row_form_needs_keys_to_average_over), so a sequence takes the form only if it has at least
    MEMVUL_CLS_ASIDE_MIN_LEN (128) tokens — decided per 256-row tile from the sequence's own length (GemmArgs::tile_both), so that a row's result
    still does not depend on the batch it travels in: short sequences give the both-terms form's bits, long ones the same bits alone or among
    short batch-mates; with the rule lifted (MIN_LEN = 1) the short ones change too.  Passes of padded length 256 (issue reports) and 512 (anchors)."""
    import make_golden

    name = "l12_trained_ragged"
    g = np.load(os.path.join(golden_dir, f"{name}.npz"))
    dk, wk, B, S, ragged, G, SA = make_golden.CASES[name]
    kw = dict(compute_dtype="precise", max_tokens=16384, max_batch=64, max_anchors=64)
    for ids, lens in ((g["ids"], g["lens"]), (g["anchor_ids"], g["anchor_lens"])):
        short = lens < 128
        assert short.any() and (~short).any() and ids.shape[1] in (256, 512)
        on = gu.engine_for(dk, wk, env={"MEMVUL_CLS_ASIDE": "1"}, **kw).encode(ids, lens)
        alone = gu.engine_for(dk, wk, env={"MEMVUL_CLS_ASIDE": "1"}, **kw).encode(ids[~short], lens[~short])
        off = gu.engine_for(dk, wk, env={"MEMVUL_CLS_ASIDE": "0"}, **kw).encode(ids, lens)
        assert np.array_equal(on[short], off[short])                                   # short sequences: the both-terms form, bit for bit
        assert all(not np.array_equal(on[i], off[i]) for i in np.flatnonzero(~short))  # long ones: the [CLS]-row form
        assert np.array_equal(alone, on[~short])                                       # ... whatever travels with them
        assert float(np.abs(on - off).max()) < 5e-4
        lifted = gu.engine_for(dk, wk, env={"MEMVUL_CLS_ASIDE": "1", "MEMVUL_CLS_ASIDE_MIN_LEN": "1"}, **kw).encode(ids, lens)
        assert all(not np.array_equal(lifted[i], off[i]) for i in np.flatnonzero(short))
        assert np.array_equal(lifted[~short], on[~short])


def test_cls_row_aside_leaves_other_pass_shapes_alone(gu):
    """The form acts on passes of padded length 256 / 512 (a 256-row tile belongs to one sequence: decided per sequence) and 192 / 384 (decided for the whole
    pass, next test): a 64-token pass (Sp = 64) gives the same bits with and without the switch, 256- and 320-token ones do not."""
    dk, wk = dict(layers=2), dict(qk_scale=2.0, match_scale=29.0, trained_like=True)
    dims, w = gu.weights_for(dk, wk)
    for S, same in ((64, True), (320, False), (256, False)):
        ids, lens = synth.make_ids(4, S, dims.vocab_size, seed=5 + S)
        a = gu.engine_for(dk, wk, compute_dtype="precise", env={"MEMVUL_CLS_ASIDE": "1"}).encode(ids, lens)
        b = gu.engine_for(dk, wk, compute_dtype="precise", env={"MEMVUL_CLS_ASIDE": "0"}).encode(ids, lens)
        assert np.array_equal(a, b) == same, S
        assert float(np.abs(a - b).max()) < 1e-3


@pytest.mark.parametrize("compute", ["precise", "f16"])
@pytest.mark.parametrize("S", [64, 192])
def test_row_bits_do_not_depend_on_the_64_row_group_a_sequence_starts_in(gu, S, compute):
    """Padded lengths 64 / 192: consecutive sequences start in different 64-row groups of a 256-row tile, i.e. in token blocks tb and tb + 4 of a wave —
    two copies of the unrolled epilogue code.  Until round 6 hipcc fused `(half_t)fmaf(..)` into v_fma_mix*_f16 (one rounding) in some of those copies and
    not in others (two roundings): 1 ulp in ~2^-13 of the Q / K / V elements, so a row's bits depended on its position in the pass (gemm_pp.h pin_f32x4).
    Eight copies of one sequence must give one embedding, bit for bit."""
    dk, wk = dict(layers=2, vocab_size=2048), dict(qk_scale=2.0, match_scale=29.0, trained_like=True)
    dims, w = gu.weights_for(dk, wk)
    eng = gu.engine_for(dk, wk, compute_dtype=compute, max_tokens=16384, max_batch=32, max_anchors=8)
    ids1, _ = synth.make_ids(1, S, dims.vocab_size, seed=3 + S)
    u = eng.encode(np.repeat(ids1, 8, axis=0).astype(np.int32), np.full((8,), S, np.int32))
    assert all(np.array_equal(u[i], u[0]) for i in range(1, 8))
    ids, lens = synth.make_ids(7, S, dims.vocab_size, seed=11 + S, ragged=True, min_len=S // 2)
    ids = (ids * (np.arange(S)[None, :] < lens[:, None])).astype(np.int32)
    a = eng.encode(ids, lens)
    perm = np.random.default_rng(S).permutation(7)
    assert np.array_equal(eng.encode(ids[perm], lens[perm]), a[perm])


@pytest.mark.parametrize("S", [192, 320, 384])
def test_cls_row_form_at_padded_lengths_192_and_384_is_decided_for_the_whole_pass(gu, S):
    """Sp = 192 / 384: a 256-row tile spans two sequences, so the form is taken by the WHOLE pass when its shortest sequence has MEMVUL_CLS_ASIDE_MIN_LEN (128)
    tokens (engine.hip encode_dev: what a length-sorted sweep hands over by construction) and by none of it otherwise.  All-long pass: the [CLS]-row form's bits
    (not the both-terms form's), independent of the order of the batch, within the contract of the oracle; one short batch-mate: the both-terms form bit for bit."""
    dk, wk = dict(layers=2, vocab_size=2048), dict(qk_scale=2.0, match_scale=29.0, trained_like=True)
    dims, w = gu.weights_for(dk, wk)
    kw = dict(compute_dtype="precise", max_tokens=16384, max_batch=32, max_anchors=8)
    on = gu.engine_for(dk, wk, env={"MEMVUL_CLS_ASIDE": "1"}, **kw)
    off = gu.engine_for(dk, wk, env={"MEMVUL_CLS_ASIDE": "0"}, **kw)
    B = 7  # an odd batch: the last tile of the pass is partly padding
    ids, lens = synth.make_ids(B, S, dims.vocab_size, seed=11 + S, ragged=True, min_len=max(130, S - 60))
    ids = (ids * (np.arange(S)[None, :] < lens[:, None])).astype(np.int32)
    assert int(lens.min()) >= 128 and (S + 63) // 64 * 64 in (192, 320, 384)  # (320 pads to 384)
    a, b = on.encode(ids, lens), off.encode(ids, lens)
    assert all(not np.array_equal(a[i], b[i]) for i in range(B))
    assert float(np.abs(a - b).max()) < 5e-4
    perm = np.random.default_rng(S).permutation(B)
    assert np.array_equal(on.encode(ids[perm], lens[perm]), a[perm])  # which tile a row lands in does not matter
    u_ref = orc.instance_forward(w, ids.astype(np.int64), synth.mask_from_lens(lens, S))
    ea, eb = float(np.abs(a - u_ref).max()), float(np.abs(b - u_ref).max())
    gu.record("cls_row_whole_pass", S=S, u_err=ea, u_err_both_terms=eb)
    assert ea < 2e-4  # (embeddings; the logit contract at these lengths: test_precise_mode_holds_1e3_in_the_trained_like_regime's ragged goldens)
    lens2 = lens.copy(); lens2[3] = 40
    ids2 = (ids * (np.arange(S)[None, :] < lens2[:, None])).astype(np.int32)
    ids2[3, 39] = ids[3, lens[3] - 1]  # its [SEP]
    assert np.array_equal(on.encode(ids2, lens2), off.encode(ids2, lens2))


def test_short_sequences_carry_v_and_p_as_two_planes(gu, golden_dir):
    """Round 5: what is left of the precise mode's error is the fp16 storage of V and P, which attention averages over the keys — short sequences
    average it least (profiles/r05_f_length_envelope.txt: a 8-token sequence's embedding error alone cost 9.4e-4 on the logits at the trained-like matcher
    norm).  Passes of padded length <= 128 therefore carry V and P as hi + lo fp16 planes through attention (attention_v2.h VLO, GemmArgs::vt_lo;
    MEMVUL_SHORT_VLO=0 is the A/B switch).  16 sequences of 8 / 16 / 32 / 64 tokens on the envelope model against the CPU reference's embeddings
    (tests/golden/r05_trained_like_refs.npz); logit error = what that embedding's error costs against 8 fixed issue-report embeddings."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import r05_make_refs as mk

    refs = np.load(os.path.join(golden_dir, "r05_trained_like_refs.npz"))
    dk, wk = dict(layers=12), dict(seed=mk.ENV_SEED, qk_scale=2.0, match_scale=29.0, trained_like=True)
    dims, w = gu.weights_for(dk, wk)
    u_ref = refs["outlier_1_u"]
    errs = {}
    for tag, env in (("two_planes", {}), ("one_plane", {"MEMVUL_SHORT_VLO": "0"})):
        eng = gu.engine_for(dk, wk, compute_dtype="precise", env=env)
        for L in (8, 16, 32, 64):
            _, ids, lens = mk.length_inputs(L)
            v = eng.encode(ids, lens)
            lg_g = orc.match(u_ref, v, w[synth.KEY_MATCH_W])[0]
            lg_r = orc.match(u_ref, refs[f"len_{L}"], w[synth.KEY_MATCH_W])[0]
            errs[(tag, L)] = float(np.abs(lg_g - lg_r).max())
    gu.record("short_sequences", **{f"{t}_{L}": e for (t, L), e in errs.items()})
    for L in (8, 16, 32, 64):
        assert errs[("two_planes", L)] <= 6e-4, errs          # with margin inside the 1e-3 contract (one plane: up to 9.4e-4)
    assert sum(errs[("two_planes", L)] for L in (8, 16, 32)) < 0.8 * sum(errs[("one_plane", L)] for L in (8, 16, 32)), errs


def test_small_pass_kernels_exclude_the_default_compute_dtype(gu):
    """the product default is MV_F16X8, which only exists on the persistent GEMM path — MEMVUL_GEMM_TILE=128 (the small-pass
    kernels forced) must fail at mv_finalize_weights with a message that names the switch, not compute something else."""
    dk, wk = dict(layers=1), dict()
    with pytest.raises(RuntimeError, match="MEMVUL_GEMM_TILE=128"):
        gu.engine_for(dk, wk, gemm_tile=128, compute_dtype="precise")


def test_unknown_qkv_aside_characters_are_rejected(gu):
    """a typo in MEMVUL_QKV_ASIDE must not silently change the numerics (the same for MEMVUL_CLS_ASIDE and its length rule)."""
    with pytest.raises(RuntimeError, match="MEMVUL_QKV_ASIDE"):
        gu.engine_for(dict(layers=1), dict(), env={"MEMVUL_QKV_ASIDE": "qx"}, compute_dtype="precise")
    with pytest.raises(RuntimeError, match="MEMVUL_CLS_ASIDE"):
        gu.engine_for(dict(layers=1), dict(), env={"MEMVUL_CLS_ASIDE": "on"}, compute_dtype="precise")
    with pytest.raises(RuntimeError, match="MEMVUL_CLS_ASIDE_MIN_LEN"):
        gu.engine_for(dict(layers=1), dict(), env={"MEMVUL_CLS_ASIDE_MIN_LEN": "0"}, compute_dtype="precise")


@pytest.mark.parametrize("compute", ["f16", "precise"])
def test_ref12_logits_against_the_reference_run(gu, compute):
    """tests/golden/ref12: the REFERENCE'S OWN CODE executed on a 12-layer trained-like model (oracle/ref_harness/, a forward
    hook on ModelMemory._projector for the logits, model_memory.py:141): 19 issue reports of up to 256 tokens against 6
    anchors of up to 512, max |logit| 2.5.  MV_F16 is reported against its measured bound; the precise mode must hold the
    contract's 1e-3 on the LOGITS."""
    import test_reference_pin as trp

    ref = trp.get_ref("ref12")
    aids, amask = trp._pad(ref["reader"]["golden"])
    ids, mask = trp._pad(ref["reader"]["test"])
    dk = dict(layers=ref["meta"]["layers"], vocab_size=ref["meta"]["vocab_size"])
    wk = dict(ref["meta"]["weight_kwargs"])
    eng = gu.engine_for(dk, wk, compute_dtype=compute, max_tokens=32 * 512, max_batch=32, max_anchors=16)
    eng.anchor_reset()
    eng.anchor_append(aids.astype(np.int32), amask.sum(1).astype(np.int32))
    v = eng.anchor_get()
    out = eng.forward(ids.astype(np.int32), mask.sum(1).astype(np.int32))
    errs = dict(v=float(np.abs(v - ref["anchors"]).max()), logits=float(np.abs(out["logits"] - ref["logits"]).max()),
                p=float(np.abs(out["probs"] - ref["probs"]).max()), logit_scale=float(np.abs(ref["logits"]).max()))
    gu.record("ref12", compute=compute, **errs)
    assert errs["logit_scale"] > 2.0
    assert errs["logits"] <= (LOGIT_TOL if compute == "precise" else TRAINED_LIKE_LOGIT_BOUND), errs
    # (this fixture's logit pairs are less saturated than the goldens': MV_F16 measures 6.8e-4 on P, the precise mode 6.5e-5)
    assert errs["p"] <= (TRAINED_LIKE_P_TOL if compute == "precise" else LOGIT_TOL), errs
    eng.anchor_reset()


@pytest.mark.parametrize("compute", ["f16", "precise"])
def test_anchor_chunking_and_padding_invariance(gu, compute):
    """Anchors appended in two chunks (128 + rest in the reference, predict_memory.py:81-83) equal one
    append; extra zero padding columns do not change an embedding (masked keys).  Both compute dtypes."""
    dk, wk = dict(layers=2, vocab_size=2048), dict(qk_scale=3.0)
    dims, w = gu.weights_for(dk, wk)
    eng = gu.engine_for(dk, wk, compute_dtype=compute)
    ids, lens = synth.make_ids(9, 96, dims.vocab_size, ragged=True, min_len=5)
    eng.anchor_reset(); eng.anchor_append(ids, lens); v1 = eng.anchor_get()
    eng.anchor_reset(); eng.anchor_append(ids[:4], lens[:4]); eng.anchor_append(ids[4:], lens[4:]); v2 = eng.anchor_get()
    assert np.array_equal(v1, v2)
    wide = np.zeros((9, 128), np.int32); wide[:, :96] = ids
    u_wide = eng.encode(wide, lens)
    assert np.array_equal(u_wide, v1)  # 96 -> Sp 128 either way: bit-identical
    short_rows = lens <= 64
    if short_rows.any():
        u_short = eng.encode(ids[short_rows][:, :64], lens[short_rows])
        assert np.abs(u_short - v1[short_rows]).max() < 5e-4  # different Sp: same maths, different tiling
    ref = orc.build_anchor_bank(w, [ids[i, : lens[i]].astype(np.int64) for i in range(9)])
    e = float(np.abs(v1 - ref).max())
    gu.record("anchor_bank", max_err=e)
    assert e < 2e-3
    eng.anchor_reset()


@pytest.mark.parametrize("compute", ["f16", "precise"])
def test_full_batch_properties(gu, compute):
    """BASELINE.json configs[1] shape (B=256, S=256, G=124) on the 12-layer model: properties that need
    no CPU reference at this size, plus agreement of the resident-corpus path with mv_forward and an
    oracle spot-check of a few rows.  Both compute dtypes (the bench runs both at this shape)."""
    dk, wk = dict(layers=12), dict()
    dims, w = gu.weights_for(dk, wk)
    eng = gu.engine_for(dk, wk, compute_dtype=compute, max_tokens=65536, max_batch=256, max_anchors=128)
    B, S, G = 256, 256, 124
    ids, lens = synth.make_ids(B, S, dims.vocab_size, ragged=False)
    aids, alens = synth.make_ids(G, 64, dims.vocab_size, seed=synth.SEED + 1, ragged=True, min_len=8)
    eng.anchor_reset(); eng.anchor_append(aids, alens)
    out = eng.forward(ids, lens, want_embed=True)
    p = out["probs"]
    assert np.isfinite(out["logits"]).all() and np.isfinite(p).all()
    assert np.abs(p.sum(-1) - 1.0).max() < 1e-6
    assert np.array_equal(out["best_idx"], np.argmax(p[:, :, 0], axis=1).astype(np.int32))
    # batch-composition independence: a permuted batch gives the permuted results bit-for-bit
    perm = np.random.default_rng(1).permutation(B)
    out2 = eng.forward(ids[perm], lens[perm], want_embed=True)
    assert np.array_equal(out2["embed"], out["embed"][perm])
    assert np.array_equal(out2["logits"], out["logits"][perm])
    # resident-corpus path == host path (bit-exact), in two batches of 128
    # (a row's result does not depend on the batch it travels in)
    eng.corpus_upload(ids, lens)
    eng.corpus_run(0, B, 128, keep_probs=True)
    best, idx, ps = eng.corpus_results(0, B, with_probs=True)
    out128 = eng.forward(ids[:128], lens[:128])
    assert np.array_equal(ps[:128], out128["probs"][:, :, 0])
    assert np.array_equal(idx[:128], out128["best_idx"]) and np.array_equal(best[:128], out128["best"])
    assert np.array_equal(ps, out["probs"][:, :, 0]) and np.array_equal(idx, out["best_idx"])
    # oracle spot-check: 2 rows of the full-size batch
    v = eng.anchor_get()
    rows = [0, 255]
    u_ref = orc.instance_forward(w, ids[rows].astype(np.int64), np.ones((2, S), bool))
    lg, pp, bb, ii = orc.match(u_ref, v, w[synth.KEY_MATCH_W])
    e = float(np.abs(out["logits"][rows] - lg).max())
    gu.record("full_batch", logits_err=e, u_err=float(np.abs(out["embed"][rows] - u_ref).max()))
    assert e <= LOGIT_TOL
    eng.anchor_reset()


@pytest.mark.parametrize("compute", ["precise", "f16"])
def test_full_batch_trained_like_rows_against_the_oracle(gu, compute):
    """VERDICT r4 weak #7 / next #5 iii: the BENCH shape (B = 256, S = 256, G = 124) in the TRAINED-LIKE regime (LayerNorm outlier
    dims, peaked attention, matcher x29 -> |logit| ~ 3) — so far that regime only ever ran at B <= 64.  16 rows spread over the
    full-size batch against the numpy oracle's embeddings AND the oracle's own embeddings of 8 of the anchors (logits of both
    operands checked, 256 logit pairs): MV_F16X8 (the default) within the 1e-3 contract, MV_F16 within its measured bound."""
    dk, wk = dict(layers=12), dict(qk_scale=2.0, match_scale=29.0, trained_like=True)
    dims, w = gu.weights_for(dk, wk)
    eng = gu.engine_for(dk, wk, compute_dtype=compute, max_tokens=65536, max_batch=256, max_anchors=128)
    B, S, G = 256, 256, 124
    ids, lens = synth.make_ids(B, S, dims.vocab_size, seed=synth.SEED + 5, ragged=False)
    aids, alens = synth.make_ids(G, 64, dims.vocab_size, seed=synth.SEED + 6, ragged=True, min_len=8)
    eng.anchor_reset(); eng.anchor_append(aids, alens)
    out = eng.forward(ids, lens, want_embed=True)
    assert np.isfinite(out["logits"]).all()
    rows = np.arange(7, B, 16)[:16]     # 7, 23, ..., 247: every 16-row block of the 256-row M tiles, both halves of the batch
    ga = np.arange(0, G, 16)[:8]        # 8 anchors spread over the bank
    u_ref = orc.instance_forward(w, ids[rows].astype(np.int64), np.ones((len(rows), S), bool))
    LA = int(alens[ga].max())
    v_ref = orc.instance_forward(w, aids[ga][:, :LA].astype(np.int64), synth.mask_from_lens(alens[ga], LA))
    lg, pp, bb, ii = orc.match(u_ref, v_ref, w[synth.KEY_MATCH_W])
    got = out["logits"][rows][:, ga]
    e = float(np.abs(got - lg).max())
    gu.record("full_batch_trained_like", compute=compute, logits_err=e, logit_scale=float(np.abs(lg).max()),
              u_err=float(np.abs(out["embed"][rows] - u_ref).max()), v_err=float(np.abs(eng.anchor_get()[ga] - v_ref).max()))
    assert float(np.abs(lg).max()) > 1.5
    # precise: 7.7e-4 with one plane of Q / K / V / P (anchors of 8 - 64 tokens average their fp16 storage over few keys), 5.3e-4 since the
    # short passes carry two (attention_v2.h VLO): the bound guards that gain, the contract is LOGIT_TOL
    assert e <= (7e-4 if compute == "precise" else TRAINED_LIKE_LOGIT_BOUND), e
    if compute == "precise":
        assert eng.x8_saturation() == 0  # nothing in this regime leaves the fp8 planes' range
    eng.anchor_reset()


def test_x8_saturation_counter_counts_and_warns(gu):
    """VERDICT r4 next #5 ii: activations beyond the +-112 range of the MV_F16X8 fp8 planes lose their correction term — counted on the
    device (mv_x8_saturation), surfaced ONCE as a RuntimeWarning by the Python wrapper.  A 1-layer model whose embedding LayerNorm has a
    +300 offset on one dimension puts one element per token out of range in the raw stream; the well-behaved model counts zero."""
    import warnings

    dk = dict(layers=1)
    dims, w = gu.weights_for(dk, dict())
    ids, lens = synth.make_ids(4, 64, dims.vocab_size)
    eng = gu.engine_for(dk, dict(), compute_dtype="precise")
    eng.x8_saturation(reset=True)
    eng.encode(ids, lens)
    assert eng.x8_saturation() == 0
    from memvul_amd.binding import Engine
    w2 = dict(w)
    k = synth.PFX_BERT + "embeddings.word_embeddings.weight"
    w2[k] = w[k].copy()
    w2[k][:, 5] += 300.0  # every token's raw embedding row carries one element near 300 (the stream is pre-LayerNorm)
    e2 = Engine(0, vocab_size=dims.vocab_size, layers=1, max_tokens=16384, max_batch=64, max_anchors=64)
    try:
        e2.load_state_dict(w2, "precise")
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            u = e2.encode(ids, lens)
            e2.encode(ids, lens)
        assert np.isfinite(u).all()
        n = e2.x8_saturation(reset=True)
        assert n >= 2 * 4 * 64                      # at least the embedding's element per token, twice
        assert e2.x8_saturation() == 0              # reset
        assert sum("fp8 correction planes" in str(r.message) for r in rec) == 1  # warned once, not per call
    finally:
        e2.close()


def test_engine_coexists_with_torch_hip_runtime():
    """archive.py (weights.th) and the CPU leg of bench.py import torch in the same process as libmemvul_hip.so.  torch bundles
    its own libamdhip64 (same SONAME): whichever is loaded first serves both.  The supported order is torch FIRST (both
    importers do that before the engine is created): checked in a fresh process.  (Loading torch AFTER the engine makes torch
    run on the system runtime it was not built against: it usually works and once hung for 10 minutes on a GPU box, so that
    order is not supported and not exercised here.  Multi-GPU runs never import torch: distributed.init_transport.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    body = (
        "import numpy as np\n"
        "from memvul_amd import synth\n"
        "from memvul_amd.binding import Engine\n"
        "dims = synth.BertDims(layers=1, vocab_size=1024)\n"
        "e = Engine(0, vocab_size=1024, layers=1, max_tokens=2048, max_batch=8, max_anchors=8)\n"
        "e.load_state_dict(synth.make_weights(dims))\n"
        "ids, lens = synth.make_ids(4, 64, 1024)\n"
        "e.anchor_set(synth.make_anchor_bank(3))\n"
        "o = e.forward(ids, lens)\n"
        "assert np.isfinite(o['logits']).all()\n"
        "print('OK', float(o['logits'][0,0,0]))\n"
    )
    torch_first = "import torch\ntorch.cuda.init()\nx = torch.ones(4, device='cuda') * 2\n" + body + "assert float(x.sum()) == 8.0\n"
    outs = []
    for name, code in (("torch_first", torch_first), ("engine_only", body)):
        r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=420)
        assert r.returncode == 0 and "OK" in r.stdout, f"{name}: {r.stderr[-1500:]}"
        outs.append([ln for ln in r.stdout.splitlines() if ln.startswith("OK")][-1])
    assert outs[0] == outs[1]  # same numbers with and without torch in the process


@pytest.mark.parametrize("gemm_tile,compute", [(0, "f16"), (512, "f16"), (0, "precise")])
@pytest.mark.parametrize("B,S,ragged", [(5, 64, False), (3, 200, True), (2, 320, True)])
def test_last_layer_pruning_matches_full_forward(gu, B, S, ragged, gemm_tile, compute):
    """MEMVUL_CLS_PRUNE: after the last layer's K / V projection only the [CLS] rows are processed (the pooler
    reads hidden[:, 0], model_memory.py:99).  Same embedding as the all-token forward up to the different
    summation order of the single-query attention (fp32 rounding, then fp16 rounding of the context), and the same
    oracle parity."""
    dk, wk = dict(layers=2, vocab_size=2048), dict(qk_scale=3.0)
    dims, w = gu.weights_for(dk, wk)
    ids, lens = synth.make_ids(B, S, dims.vocab_size, ragged=ragged, min_len=7)
    # ("precise": the pruned tail of MV_F16X8 runs in full fp32 on the fp32-input matrix cores, misc_kernels.h dense768_kernel)
    u_full = gu.engine_for(dk, wk, gemm_tile=gemm_tile, compute_dtype=compute, env={"MEMVUL_CLS_PRUNE": "0"}).encode(ids, lens)
    u_cls = gu.engine_for(dk, wk, gemm_tile=gemm_tile, compute_dtype=compute).encode(ids, lens)
    u_ref = orc.instance_forward(w, ids.astype(np.int64), synth.mask_from_lens(lens, S))
    d = float(np.abs(u_full - u_cls).max())
    gu.record("cls_prune", B=B, S=S, gemm_tile=gemm_tile, compute=compute, full_vs_pruned=d, pruned_vs_oracle=float(np.abs(u_cls - u_ref).max()),
              full_vs_oracle=float(np.abs(u_full - u_ref).max()))
    # one fp16 ulp of a context value (different summation order) reaches u at the 5e-5 level; on the persistent
    # path the full forward's last layer also runs with the virtual LayerNorm while the [CLS] tail uses the explicit one
    # (different fp16 roundings of the same mathematics: the 2e-4 level, like either of them against the oracle)
    assert d < 6e-4
    assert np.abs(u_cls - u_ref).max() < 2e-3


@pytest.mark.parametrize("compute", ["precise", "f16"])
@pytest.mark.parametrize("outliers", [False, True])
@pytest.mark.parametrize("prune", ["0", "1"])
@pytest.mark.parametrize("B,S", [(6, 128), (3, 256), (5, 200)])
def test_virtual_layernorm_matches_explicit_layernorm(gu, B, S, prune, outliers, compute):
    """The persistent path has no LayerNorm kernel between the GEMMs — the consumer GEMMs read the raw stream (two fp16
    planes) with gamma / beta / the row mean folded into their weights and scale rows by rstd in the epilogue
    (W LN(r) + b = rstd (W'' r) + b'), the residual GEMMs emit the rows' partial sums — while the small-pass path
    (MEMVUL_GEMM_TILE=128) runs explicit LayerNorm kernels on an fp32 stream.  Same mathematics, different roundings:
    agreement of the two engines and of each with the oracle at the fp16-operand level, also with
    trained-checkpoint-like outlier dimensions in every LayerNorm."""
    dk, wk = dict(layers=4, vocab_size=2048), dict(qk_scale=2.0, ln_outliers=outliers)
    dims, w = gu.weights_for(dk, wk)
    ids, lens = synth.make_ids(B, S, dims.vocab_size, ragged=True, min_len=9)
    # the virtual side in both compute dtypes (the product default runs the persistent kernels at every size; MV_F16 has them forced here); the explicit
    # side exists in MV_F16 only (the small-pass kernels)
    u_v = gu.engine_for(dk, wk, gemm_tile=0 if compute == "precise" else 512, compute_dtype=compute, env={"MEMVUL_CLS_PRUNE": prune}).encode(ids, lens)
    u_e = gu.engine_for(dk, wk, gemm_tile=128, compute_dtype="f16", env={"MEMVUL_CLS_PRUNE": prune}).encode(ids, lens)
    u_ref = orc.instance_forward(w, ids.astype(np.int64), synth.mask_from_lens(lens, S))
    ev, ee = float(np.abs(u_v - u_ref).max()), float(np.abs(u_e - u_ref).max())
    gu.record("virtual_ln", B=B, S=S, prune=prune, outliers=outliers, compute=compute, virtual_vs_oracle=ev, explicit_vs_oracle=ee,
              virtual_vs_explicit=float(np.abs(u_v - u_e).max()), u_scale=float(np.abs(u_ref).max()))
    assert ev < 2e-3 and ev < 3 * ee + 2e-4


@pytest.mark.parametrize("compute", ["precise", "f16"])
def test_length_bucketed_sweep_matches_padded_sweep(gu, compute):
    """Engine.bucketed_sweep: rows sorted by length, every batch processed at its own longest member's length
    (mv_corpus_run_len) instead of the corpus-wide S, results returned in the original order.  Same per-row mathematics
    at a different padded length (different tiling / kernel instantiation): probabilities agree at the fp16-operand
    level and the decisions agree wherever the top-2 margin is clear."""
    dk, wk = dict(layers=3, vocab_size=2048), dict(qk_scale=2.0, match_scale=6.0)
    dims, w = gu.weights_for(dk, wk)
    eng = gu.engine_for(dk, wk, compute_dtype=compute, max_tokens=64 * 256, max_batch=64, max_anchors=32)
    ids, lens = synth.make_ids(150, 256, dims.vocab_size, ragged=True, min_len=5)
    eng.anchor_set(synth.make_anchor_bank(24))
    eng.corpus_upload(ids, lens)
    eng.corpus_run(0, 150, 64, keep_probs=True)
    best0, idx0, ps0 = eng.corpus_results(0, 150, with_probs=True)
    best1, idx1, ps1 = eng.bucketed_sweep(ids, lens, 64, with_probs=True)
    d = float(np.abs(ps0 - ps1).max())
    gu.record("bucketed_sweep", compute=compute, max_p_diff=d)
    assert d < 1e-3
    srt = np.sort(ps0, axis=1)
    clear = (srt[:, -1] - srt[:, -2]) > 4e-3
    assert np.array_equal(idx0[clear], idx1[clear])
    # and against the oracle on the shortest / longest rows
    v = eng.anchor_get()
    rows = [int(np.argmin(lens)), int(np.argmax(lens))]
    for r in rows:
        L = int(lens[r])
        u_ref = orc.instance_forward(w, ids[r:r + 1, :L].astype(np.int64), np.ones((1, L), bool))
        lg, pp, bb, ii = orc.match(u_ref, v, w[synth.KEY_MATCH_W])
        assert np.abs(ps1[r] - pp[0, :, 0]).max() < 2e-3
    eng.anchor_reset()


@pytest.mark.parametrize("compute", ["precise", "f16"])
def test_forward_by_length_matches_the_padded_forward(gu, compute):
    """Engine.forward_by_length (what ModelMemory.forward calls): the rows of a pad-to-longest batch grouped by their own padded length, one mv_forward per
    group, results back in place.  Same per-row mathematics as the one padded pass at a different padded length: probabilities agree at the fp16-operand
    level, decisions wherever the top-2 margin is clear; rows of the longest group ran at the batch's own length in both forms and (MV_F16X8) agree bit for bit."""
    dk, wk = dict(layers=3, vocab_size=2048), dict(qk_scale=2.0, match_scale=6.0)
    dims, w = gu.weights_for(dk, wk)
    eng = gu.engine_for(dk, wk, compute_dtype=compute, max_tokens=128 * 512, max_batch=128, max_anchors=32)
    ids, lens = synth.make_ids(128, 512, dims.vocab_size, ragged=True, min_len=5)
    ids = (ids * (np.arange(512)[None, :] < lens[:, None])).astype(np.int32)
    eng.anchor_set(synth.make_anchor_bank(24))
    a = eng.forward(ids, lens)
    b = eng.forward_by_length(ids, lens, min_tokens=4096)
    # b came from ONE mv_forward_ragged call; the same grouping done in Python around mv_forward_groups, and around one mv_forward per group: the same bits
    eng._forward_ragged = None
    try:
        c = eng.forward_by_length(ids, lens, min_tokens=4096)
        eng._forward_groups = None
        d2 = eng.forward_by_length(ids, lens, min_tokens=4096)
    finally:
        del eng._forward_ragged
        if "_forward_groups" in eng.__dict__:
            del eng._forward_groups
    assert all(np.array_equal(b[k], c[k]) and np.array_equal(b[k], d2[k]) for k in ("logits", "probs", "best", "best_idx"))
    e1 = eng.forward_by_length(ids, lens, want_logits=False, want_embed=True, min_tokens=4096)
    assert e1["logits"] is None and np.array_equal(e1["probs"], b["probs"]) and np.array_equal(e1["embed"], eng.forward_by_length(ids, lens, want_embed=True, min_tokens=4096)["embed"])
    d = float(np.abs(a["probs"] - b["probs"]).max())
    gu.record("forward_by_length", compute=compute, max_p_diff=d)
    assert d < 1e-3
    srt = np.sort(a["probs"][:, :, 0], axis=1)
    clear = (srt[:, -1] - srt[:, -2]) > 4e-3
    assert np.array_equal(a["best_idx"][clear], b["best_idx"][clear])
    longest = lens > 384
    assert longest.sum() >= 8
    if compute == "precise":  # (MV_F16 picks its kernel path by the size of the pass: the smaller group runs the small-pass kernels)
        assert np.array_equal(a["probs"][longest], b["probs"][longest]) and np.array_equal(a["logits"][longest], b["logits"][longest])
    short = lens <= 64
    assert short.any() and not np.array_equal(a["probs"][short], b["probs"][short])  # (they did run at another padded length)
    # the two halves (mv_forward_ragged_begin / _end): two batches in flight on the two workspace sets, collected in order, the bits of the one-call form
    t1 = eng.forward_by_length_begin(ids, lens, min_tokens=4096)
    t2 = eng.forward_by_length_begin(ids[::-1].copy(), lens[::-1].copy(), want_logits=False, min_tokens=4096)
    assert t1[0] == "pending" and t2[0] == "pending"
    t3 = eng.forward_by_length_begin(ids[:16], lens[:16])  # too small to be worth grouping: scored at once, next to the two in flight
    assert t3[0] == "done"
    with pytest.raises(RuntimeError):
        eng.forward_by_length_end(t2)  # out of order
    r1, r2, r3 = eng.forward_by_length_end(t1), eng.forward_by_length_end(t2), eng.forward_by_length_end(t3)
    assert all(np.array_equal(r1[k], b[k]) for k in ("logits", "probs", "best", "best_idx"))
    assert r2["logits"] is None and np.array_equal(r2["probs"], b["probs"][::-1]) and np.array_equal(r2["best_idx"], b["best_idx"][::-1])
    assert np.array_equal(r3["probs"], eng.forward(ids[:16], lens[:16])["probs"])
    # mv_forward_groups rejects what it cannot run as handed over (before any GPU work)
    bufs = {"logits": None, "probs": None, "best": np.empty((128, 2), np.float32), "best_idx": np.empty(128, np.int32), "embed": None}
    for ends, widths in (([64], [512]), ([128], [64]), ([64, 64], [512, 512]), ([64, 128], [512, 600])):  # not the whole batch / a row longer than its group / empty group / wider than S
        with pytest.raises(RuntimeError):
            eng._forward_groups(ids, lens, ends, widths, bufs)
    eng.anchor_reset()


@pytest.mark.parametrize("gemm_tile,compute", PATHS)
def test_edge_shapes_against_the_oracle(gu, gemm_tile, compute):
    """The smallest and the largest inputs the path accepts: a one-token issue report, one anchor, a batch whose rows
    are 1 / 33 / 64 tokens long, and a 512-token row (max_pos) next to a 3-token one; errors for what it must reject."""
    dk, wk = dict(layers=2, vocab_size=2048), dict(qk_scale=3.0)
    dims, w = gu.weights_for(dk, wk)
    eng = gu.engine_for(dk, wk, gemm_tile=gemm_tile, compute_dtype=compute, max_tokens=8 * 512, max_batch=8, max_anchors=4)

    def check(ids, lens, G):
        S = ids.shape[1]
        aids, alens = synth.make_ids(G, 40, dims.vocab_size, seed=synth.SEED + 5, ragged=True, min_len=1)
        eng.anchor_reset(); eng.anchor_append(aids, alens)
        out = eng.forward(ids, lens, want_embed=True)
        v = orc.build_anchor_bank(w, [aids[i, : alens[i]].astype(np.int64) for i in range(G)])
        u, logits, p, best, idx = orc.predict(w, ids.astype(np.int64), synth.mask_from_lens(lens, S), v)
        assert np.abs(out["embed"] - u).max() < 2e-3
        assert np.abs(out["logits"] - logits).max() <= LOGIT_TOL
        assert out["logits"].shape == (ids.shape[0], G, 2)

    one = np.array([[101]], np.int32)
    check(one, np.array([1], np.int32), 1)                                   # B = 1, S = 1, G = 1
    ids, _ = synth.make_ids(3, 64, dims.vocab_size, ragged=False)
    lens = np.array([1, 33, 64], np.int32)
    ids = ids * (np.arange(64)[None, :] < lens[:, None])
    check(ids.astype(np.int32), lens, 3)
    ids, _ = synth.make_ids(2, 512, dims.vocab_size, ragged=False)
    lens = np.array([512, 3], np.int32)
    ids = ids * (np.arange(512)[None, :] < lens[:, None])
    check(ids.astype(np.int32), lens, 2)                                      # max_pos next to a 3-token row
    with pytest.raises(RuntimeError):
        eng.forward(np.zeros((1, 513), np.int32), np.array([513], np.int32))  # longer than max_pos
    # more rows than max_batch / max_tokens hold: mv_forward walks them in chunks, results as for the rows alone
    ids9, lens9 = synth.make_ids(9, 512, dims.vocab_size, ragged=True, min_len=300)
    eng.anchor_set(synth.make_anchor_bank(3))
    o9 = eng.forward(ids9, lens9)
    o1 = eng.forward(ids9[8:], lens9[8:])
    assert np.array_equal(o9["logits"][8:], o1["logits"])
    eng.anchor_reset()
    with pytest.raises(RuntimeError):
        eng.forward(one, np.array([1], np.int32))                              # empty anchor bank


@pytest.mark.parametrize("compute", ["f16", "precise"])
def test_use_header_false_matches_the_oracle(gu, compute):
    """`use_header: false` (model_memory.py:69-73): mv_config.proj_dim = 768 — no header launch, the pooler output is the
    embedding, the fused matcher runs its 768-wide instantiation (anchors [G, 768], W_m [2, 2304]); checked against the oracle
    on the same header-less weights, plus the matcher's own consistency (best = row of probs at the first arg-max, top-k)."""
    # (the 768-d tanh embedding has a larger norm than the 512-d header output: at match_scale 4 the logits reach a few units,
    #  where MV_F16 measures 1.1e-3 — the trained-like effect of DESIGN.md §2 — so that scale is kept for the precise mode)
    dk, wk = dict(layers=2, vocab_size=2048), dict(qk_scale=3.0, match_scale=4.0 if compute == "precise" else 1.0, use_header=False)
    dims, w = gu.weights_for(dk, wk)
    assert synth.KEY_HEAD_W not in w and w[synth.KEY_MATCH_W].shape == (2, 3 * 768)
    eng = gu.engine_for(dk, wk, compute_dtype=compute, proj_dim=768, max_tokens=16384, max_batch=64, max_anchors=300)
    aids, alens = synth.make_ids(7, 96, dims.vocab_size, seed=synth.SEED + 1, ragged=True, min_len=8)
    ids, lens = synth.make_ids(9, 128, dims.vocab_size, ragged=True, min_len=5)
    eng.anchor_reset(); eng.anchor_append(aids, alens)
    v = eng.anchor_get()
    out = eng.forward(ids, lens, want_embed=True)
    assert v.shape == (7, 768) and out["embed"].shape == (9, 768)
    v_ref = orc.build_anchor_bank(w, [aids[i, : alens[i]].astype(np.int64) for i in range(7)])
    u, logits, p, best, idx = orc.predict(w, ids.astype(np.int64), synth.mask_from_lens(lens, 128), v_ref)
    errs = dict(v=float(np.abs(v - v_ref).max()), u=float(np.abs(out["embed"] - u).max()), logits=float(np.abs(out["logits"] - logits).max()))
    gu.record("use_header_false", compute=compute, logit_scale=float(np.abs(logits).max()), **errs)
    assert errs["logits"] <= LOGIT_TOL and errs["u"] < 2e-3, errs
    # the matcher alone at this width, also through the 256-anchor chunks + merge (G = 300)
    rng = np.random.default_rng(11)
    uu = np.tanh(rng.standard_normal((37, 768))).astype(np.float32)
    vv = np.tanh(rng.standard_normal((300, 768))).astype(np.float32)
    eng.anchor_set(vv)
    o = eng.match(uu)
    lg, pp, bb, ii = orc.match(uu, vv, w[synth.KEY_MATCH_W], same_idx=0)
    assert np.abs(o["logits"] - lg).max() < 5e-5 and np.abs(o["probs"] - pp).max() < 1e-5
    ps = o["probs"][:, :, 0]
    assert np.array_equal(o["best_idx"], np.argmax(ps, axis=1).astype(np.int32))
    tp, ti = eng.topk(uu, 5)
    rp, ri = orc.topk_match(ps, 5)
    assert np.array_equal(ti, ri.astype(np.int32)) and np.array_equal(tp, rp)
    with pytest.raises(ValueError):
        eng.anchor_set(np.zeros((3, 512), np.float32))  # a 512-wide bank on a 768-wide engine
    eng.anchor_reset()


@pytest.mark.parametrize("compute", ["precise", "f16"])
def test_sweeps_chunk_batches_larger_than_one_pass(gu, compute):
    """batch_size = 512 (the reference __main__ value, predict_memory.py:207) with issue reports longer than
    max_tokens / 512 used to fail with MV_ERR_CAPACITY on the resident sweeps.  mv_corpus_run_len now walks such a batch in
    passes of what fits (as mv_forward / mv_encode do) with bit-identical per-row results."""
    dk, wk = dict(layers=2, vocab_size=2048), dict(qk_scale=2.0, match_scale=6.0)
    dims, w = gu.weights_for(dk, wk)
    eng = gu.engine_for(dk, wk, compute_dtype=compute, max_tokens=24 * 256, max_batch=512, max_anchors=32)  # 24 rows of 256 tokens per pass
    ids, lens = synth.make_ids(130, 256, dims.vocab_size, ragged=True, min_len=120)
    eng.anchor_set(synth.make_anchor_bank(9))
    best, idx, ps = eng.bucketed_sweep(ids, lens, 512, with_probs=True)       # one "batch" of 130 rows > 24 per pass
    eng.corpus_upload(ids, lens)
    eng.corpus_run(0, 130, 512, keep_probs=True)
    best2, idx2, ps2 = eng.corpus_results(0, 130, with_probs=True)
    ref = eng.forward(ids, lens)                                              # mv_forward chunks on its own
    assert np.array_equal(ps2, ref["probs"][:, :, 0]) and np.array_equal(idx2, ref["best_idx"]) and np.array_equal(best2, ref["best"])
    assert np.abs(ps - ps2).max() < 1e-3 and best.shape == (130, 2)           # bucketed: rows run at other padded lengths
    eng.anchor_reset()


def test_rejects_token_ids_outside_the_vocabulary(gu):
    """an id >= vocab_size (tokenizer / checkpoint vocabulary mismatch) used to be clamped silently by the
    embedding kernel; HF raises.  Every entry point that takes ids now rejects it."""
    dk, wk = dict(layers=2, vocab_size=2048), dict(qk_scale=3.0)
    eng = gu.engine_for(dk, wk)
    ids, lens = synth.make_ids(3, 64, 2048)
    bad = ids.copy(); bad[1, 7] = 2048
    neg = ids.copy(); neg[2, 0] = -1
    eng.anchor_set(synth.make_anchor_bank(3))
    for arr in (bad, neg):
        for call in (lambda a: eng.encode(a, lens), lambda a: eng.forward(a, lens), lambda a: eng.anchor_append(a, lens),
                     lambda a: eng.corpus_upload(a, lens)):
            with pytest.raises(RuntimeError, match="vocab"):
                call(arr)
    assert np.isfinite(eng.encode(ids, lens)).all()
    eng.anchor_reset()


def test_rccl_bound_in_the_library_one_rank(gu):
    """mv_comm_prepare / mv_comm_unique_id / mv_comm_init / mv_comm_allgather: librccl.so opened at run time, the unique id drawn
    by rank 0 and handed over as BYTES (no id file), a real one-rank communicator on the engine's stream; then the module-level
    transport (distributed.init_transport) that bench.py and test_siamese_sharded use — with no torch.distributed anywhere."""
    from memvul_amd import distributed as mvdist

    dk, wk = dict(layers=2, vocab_size=2048), dict(qk_scale=3.0)
    eng = gu.engine_for(dk, wk)
    eng.comm_prepare()
    uid = eng.comm_unique_id()
    assert len(uid) == 128
    eng.comm_init(0, 1, uid)
    x = np.arange(1000, dtype=np.float32).reshape(250, 4)
    out = eng.comm_allgather(x)
    assert out.shape == (1, 250, 4) and np.array_equal(out[0], x)
    ids, lens = synth.make_ids(4, 64, 2048)
    u0 = eng.encode(ids, lens)                    # engine work and the communicator share the stream
    assert np.array_equal(eng.comm_allgather(u0)[0], u0)
    eng.comm_destroy()
    with pytest.raises(RuntimeError):
        eng.comm_init(0, 2, b"short")             # a malformed id is rejected before RCCL sees it
    assert mvdist.init_transport(eng, 0, 1).startswith("none")
    s, l = mvdist.all_gather_stats(np.array([0.25, 0.75], np.float32), np.array([0, 1], np.uint8))
    assert s.tolist() == [0.25, 0.75] and l.tolist() == [0, 1] and mvdist.all_reduce_max(3.5) == 3.5
    mvdist.barrier()
    mvdist.shutdown()


def test_bench_scale_precise_sweep_is_deterministic_and_independent_of_streams():
    """The bench-scale form of the default compute dtype (256 x 256 tokens per batch: full persistent grids; the QKV projection's fp8 sweep
    walks a per-tile K-tile count since round 4): four batches through mv_corpus_run three times with two batches in flight and once with
    one — every per-IR result (best pair, best index, P(same) of all anchors) bit-identical.  scripts/r04_determinism.py is the 12-layer form."""
    from memvul_amd.binding import Engine

    dims = synth.BertDims(layers=2)
    w = synth.make_weights(dims, qk_scale=2.0, match_scale=29.0, trained_like=True)
    B, S, G, NB = 256, 256, 124, 4
    eng = Engine(0, vocab_size=dims.vocab_size, layers=2, max_tokens=B * S, max_batch=B, max_anchors=128)
    try:
        eng.load_state_dict(w)  # the default: MV_F16X8
        aids, alens = synth.make_ids(G, 256, dims.vocab_size, seed=synth.SEED + 1, ragged=True, min_len=32)
        eng.anchor_append(aids[:, :int(alens.max())], alens)
        ids, lens = synth.make_ids(NB * B, S, dims.vocab_size, seed=5)
        eng.corpus_upload(ids, lens)
        ref = None
        for streams in (2, 2, 2, 1):
            eng.set_streams(streams)
            for i in range(NB):
                eng.corpus_run(i * B, B, B, keep_probs=True)
            got = eng.corpus_results(0, NB * B, with_probs=True)
            assert np.isfinite(got[2]).all()
            if ref is None:
                ref = tuple(x.copy() for x in got)
            else:
                assert all(np.array_equal(a, b) for a, b in zip(ref, got)), streams
    finally:
        eng.close()
