/*
 * memvul_hip.h — C ABI of libmemvul_hip.so: the MI355X (gfx950) inference engine for MemVul's
 * predict_memory.py hot loop (BERT issue-encoder forward + CWE golden-anchor memory matching).
 *
 * The reference is pure Python and has no FFI of its own; this header is the boundary a
 * maintainer binds with ctypes from `MemVul/model_memory.py`.  Each entry
 * point names the reference code it replaces (paths relative to the MemVul repository).
 *
 * Conventions
 *   - every function returns MV_OK (0) or a negative mv_status; the message is available from
 *     mv_last_error(); no C++ exception crosses the ABI; HIP errors are captured and translated.
 *   - the caller owns every host buffer; the library owns all device memory (weights, anchor bank,
 *     workspaces, resident corpus).  No device pointer is ever returned.
 *   - one handle <-> one GPU <-> one HIP stream; a handle is not thread-safe; distinct handles are
 *     independent (one process per GPU in multi-GPU runs).
 *   - entry points that launch device work are asynchronous with respect to the host until
 *     mv_sync(), except where they copy results back to host memory (they synchronise first).
 *   - token ids are int32, sequences are 0-padded ([PAD]=0) to S columns, `lens[b]` is the number of
 *     real tokens of row b (the reference's boolean `mask` is `arange(S) < lens[b]`,
 *     custom_PTM_embedder.py:215-228).  S may be any value in [1, max_pos]; the engine pads
 *     internally to a multiple of 64 with masked keys.
 */
#ifndef MEMVUL_HIP_H
#define MEMVUL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mv_handle mv_handle;

typedef enum mv_status {
  MV_OK = 0,
  MV_ERR_INVALID = -1,        /* bad argument / shape */
  MV_ERR_HIP = -2,            /* a HIP runtime call or kernel launch failed */
  MV_ERR_STATE = -3,          /* call order violated (e.g. forward before finalize / no anchors) */
  MV_ERR_MISSING_WEIGHT = -4, /* mv_finalize_weights: a state-dict key was never loaded */
  MV_ERR_CAPACITY = -5,       /* B*S, B or G exceeds what mv_create reserved */
  MV_ERR_NOMEM = -6,          /* host or device allocation failed */
  MV_ERR_INTERNAL = -7        /* a C++ exception was caught at the ABI boundary (never propagated to the caller) */
} mv_status;

typedef enum mv_dtype { MV_F32 = 0, MV_F16 = 1, MV_BF16 = 2, MV_I32 = 3, MV_I64 = 4,
                        /* compute dtype only ("precise"): the fp16 MFMA sweep of every encoder GEMM plus ONE correction sweep on the
                         * fp8 matrix path (OCP e4m3, v_mfma_scale_f32_16x16x128_f8f6f4) over the first-order terms of the split-operand
                         * product, A_lo8 W_hi8 + A_hi8 W_lo8 — ~15.5-bit operands at 2x the GEMM main loop: the mode that holds 1e-3 on
                         * the logits in the trained-like regime.  (5 was MV_F16X2, the three-sweep fp16 split
                         * of round 2 that this mode replaces; it is rejected now.) */
                        MV_F16X8 = 6 } mv_dtype;

/* Geometry + capacities.  The kernels are specialised to bert-base geometry (hidden 768, 12 heads
 * of 64, intermediate 3072, header 512); `layers`, `vocab_size`, `max_pos` are free.
 * (HF BertConfig defaults; model hyper-parameters MemVul/config_memory.json:31-49.) */
typedef struct mv_config {
  int32_t vocab_size;   /* 30522 */
  int32_t hidden;       /* 768  (must be 768) */
  int32_t layers;       /* 12 */
  int32_t heads;        /* 12   (must be 12) */
  int32_t intermediate; /* 3072 (must be 3072) */
  int32_t max_pos;      /* 512 */
  int32_t type_vocab;   /* 2 */
  int32_t proj_dim;     /* 512: the header output (FeedForward(768,1,[512],ReLU), model_memory.py:70; use_header = true, every reference
                         * config) — or 768: use_header = false (l.69-73): no `_projector_single`, the embedding is the pooler output
                         * and `_projector.weight` is [2, 3 * 768] */
  float ln_eps;         /* 1e-12 */
  int32_t max_tokens;   /* capacity of one forward in padded tokens, B * Sp (Sp = S rounded up to 64, above 256 to 128) */
  int32_t max_batch;    /* capacity of one forward in issue reports */
  int32_t max_anchors;  /* capacity of the anchor bank (G) */
  int32_t same_idx;     /* index of label "same" in the `labels` vocabulary (model_memory.py:61) */
} mv_config;

/* ---- lifetime ----------------------------------------------------------------------------- */

/* Replaces Model.from_params + model.to(cuda_device) (predict_memory.py:62-70): binds `device`,
 * creates the stream and reserves all workspaces. */
/* Environment switches read HERE — five, each with a tested default, each parsed strictly (a value the library does not understand fails mv_create with a
 * message; a typo never selects other numerics silently):
 *   MEMVUL_CLS_ASIDE          1 (default) | 0.  MV_F16X8: 1 = the [CLS]-row form (every GEMM sweeps the weight-side correction term, the A-side term is
 *                             restored for the [CLS] row of each sequence alone: only that row reaches the pooler, model_memory.py:99); 0 = both first-order
 *                             terms in every row (rounds 3-4: -13 % issue reports/s, same trained-like logit error on diffuse attention).
 *   MEMVUL_CLS_ASIDE_MIN_LEN  1 .. 512 (default 128): sequences shorter than this keep the both-terms form (few keys to average over) — decided per sequence in
 *                             passes of padded length 256 / 512, for the whole pass (by its shortest sequence) at 192 / 384; shorter passes always keep it.
 *   MEMVUL_QKV_ASIDE          a subset of "qkv", "" or "none" (default "none"): the blocks of the QKV projection that sweep the A-side term for EVERY row (the special
 *                             rows get it in every block either way; "q" = the default of rounds 4 - 6a: -2.7 % issue reports/s, 3 % less logit error).
 *   MEMVUL_CLS_PRUNE          1 (default) | 0: after the last layer's K / V projection only the [CLS] rows are processed.
 *   MEMVUL_STREAMS            2 (default) | 1: batches of the resident sweep in flight (mv_set_streams changes it later).
 * (The sixth switch of the product, MEMVUL_COMPUTE = precise | f16, is read by the Python surface: memvul_amd/binding.py default_compute.)
 * Development A/B knobs (kernel path forced at test sizes, raster, grid share, one-plane short passes) exist only in the -DMEMVUL_DEV_SWITCHES build
 * (libmemvul_hip_dev.so: memvul_amd/build.py, loaded by the GPU tests and A/B scripts that need them); this library does not read them. */
int mv_create(int device, const mv_config* cfg, mv_handle** out);
void mv_destroy(mv_handle* h);
/* Last error message of this handle (or of a failed mv_create when h == NULL). */
const char* mv_last_error(mv_handle* h);
int mv_sync(mv_handle* h);

/* ---- weights (replaces model.load_state_dict(weights.th), AllenNLP archival) ---------------- */

/* `name` is a key of the reference model's state_dict:
 *   _text_field_embedder.token_embedder_tokens.transformer_model.<HF BertModel key>
 *   _bert_pooler.pooler.dense.{weight,bias}            (model_memory.py:64)
 *   _projector_single._linear_layers.0.{weight,bias}   (model_memory.py:70)
 *   _projector.weight                                  (model_memory.py:73)
 * Unknown keys (e.g. ...embeddings.position_ids, custom_PTM_embedder.py:64) are accepted and
 * ignored.  dtype MV_F32 / MV_F16 / MV_BF16; the data is copied, the caller may free it. */
int mv_load_tensor(mv_handle* h, const char* name, const void* host_ptr, int dtype, const int64_t* shape, int ndim);
/* Checks that every needed key is present and well-shaped, packs QKV, converts the GEMM weights to
 * `compute_dtype` and uploads.  Compute dtypes: MV_F16X8 (fp16 MFMA sweep + an fp8 correction sweep per GEMM, see mv_dtype:
 * what the Python surface passes by default and what bench.py's headline is measured in — it holds the 1e-3 logit tolerance of
 * model_memory.py:133-147 on trained-like weights) and MV_F16 (one fp16 sweep, fp32 accumulation: the explicit "fast" opt-in,
 * 3.0-5.6e-3 on such weights); anything else returns MV_ERR_INVALID.  MV_BF16 is a STORAGE dtype of mv_load_tensor only (bf16 checkpoints load): as MFMA
 * operand format it was measured and rejected — 8 significand bits put the match logits 1.5e-2 off at |logit| ~ 3
 * and 2.5e-3 off even on random-init weights (oracle/precision_model.py), against a 1e-3 budget, at the
 * same MFMA rate as fp16.  Embeddings, LayerNorm, biases, pooler, header and matcher stay fp32. */
int mv_finalize_weights(mv_handle* h, int compute_dtype);

/* ---- anchor memory (replaces ModelMemory.forward_gold_instances, model_memory.py:105-115, as
 *      driven by predict_memory.py:81-83 and callbacks.py:48-53) ------------------------------ */

int mv_anchor_reset(mv_handle* h); /* _golden_instances_embeddings = None */
/* Encodes n anchors (ids [n,S], lens [n]) and appends their 512-d embeddings to the bank. */
int mv_anchor_append(mv_handle* h, const int32_t* ids, const int32_t* lens, int n, int S);
int mv_anchor_count(mv_handle* h);
/* Copies the bank to host: out fp32 [G,512]. */
int mv_anchor_get(mv_handle* h, float* out);
/* Installs a precomputed bank v fp32 [G,512] (BASELINE.json configs[4]: synthetic 1000-anchor bank). */
int mv_anchor_set(mv_handle* h, const float* v, int G);

/* ---- the hot loop (replaces ModelMemory.forward test/unlabel branch, model_memory.py:133-147,
 *      including _instance_forward l.90-103 and the embedder forward custom_PTM_embedder.py:172-242)
 * ids int32 [B,S] host, lens int32 [B] host.  Any output pointer may be NULL.
 *   logits fp32 [B,G,2]   W_m [u; v; |u-v|]                       (l.141)
 *   probs  fp32 [B,G,2]   softmax(logits, -1)                     (l.142; `output_dict['probs']`)
 *   best   fp32 [B,2]     probs[b, argmax_g probs[b,g,same_idx]]  (l.144-147)
 *   best_idx int32 [B]    that argmax (first maximal g)
 *   embed  fp32 [B,512]   u = header(pooler(BERT(ids)[:,0]))      (l.133)
 * (The matcher accumulates delta = logit_0 - logit_1 as one fp32 chain with the class-difference weights and derives probs from
 *  it — softmax_2 depends on nothing else — on every entry point; when `logits` is requested the class-0 chain runs too and
 *  logit_1 = logit_0 - delta.  Both agree with the reference's two separate sums to fp32 rounding: ~1e-6 on the logits.)
 * Returns after the results are in host memory. */
int mv_forward(mv_handle* h, const int32_t* ids, const int32_t* lens, int B, int S,
               float* logits, float* probs, float* best, int32_t* best_idx, float* embed);
/* mv_forward on a batch whose rows the caller has ordered by length and cut into groups (binding.Engine.forward_by_length: the reference's pad-to-longest
 * batch of UNSORTED issue reports, predict_memory.py:97-101, scored without its padding): group g = rows [group_end[g - 1], group_end[g]) is one pass at
 * group_width[g] <= S tokens per row (every row of it at most that long), the groups run back to back, one synchronisation.  ids [B][S], outputs as
 * mv_forward's, in the order of the rows handed over.  B <= max_batch, B * S <= max_tokens. */
int mv_forward_groups(mv_handle* h, const int32_t* ids, const int32_t* lens, int B, int S, int n_groups, const int32_t* group_end,
                      const int32_t* group_width, float* logits, float* probs, float* best, int32_t* best_idx, float* embed);
/* All of that in one call for a batch in ANY row order: rows ordered by the padded length of their own token count (64 .. 256 in steps of 64, 384, 512),
 * groups of fewer than min_tokens padded tokens merged into the next longer one, results in the caller's row order (binding.Engine.forward_by_length). */
int mv_forward_ragged(mv_handle* h, const int32_t* ids, const int32_t* lens, int B, int S, int min_tokens, float* logits, float* probs, float* best,
                      int32_t* best_idx, float* embed);
/* mv_forward_ragged in two halves: `begin` enqueues the batch (upload, passes, download into pinned staging) on the stream of the next workspace set and returns a
 * ticket without waiting; `end` waits for it and fills the caller's arrays (those of the outputs `begin` was asked for; best / best_idx always).  One batch per
 * workspace set (MEMVUL_STREAMS, 2 by default) may be in flight; collect tickets in the order they were issued.  predict_memory.evaluate hands over batch k + 1
 * before it collects batch k, so the GPU does not wait for the host between batches (the reference's loop is serial: predict_memory.py:103-110).  The batches
 * use the workspace sets of the resident sweep (mv_corpus_run): collect every ticket before starting one, and the other way round; like the rest of a handle's
 * entry points these two are not thread-safe. */
int mv_forward_ragged_begin(mv_handle* h, const int32_t* ids, const int32_t* lens, int B, int S, int min_tokens, int want_logits, int want_probs, int want_embed,
                            int* ticket);
int mv_forward_ragged_end(mv_handle* h, int ticket, float* logits, float* probs, float* best, int32_t* best_idx, float* embed);
/* Encoder only (ModelMemory._instance_forward, model_memory.py:90-103): embed fp32 [B,512]. */
int mv_encode(mv_handle* h, const int32_t* ids, const int32_t* lens, int B, int S, float* embed);
/* Matcher only on host embeddings u fp32 [B,512] against the resident bank (model_memory.py:135-147). */
int mv_match(mv_handle* h, const float* u, int B, float* logits, float* probs, float* best, int32_t* best_idx);
/* Fused match + top-k over the resident bank (BASELINE.json configs[4]): for each u[b] the k anchors
 * with the largest P(same), ties to the lower anchor index; topk_p fp32 [B,k], topk_idx int32 [B,k]. */
int mv_topk(mv_handle* h, const float* u, int B, int k, float* topk_p, int32_t* topk_idx);

/* ---- HBM-resident corpus (the MI355X-native form of the AllenNLP `evaluate` loop,
 *      predict_memory.py:103-110): the whole tokenised shard (1.2 M x 256 x int32 = 1.25 GB) and all
 *      per-IR results live in HBM; the host launches batches back-to-back and downloads once. ---- */
int mv_corpus_upload(mv_handle* h, const int32_t* ids, const int32_t* lens, int64_t n, int S);
/* Runs IRs [first, first+count) in batches of `batch`; asynchronous. keep_probs != 0 also keeps
 * P(same) for every (IR, anchor) pair (what make_output_human_readable serialises, l.169-191). */
int mv_corpus_run(mv_handle* h, int64_t first, int64_t count, int batch, int keep_probs);
/* Same, processing only the first s_eff tokens of every row in the range (0 = all S): for a corpus uploaded sorted by
 * length, a batch runs at its own longest member's length (padded to 64) instead of the corpus-wide S — the engine
 * form of padding each batch to its longest instance (predict_memory.py:97-101). Rows longer than s_eff must not be
 * in the range (their tail would be cut). */
int mv_corpus_run_len(mv_handle* h, int64_t first, int64_t count, int batch, int keep_probs, int s_eff);
/* Batches of the resident sweep in flight at once: 2 (default; consecutive batches alternate between two workspace
 * sets on two HIP streams and overlap on the GPU) or 1.  Results are identical either way. */
int mv_set_streams(mv_handle* h, int n);
/* best fp32 [count,2], best_idx int32 [count], p_same fp32 [count,G] (NULL unless kept). Synchronises. */
int mv_corpus_results(mv_handle* h, int64_t first, int64_t count, float* best, int32_t* best_idx, float* p_same);
/* MV_F16X8 only (always 0 in MV_F16).  The fp8 planes of the activations (raw residual stream, attention context, GELU output) use ONE
 * static scale: |x| <= 112 is representable; an element beyond it keeps its fp16 accuracy but loses its correction term (the precision of
 * MV_F16 for that element) — the computation never fails over it.  *clamped = the number of such elements since the handle was created
 * (or since the last call with reset != 0; a 64-bit device counter: it does not wrap); synchronises.  A non-zero count on a real checkpoint means the 1e-3
 * logit contract of model_memory.py:141 is no longer backed by the measurements in DESIGN.md section 2 for that model: the Python
 * wrapper warns once (binding.Engine).  NaN activations are not counted (the range test is a floating-point maximum, which skips them): they
 * propagate to the outputs as NaN, where they are visible.  No reference counterpart (the reference computes in fp32). */
int mv_x8_saturation(mv_handle* h, int64_t* clamped, int reset);

/* MV_F16X8 only (0 / 0 in MV_F16).  The concentration monitor: what the default form's 1e-3 is measured for is diffuse attention and attention sinks on the two
 * delimiter tokens — the [CLS] and the [SEP] token of a sequence sit in its rows 0 and 1 (the "special rows": A-side correction terms in every GEMM, V as hi + lo;
 * DESIGN.md section 2) — as trained BERT heads have them (custom_PTM_embedder.py:228 runs HF BertModel).  A head whose [CLS] row puts most of its mass on ONE
 * ORDINARY token is outside that envelope (measured 0.8 - 2.7e-3 with 50 - 80 % of the mass there, profiles/r06_n_sink_envelope.txt).  The attention kernel
 * therefore keeps, at no measurable cost, *max_collision = the maximum over every (sequence, head, layer) processed since the handle was created (or the last reset)
 * of sum_{j >= 2} p[CLS row][j]^2 (>= f^2 when one ordinary token holds the share f), *items_over = how many of them exceeded 0.25 (f > 0.5) and *items_total = how
 * many were looked at (sequences of at least 16 tokens); synchronises.  The Python wrapper warns once when more than 2 % of the items are over (binding.Engine).
 * No reference counterpart (the reference computes in fp32). */
int mv_attention_concentration(mv_handle* h, float* max_collision, int64_t* items_over, int64_t* items_total, int reset);

/* ---- multi-GPU exchange (the reference is single-process, predict_memory.py:103) --------------------
 * One process per GPU, contiguous corpus shards, no data-path collective; the ONE exchange is an all-gather of the
 * per-rank (score, label) statistics.  RCCL (librccl.so, opened at run time) is bound directly: the collective runs
 * on the engine's own stream and the process needs neither torch nor a launcher-specific runtime.
 * mv_comm_prepare: opens librccl.so and resolves its entry points (so that every rank can report "RCCL usable here" BEFORE
 * any rank enters the collective ncclCommInitRank).  mv_comm_unique_id: rank 0 draws the 128-byte ncclUniqueId (returns the
 * byte count).  mv_comm_init: ncclCommInitRank with those bytes — how they reach the other ranks is the host's business
 * (memvul_amd/distributed.py broadcasts them over its rendezvous socket; nothing is written to a shared temp directory and
 * nothing assumes one node).  mv_comm_allgather: `bytes_per_rank` bytes of host memory per rank -> world * bytes_per_rank
 * bytes on every rank, in rank order (staged through device buffers the library owns).  world == 1 needs no init: the
 * gather is then a copy (world == 1 WITH an id builds a real one-rank communicator: the single-GPU test of this path). */
int mv_comm_prepare(mv_handle* h);
int mv_comm_unique_id(mv_handle* h, void* id_out, int capacity);
int mv_comm_init(mv_handle* h, int rank, int world, const void* id, int id_bytes);
int mv_comm_allgather(mv_handle* h, const void* send, void* recv, int64_t bytes_per_rank);
int mv_comm_destroy(mv_handle* h);
/* What the transport is, as RCCL reports it: info[0] = ranks of the live communicator (ncclCommCount; 0 = no communicator),
 * info[1] = this rank in it (ncclCommUserRank), info[2] = RCCL version code (ncclGetVersion; 0 while librccl.so is not open),
 * info[3] = the world mv_comm_allgather gathers over.  bench.py --gpus N records it in its line. */
int mv_comm_info(mv_handle* h, int* info, int n);
/* GPUs visible to this process (hipGetDeviceCount; 0 without one; <= 0 means none): lets a test or a launcher decide whether a
 * two-rank RCCL run is possible here. */
int mv_device_count(void);

/* ---- measurement / test hooks ---------------------------------------------------------------- */

/* Per-kernel-class HIP-event timing on the engine's own stream. Classes: see mv_kernel_class_name. */
#define MV_NUM_KERNEL_CLASSES 14
int mv_profile_enable(mv_handle* h, int on);
/* Restrict the events to the classes whose bit is set (default: all).  bench.py times its K steps with events
 * on the dominant GEMM class only (the `roofline` figure) and takes the full breakdown in a separate pass, so
 * the timed region carries ~12 event pairs per step instead of ~90. */
int mv_profile_select(mv_handle* h, uint32_t class_mask);
/* Synchronises, adds up the recorded launches since the last read: ms[c], launches[c]; then clears. */
int mv_profile_read(mv_handle* h, double* ms, int64_t* launches, int n);
const char* mv_kernel_class_name(int cls);

/* Debug taps for per-kernel parity tests: run the encoder on (ids,lens) and stop after `n_layers`
 * encoder layers (0 = embeddings only, <0 = all), then copy an internal buffer to host.
 * buffer ids: 0 hidden fp32 [B*Sp,768]; 1 hidden fp16; 2 Q fp16 [B,12,Sp,64]; 3 K fp16 [B,12,Sp,64];
 * 4 V^T fp16 [B,12,64,Sp]; 5 attention context fp16 [B*Sp,768]; 6 FFN intermediate fp16 [B*Sp,3072].
 * (Sp = S rounded up to a multiple of 64, above 256 to a multiple of 128; buffers hold the state of the LAST executed layer; Q carries
 * the folded 1/8.  The pass takes the path its size selects — persistent kernels or the small-pass kernels — with last-layer pruning
 * off and the final LayerNorm applied, so buffer 0 is the normalised output of layer n_layers.) */
int mv_debug_encode(mv_handle* h, const int32_t* ids, const int32_t* lens, int B, int S, int n_layers);
int mv_debug_read(mv_handle* h, int buffer, void* dst, int64_t bytes);
/* Stand-alone GEMM check/bench on caller data: C[M,N] = A[M,K] (fp16 bits) x W[N,K]^T (fp16 bits)
 * + bias, fp32 out.  variant 0 = the 128^2-tile kernel of small passes (M,N multiples of 128), 19 = the 64^2-tile ring kernel of
 * the [CLS] tail (multiples of 64); K a multiple of 64.  iters > 1 repeats for timing; *ms = average milliseconds per launch. */
int mv_test_gemm(mv_handle* h, int variant, int M, int N, int K, const uint16_t* A, const uint16_t* W,
                 const float* bias, float* C, int iters, float* ms);
/* The persistent FFN-1 kernel (gemm_pp.h PP_GELU) on caller data with unit row statistics: out16 [M][N] = fp16 bits of
 * gelu(A W^T + bias) for fp32 A [M][K], W [N][K]; x8 != 0 runs the MV_F16X8 build (the operands are split into their fp16 / fp8
 * planes on the host) and, with out8, returns the [lo8 | hi8] e4m3 planes of the output [M][2 N].  M,N % 256, K % 128, K >= 256. */
int mv_test_gemm_pp(mv_handle* h, int x8, int M, int N, int K, const float* A, const float* W, const float* bias, uint16_t* out16,
                    uint8_t* out8, int iters, float* ms);
/* The host-side e4m3 encoder used for the MV_F16X8 weight planes (needs no GPU, h may be NULL elsewhere): out[i] = OCP e4m3fn bits of in[i]. */
int mv_test_e4m3(const float* in, uint8_t* out, int64_t n);

/* Host only, no GPU work, callable from any thread: the JSON line of one batch's records — replaces json.dumps(make_output_human_readable(...))
 * (model_memory.py:169-191 -> predict_memory.py:111), whose cost is CPython's repr() of B x G doubles.
 *   out = "[" + ", ".join(prefix_i + piece_0 + repr(p[i][0]) + ... + piece_{cols-1} + repr(p[i][cols-1]) + row_suffix for i in rows) + "]"
 * prefixes / pieces: the strings back to back, *_off[k] .. *_off[k + 1] the bytes of string k (rows + 1 / cols + 1 offsets); p: double [rows][cols];
 * every double is printed exactly as Python's repr(float) prints it.  MV_ERR_CAPACITY: `cap` too small; MV_ERR_INVALID: a non-finite value (json.dumps
 * spells those NaN / Infinity: the caller formats such a batch itself). */
int mv_format_records(const char* prefixes, const int64_t* prefix_off, int64_t rows, const char* pieces, const int64_t* piece_off, int64_t cols,
                      const char* row_suffix, const double* p, char* out, int64_t cap, int64_t* written);

#ifdef __cplusplus
}
#endif
#endif /* MEMVUL_HIP_H */
