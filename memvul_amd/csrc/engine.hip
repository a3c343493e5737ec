// libmemvul_hip.so — C ABI (include/memvul_hip.h) over the gfx950 kernels in this directory.
// Host side: weight staging/packing, workspace ownership, launch sequencing on one HIP stream,
// HIP-event profiling per kernel class, error translation.  No torch, no exceptions across the ABI.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: librccl.so is opened at run time (mv_comm_init), never linked
#include <unistd.h>

#include <algorithm>
#include <charconv>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/memvul_hip.h"
#include "common.h"
#include "gemm.h"
#include "gemm_pp.h"
#include "attention.h"
#include "attention_v2.h"
#include "misc_kernels.h"
#include "match_topk.h"

namespace {

enum KernelClass {
  KC_EMBED_LN = 0, KC_GEMM_QKV, KC_ATTENTION, KC_GEMM_OUT, KC_LN, KC_GEMM_FFN1, KC_GEMM_FFN2,
  KC_POOL_HEAD, KC_MATCH, KC_TOPK, KC_TEST_GEMM, KC_CLS_ROW_TERM, KC_GEMM_KV_LAST, KC_CLS_TAIL
};
const char* kKernelClassNames[MV_NUM_KERNEL_CLASSES] = {
    "embed_ln", "gemm_qkv", "attention", "gemm_attn_out", "layernorm", "gemm_ffn1_gelu", "gemm_ffn2",
    "pool_head", "match", "topk", "test_gemm", "cls_row_term", "gemm_kv_last", "cls_tail"};

thread_local std::string g_create_error;

struct HostTensor {
  std::vector<float> data;
  std::vector<int64_t> shape;
};

struct LayerW {
  half_t *wqkv = nullptr, *wo = nullptr, *w1 = nullptr, *w2 = nullptr;
  float *bqkv = nullptr, *bo = nullptr, *b1 = nullptr, *b2 = nullptr;
  // virtual LayerNorm (gemm_pp.h): the preceding LayerNorm folded in: W'' = rowcentre(W gamma), b' = b + W beta
  half_t *wqkv_f = nullptr, *w1_f = nullptr;
  float *bqkv_f = nullptr, *b1_f = nullptr;
  // MV_F16X8 (gemm_pp.h): fp8 planes [hi8 | lo8] of the four GEMM weights the persistent path uses, rows of 2 K bytes, and the
  // E8M0 scale word of each GEMM's correction sweep (2^-(11 + MV_X8_ACT_SHIFT + the matrix' own shift))
  uint8_t *wqkv_f8 = nullptr, *wo8 = nullptr, *w1_f8 = nullptr, *w28 = nullptr;
  int sc_qkv = 0, sc_o = 0, sc_1 = 0, sc_2 = 0;
  // MV_F16X8, last layer only: fp32 transposed ([k][n]) weights of the [CLS] tail (misc_kernels.h dense768_kernel)
  float *wqT32 = nullptr, *woT32 = nullptr, *w1T32 = nullptr, *w2T32 = nullptr;
  float *ln1g = nullptr, *ln1b = nullptr, *ln2g = nullptr, *ln2b = nullptr;
};

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// fp32 -> fp16 bits, round-to-nearest-even (same result as numpy astype(float16))
inline uint16_t f32_to_f16_bits(float f) {
  uint32_t x;
  std::memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  x &= 0x7fffffffu;
  if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((x > 0x7f800000u) ? 0x200u : 0));
  if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);  // rounds to >= 65520 -> inf
  if (x < 0x38800000u) {                                     // subnormal half or zero
    if (x < 0x33000000u) return (uint16_t)sign;              // < 2^-25 -> 0
    const int e = (int)(x >> 23);
    uint32_t m = (x & 0x7fffffu) | 0x800000u;
    const int shift = 126 - e;  // 14..24 -> bits to drop
    const uint32_t half_m = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1), halfway = 1u << (shift - 1);
    uint32_t r = half_m;
    if (rem > halfway || (rem == halfway && (half_m & 1))) r++;
    return (uint16_t)(sign | r);
  }
  const uint32_t e = (x >> 23) - 112, m = x & 0x7fffffu;
  uint32_t h = (e << 10) | (m >> 13);
  const uint32_t rem = m & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) h++;
  return (uint16_t)(sign | h);
}
inline float f16_bits_to_f32(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 0x1f, m = h & 0x3ffu, x;
  if (e == 0) {
    if (m == 0) x = sign;
    else {
      int sh = 0;
      while (!(m & 0x400u)) { m <<= 1; sh++; }
      m &= 0x3ffu;
      x = sign | ((uint32_t)(113 - sh) << 23) | (m << 13);
    }
  } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
  else x = sign | ((e + 112) << 23) | (m << 13);
  float f;
  std::memcpy(&f, &x, 4);
  return f;
}
inline float bf16_bits_to_f32(uint16_t h) {
  uint32_t x = (uint32_t)h << 16;
  float f;
  std::memcpy(&f, &x, 4);
  return f;
}

struct ProfRec {
  int cls;
  hipEvent_t e0, e1;
};

}  // namespace

// Activation buffers of ONE in-flight batch and the stream its kernels run on
struct Work {
  hipStream_t stream = nullptr;
  int32_t *d_ids = nullptr, *d_lens = nullptr;  // host-path inputs
  float* xres = nullptr;                        // residual stream fp32 [T][768]
  half_t *x16 = nullptr, *q = nullptr, *k = nullptr, *vt = nullptr, *ctx = nullptr, *h16 = nullptr;
  half_t *vt_lo = nullptr, *q_lo = nullptr, *k_lo = nullptr;  // MV_F16X8, passes of padded length <= 128: second fp16 planes of V^T, Q, K (attention_v2.h VLO)
  float *u = nullptr, *pooled = nullptr;
  float *logits = nullptr, *probs = nullptr, *psame = nullptr, *best = nullptr;
  int32_t* best_idx = nullptr;
  float* topk_p = nullptr;
  int32_t* topk_idx = nullptr;
  float *part_p = nullptr, *part_q = nullptr;   // per-chunk top-k candidates of the fused matcher (G > 256)
  int32_t* part_i = nullptr;
  float* u_in = nullptr;                        // host-provided embeddings for mv_match / mv_topk
  float *c32 = nullptr, *cq = nullptr;          // [CLS]-row buffers of the pruned last layer
  float* ch32 = nullptr;                        // MV_F16X8: the fp32 [CLS] tail's FFN intermediate [Bp][3072]
  half_t *c16 = nullptr, *cctx = nullptr, *ch16 = nullptr;
  float *lnstats = nullptr, *lnpart = nullptr;  // the two vstats buffers [T][3][2] of the virtual LayerNorm (layer input / mid-layer;
                                                // each residual GEMM reads one, writes the other)
  half_t* xlo = nullptr;                        // MV_F16: lo plane of the two-plane raw stream (PP_RESLN3), allocated by mv_finalize_weights.  MV_F16X8 has none: the stream's low part is the lo8 plane of x8 (+ st_lo)
  uint8_t *x8 = nullptr, *ctx8 = nullptr, *h8 = nullptr;  // MV_F16X8: [lo8 | hi8] planes of the raw stream [T][1536], the attention
                                                          // context [T][1536] and the GELU output [T][6144]
  half_t* cls_lo = nullptr;   // MV_F16X8, special rows (rows 0, 1 of every sequence: its [CLS] and [SEP] token): 2^11 x the low parts of those rows of the NEXT GEMM's A operand,
                              // compact [2 Bp][3072] fp16, written by the producing kernel's epilogue (GemmArgs::sp_lo_out / AttnArgs::sp_lo_out / embed_ln_kernel)
  half_t* st_lo = nullptr;    // ... 2^11 x the low parts of those rows of the RAW STREAM [2 Bp][768]: the A operand of the QKV / FFN-1 row terms, and — in the [CLS]-row form,
                              // with the stream of every other row being hi + the lo8 plane of its fp8 planes (gemm.h GemmArgs::out16b) — what the residual GEMMs read back
  float* cls_corr = nullptr;  // ... and 2^11 x their A-side correction term A_lo W_hi^T [2 Bp][3072] (GemmArgs::cls_corr)
  half_t* vlo_sp = nullptr;   // 2^11 x the low parts of V of the special rows [B 12][64][2] (GemmArgs::vlo_sp -> AttnArgs::vlo_sp)
  int32_t* tile_both = nullptr;  // cls_aside: per 256-row tile of the pass, non-zero = its sequence is shorter than cls_min_len (GemmArgs::tile_both)
};

struct mv_handle {
  int device = 0;
  mv_config cfg{};
  std::string err;
  bool finalized = false;
  int compute_dtype = MV_F16;
  bool precise = false;    // MV_F16X8: every persistent GEMM adds the fp8 correction sweep (gemm_pp.h X8)
  std::map<std::string, HostTensor> staged;
  std::vector<void*> allocs;

  // weights
  float *wemb = nullptr, *pemb = nullptr, *temb = nullptr, *embg = nullptr, *embb = nullptr;
  std::vector<LayerW> L;
  float *WpT = nullptr, *bp = nullptr, *WhT = nullptr, *bh = nullptr, *Wm = nullptr;
  int P = MV_PROJ;  // width of the embedding the matcher runs on: 512 = header output (use_header, every reference config),
                    // 768 = the pooler output itself (use_header = False, model_memory.py:69-73): mv_config.proj_dim

  // workspaces: two sets, each with its own stream.  mv_corpus_run alternates the batches of a sweep between them,
  // so two batches are in flight on the GPU at once: the persistent kernels of one batch fill the CUs the other
  // batch's kernel tails, small kernels and memory phases leave idle (+5 % issue reports/s, scripts/dual_stream_probe.py).
  // Every other entry point works on set 0 (`w` points at the set in use).
  int64_t cap_tokens = 0;  // rows every activation buffer holds (multiple of 128, + slack)
  Work work[2];
  Work* w = &work[0];
  int n_streams = 2;       // sets in use by the resident sweep (mv_set_streams); env MEMVUL_STREAMS=1: only one is created
  int n_alloc = 2;         // sets created
  bool dual_pending = false;  // work[1] may still be running a batch
  int rr = 0;                 // workspace set of the next resident-sweep batch
  float* anchors = nullptr;
  int n_anchors = 0;
  unsigned long long* attn_conc = nullptr;  // MV_F16X8: [0] max collision mass of the [CLS] row on ordinary keys (float bits), [1] items above 0.25 (AttnArgs::conc)
  unsigned long long* x8_sat = nullptr;  // MV_F16X8: device counter (64-bit: it cannot wrap within a run) of activation elements beyond the fp8 planes' range (mv_x8_saturation)

  // resident corpus
  int32_t *c_ids = nullptr, *c_lens = nullptr;
  std::vector<int32_t> c_lens_host;  // the lengths as uploaded (encode_dev's min_len of each pass)
  struct {  // mv_forward_ragged: the batch in length order and its results before they go back to the caller's row order
    std::vector<int32_t> ids, lens, idx;
    std::vector<float> logits, probs, best, embed;
  } ragged;
  struct RaggedSlot {  // mv_forward_ragged_begin / _end: one batch in flight per workspace set, its staging in PINNED host memory (the copies really are asynchronous)
    bool busy = false;
    int B = 0, G = 0;
    bool logits = false, probs = false, embed = false;
    std::vector<int> order;
    int32_t *ids = nullptr, *lens = nullptr, *idx = nullptr;   // [cap_tokens], [max_batch], [max_batch]
    float *lg = nullptr, *pr = nullptr, *best = nullptr, *emb = nullptr;  // [max_batch][max_anchors][2] x 2, [max_batch][2], [max_batch][P]
  } rslot[2];
  int rnext = 0;
  std::vector<void*> pinned;
  int64_t c_n = 0;
  int c_S = 0;
  float* c_best = nullptr;
  int32_t* c_idx = nullptr;
  float* c_psame = nullptr;
  int64_t c_psame_rows = 0;
  int c_G = 0;

  // last-layer pruning ([CLS] rows only after the last layer's K / V projection) and its compact buffers
  bool cls_prune = true;   // env MEMVUL_CLS_PRUNE=0 disables
  int pp_gn_max = 4;       // raster group width cap of the persistent GEMM (env MEMVUL_GN_MAX: the A/B of profiles/r04_*)
  int pp_raster = 0;       // env MEMVUL_RASTER=1: the A-stationary raster where it applies (FFN-1, QKV at full-size grids)
  bool short_vlo = true;   // MV_F16X8, env MEMVUL_SHORT_VLO=0 disables: passes of padded length <= 128 carry V and P as hi + lo fp16 planes through attention
                           // (attention_v2.h VLO): the fp16 storage of V and P is what is left of the precise mode's error and short sequences average it least
  bool cls_aside = true;   // MV_F16X8, the [CLS]-row form (default; env MEMVUL_CLS_ASIDE=0 = both correction terms in every row, the form of rounds 3-4): passes of
                           // padded length 256 / 512 sweep the weight-side term only in every GEMM (the Q block of the QKV projection keeps both) and add the A-side
                           // term for the [CLS] row of each sequence alone (a skinny fp16 GEMM over those B rows in front of each launch, GemmArgs::cls_corr): the
                           // pooler reads only that row, every other row's A-side rounding reaches it averaged over the keys.  +14 % at the same error (r05_j*, r05_k*)
  int cls_min_len = 128;   // ... for sequences of at least this many tokens (env MEMVUL_CLS_ASIDE_MIN_LEN): a short sequence averages over few keys (model: 1.2 -
                           // 1.6x the error at 16 - 128 tokens), so its row tiles run the both-terms form, bit for bit (GemmArgs::tile_both, cls_tile_flags_kernel)
  int qkv_aside_mask = 0;  // MV_F16X8: which of the Q / K / V blocks of the QKV projection sweep the A-side correction term for EVERY row (bit 0 / 1 / 2;
                           // gemm_pp.h x8_aside_mask).  Default (round 6, second half): none — the special rows get the term from their row term in every block, and an
                           // ordinary row's Q rounding, like its K and V rounding, reaches the pooler only through attention, averaged over the keys (model: q / none / qkv
                           // within 7 % of each other, scripts/r06_qkv_model.py; GPU, 60 draws: +3 % error for +2.7 % issue reports/s, profiles/r06_m_*).  Rounds 4 - 6a: Q ("q").
                           // env MEMVUL_QKV_ASIDE = a subset of "qkv", "" or "none" ("qkv" = round 3's form)

  // profiling
  uint32_t prof_mask = 0xffffffffu;  // kernel classes that get HIP events while profiling is on
  bool prof = false;
  std::vector<ProfRec> recs;
  std::vector<hipEvent_t> free_events;

  // GEMM path (env MEMVUL_GEMM_TILE): 0 auto (persistent ping-pong kernels when the pass fills the chip, else the
  // one-tile-per-workgroup kernels on an fp32 stream), 128 forces the small path, 512 the persistent one.
  int gemm_tile = 0;
  int num_cu = 256;

  // debug
  int dbg_B = 0, dbg_Sp = 0;

  // multi-GPU exchange (mv_comm_*): RCCL entry points resolved from librccl.so at run time
  void* rccl_lib = nullptr;
  ncclComm_t comm = nullptr;
  int comm_rank = 0, comm_world = 1;
  decltype(&ncclGetUniqueId) p_ncclGetUniqueId = nullptr;
  decltype(&ncclCommInitRank) p_ncclCommInitRank = nullptr;
  decltype(&ncclAllGather) p_ncclAllGather = nullptr;
  decltype(&ncclCommDestroy) p_ncclCommDestroy = nullptr;
  decltype(&ncclGetErrorString) p_ncclGetErrorString = nullptr;
  decltype(&ncclGetVersion) p_ncclGetVersion = nullptr;      // optional (mv_comm_info)
  decltype(&ncclCommCount) p_ncclCommCount = nullptr;        // optional
  decltype(&ncclCommUserRank) p_ncclCommUserRank = nullptr;  // optional
  void *comm_send = nullptr, *comm_recv = nullptr;
  int64_t comm_send_cap = 0, comm_recv_cap = 0;
};

namespace {

int fail(mv_handle* h, int code, const std::string& msg) {
  if (h) h->err = msg; else g_create_error = msg;
  return code;
}

// No C++ exception crosses the ABI (include/memvul_hip.h): every entry point is a function-try-block whose handler lands here
// (std::bad_alloc of the host-side staging vectors / maps -> MV_ERR_NOMEM, anything else -> MV_ERR_INTERNAL).
int on_exception(mv_handle* h) noexcept {
  int code = MV_ERR_INTERNAL;
  const char* what = "unknown C++ exception";
  try {
    throw;
  } catch (const std::bad_alloc&) {
    code = MV_ERR_NOMEM;
    what = "out of host memory";
  } catch (const std::exception& e) {
    what = e.what();
  } catch (...) {
  }
  try {
    fail(h, code, std::string("internal: ") + what);
  } catch (...) {  // not even the message could be stored
  }
  return code;
}

#define HIPCHK(h, expr)                                                                             \
  do {                                                                                              \
    hipError_t _e = (expr);                                                                         \
    if (_e != hipSuccess)                                                                           \
      return fail(h, MV_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));                \
  } while (0)

template <typename T>
int dev_alloc(mv_handle* h, T** p, int64_t count, bool zero = true) {
  void* d = nullptr;
  const size_t bytes = (size_t)count * sizeof(T);
  hipError_t e = hipMalloc(&d, bytes ? bytes : 16);
  if (e != hipSuccess) return fail(h, MV_ERR_NOMEM, std::string("hipMalloc failed: ") + hipGetErrorString(e));
  if (zero) {
    e = hipMemsetAsync(d, 0, bytes ? bytes : 16, h->w->stream);
    if (e != hipSuccess) return fail(h, MV_ERR_HIP, std::string("hipMemset failed: ") + hipGetErrorString(e));
  }
  h->allocs.push_back(d);
  *p = (T*)d;
  return MV_OK;
}
void dev_free(mv_handle* h, void* p) {
  if (!p) return;
  for (auto it = h->allocs.begin(); it != h->allocs.end(); ++it)
    if (*it == p) { h->allocs.erase(it); break; }
  hipFree(p);
}

hipEvent_t get_event(mv_handle* h) {
  if (!h->free_events.empty()) {
    hipEvent_t e = h->free_events.back();
    h->free_events.pop_back();
    return e;
  }
  hipEvent_t e;
  hipEventCreate(&e);
  return e;
}

struct ProfScope {
  mv_handle* h;
  ProfRec rec;
  bool on;
  ProfScope(mv_handle* h_, int cls) : h(h_), on(h_->prof && ((h_->prof_mask >> cls) & 1u)) {
    if (on) {
      rec.cls = cls;
      rec.e0 = get_event(h);
      rec.e1 = get_event(h);
      hipEventRecord(rec.e0, h->w->stream);
    }
  }
  ~ProfScope() {
    if (on) {
      hipEventRecord(rec.e1, h->w->stream);
      h->recs.push_back(rec);
    }
  }
};

int launch_check(mv_handle* h, const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(h, MV_ERR_HIP, std::string("launch ") + what + ": " + hipGetErrorString(e));
  return MV_OK;
}

int choose_gn(int tn, int gn_max) {
  int g = 1;
  for (int d = 1; d <= gn_max && d <= tn; ++d)
    if (tn % d == 0) g = d;
  return g;
}

template <int EPI>
int launch_gemm128(mv_handle* h, int cls, GemmArgs a) {
  if (a.M % 128 || a.N % 128 || a.K % 64) return fail(h, MV_ERR_INVALID, "gemm128: M,N % 128, K % 64 required");
  a.GN = choose_gn(a.N / 128, 8);
  const int grid = (a.M / 128) * (a.N / 128);
  ProfScope ps(h, cls);
  hipLaunchKernelGGL((gemm128_kernel<EPI>), dim3(grid), dim3(256), G128_LDS_BYTES, h->w->stream, a);
  return launch_check(h, "gemm128");
}

// skinny problems (the [CLS] tail of the pruned last layer: M = batch rows): 64 x 64 tiles on a 4-stage LDS ring,
// 4x the workgroups of the 128^2 kernel and a K loop that is DMA-latency-bound per step rather than per tile
constexpr int RING64_LDS = 4 * (64 + 64) * 64 * 2;
template <int EPI>
int launch_ring64(mv_handle* h, int cls, GemmArgs a) {
  if (a.M % 64 || a.N % 64 || a.K % 64) return fail(h, MV_ERR_INVALID, "gemm_ring: shape not a multiple of the 64 x 64 x 64 tile");
  a.GN = choose_gn(a.N / 64, 8);
  const int grid = (a.M / 64) * (a.N / 64);
  ProfScope ps(h, cls);
  hipLaunchKernelGGL((gemm_ring_kernel<EPI, 1, 1, 2, 2, 64, 4, 2>), dim3(grid), dim3(256), RING64_LDS, h->w->stream, a);
  return launch_check(h, "gemm_ring");
}

// The persistent ping-pong GEMM (gemm_pp.h): one workgroup per CU walks the 256^2 output tiles.  a.A8 set = the
// MV_F16X8 build (a second, fp8 sweep over [A8 | W8]).
template <int PPEPI>
int launch_pp(mv_handle* h, int cls, GemmArgs a) {
  constexpr int RAW = PPEPI != PP_RESLN3;
  if (a.M % 256 || a.N % 256 || a.K % 128 || a.K < 256 || a.N > MV_INTER)
    return fail(h, MV_ERR_INVALID, "gemm_pp: M,N % 256, K % 128, K >= 256, N <= 3072 required");  // K >= 256: the RAW kernels stage the
                                                                                              // next tile's statistics at K-tile 2
  if (!a.bias || !a.lnstats) return fail(h, MV_ERR_STATE, "internal: gemm_pp without bias / row statistics");
  // a weight-side-only fp8 sweep walks K / 128 K-tiles IN PAIRS (gemm_pp.h two_ktiles): the staging and consume cursors only stay in
  // step when that count is even
  if (a.A8 && (a.x8_terms == 1 || a.x8_terms == 3) && a.K % 256)
    return fail(h, MV_ERR_INVALID, "gemm_pp: a weight-side-only fp8 correction sweep needs K % 256 == 0");
  a.GN = choose_gn(a.N / 256, h->pp_gn_max);  // widths 2 / 3 / 6 / 12 measured: 4 (or the largest divisor below it) is the fastest
  const int tiles = (a.M / 256) * (a.N / 256);
  const int grid = tiles < h->num_cu ? tiles : h->num_cu;
  // the A-stationary raster (gemm_pp.h raster_pp; MEMVUL_RASTER=1): only where its windows tile the sequence exactly
  a.raster_mode = (h->pp_raster == 1 && a.N / 256 > a.GN && ((a.M / 256) * a.GN) % grid == 0) ? 1 : 0;
  const int lds = RAW ? PP_LDS_BYTES_RAW : PP_LDS_BYTES;
  ProfScope ps(h, cls);
  if (a.A8) {
    if (!a.W8 || (PPEPI != PP_QK && !a.out8)) return fail(h, MV_ERR_STATE, "internal: MV_F16X8 GEMM without its fp8 planes");
    hipLaunchKernelGGL((gemm_pp_kernel<PPEPI, RAW, 1>), dim3(grid), dim3(512), lds, h->w->stream, a);
  } else {
    hipLaunchKernelGGL((gemm_pp_kernel<PPEPI, RAW, 0>), dim3(grid), dim3(512), lds, h->w->stream, a);
  }
  return launch_check(h, "gemm_pp");
}

// path choice: the persistent kernels need enough 256^2 tiles to fill the CUs (one workgroup each); both residual GEMMs
// have N = 768 and every K is a multiple of 128, so ONE predicate (on the padded token count) decides the path of a pass
bool pp_selected(const mv_handle* h, int64_t M) {
  if (M % 256) return false;
  if (h->gemm_tile == 128) return false;
  return h->gemm_tile == 512 || h->precise || (M / 256) * (MV_HIDDEN / 256) >= 256;
}

// the mid-size / skinny GEMMs of a pass that does not fill the chip (and of the [CLS] tail)
template <int EPI>
int launch_small(mv_handle* h, int cls, const GemmArgs& a) {
  if (h->gemm_tile == 0 && a.M <= 512 && a.M % 64 == 0 && a.N % 64 == 0) return launch_ring64<EPI>(h, cls, a);
  return launch_gemm128<EPI>(h, cls, a);
}

// K7 + K8: pooler on the [CLS] rows (row_stride floats apart), then the header
int pool_head(mv_handle* h, const float* x, size_t row_stride, int B, float* u_out) {
  const unsigned gx = (unsigned)((B + 31) / 32);
  hipLaunchKernelGGL(dense768_kernel<0>, dim3(gx, MV_HIDDEN / 32), dim3(512), 0, h->w->stream, x, row_stride, B, h->WpT, h->bp,
                     MV_HIDDEN, h->P == MV_HIDDEN ? u_out : h->w->pooled);
  if (int rc = launch_check(h, "pooler")) return rc;
  if (h->P == MV_HIDDEN) return MV_OK;  // use_header = False: the pooler output is the embedding
  hipLaunchKernelGGL(dense768_kernel<1>, dim3(gx, MV_PROJ / 32), dim3(512), 0, h->w->stream, h->w->pooled, (size_t)MV_HIDDEN, B, h->WhT,
                     h->bh, MV_PROJ, u_out);
  return launch_check(h, "header");
}

// padded sequence length of a pass: attention_v2 runs 64-key blocks up to 256 and 128-key chunks above
inline int padded_len(int S_in) { return (int)round_up(S_in, S_in <= 256 ? 64 : 128); }

int launch_attention(mv_handle* h, const int32_t* d_lens, int B, int Sp, bool x8, bool sp_out = false) {
  const bool vlo = x8 && h->short_vlo && Sp <= 128;  // the QKV projection of this pass wrote V^T's lo plane (encode_dev: the same predicate)
  AttnArgs a{h->w->q, h->w->k, h->w->vt, d_lens, h->w->ctx, Sp, B, x8 ? h->w->ctx8 : nullptr, h->x8_sat, vlo ? h->w->vt_lo : nullptr,
             vlo ? h->w->q_lo : nullptr, vlo ? h->w->k_lo : nullptr,
             (x8 && !vlo) ? h->w->vlo_sp : nullptr,     // special rows: V of keys 0, 1 as hi + lo (the two-plane short passes carry every key's lo plane)
             x8 ? h->attn_conc : nullptr,               // concentration monitor (mv_attention_concentration)
             sp_out ? h->cls_min_len : 0,               // [CLS]-row form: no lo8 plane of the context for the sequences that take it
             sp_out ? h->w->cls_lo : nullptr};
  ProfScope ps(h, KC_ATTENTION);
  if (vlo) {
    const int nkb = Sp / 64, items = B * MV_HEADS;
    const int slots = h->num_cu * (nkb == 1 ? 2 : 1);  // resident workgroups by LDS: 64 / 128 KiB each
    const int grid = items < slots ? items : slots;
    if (nkb == 1) hipLaunchKernelGGL((attention_v2_kernel<1, 1, 1, 1>), dim3(grid), dim3(128), ATT2_LDS_BYTES_VLO(1), h->w->stream, a, items);
    else hipLaunchKernelGGL((attention_v2_kernel<2, 1, 1, 1>), dim3(grid), dim3(256), ATT2_LDS_BYTES_VLO(2), h->w->stream, a, items);
  } else if (Sp <= 256) {
    const int nkb = Sp / 64, items = B * MV_HEADS;
    const int slots = h->num_cu * (nkb == 1 ? 4 : nkb == 2 ? 2 : 1);  // resident workgroups: 8 waves and <= 128 KiB LDS per CU
    const int grid = items < slots ? items : slots;
#define MV_ATT(NKB)                                                                                                              \
    if (x8) hipLaunchKernelGGL((attention_v2_kernel<NKB, 1, 1>), dim3(grid), dim3(NKB * 128), ATT2_LDS_BYTES(NKB), h->w->stream, a, items); \
    else hipLaunchKernelGGL((attention_v2_kernel<NKB, 1, 0>), dim3(grid), dim3(NKB * 128), ATT2_LDS_BYTES(NKB), h->w->stream, a, items)
    switch (nkb) {
      case 1: MV_ATT(1); break;
      case 2: MV_ATT(2); break;
      case 3: MV_ATT(3); break;
      default: MV_ATT(4); break;
    }
#undef MV_ATT
  } else if (Sp == 384 || Sp == 512) {
    // chunks of 128 keys per (row, head, 128-query block) through the same ring: 64 score registers per lane, two
    // workgroups of 4 waves per CU; consecutive units of a workgroup are the query blocks of one head (K / V^T from L2)
    const int nch = Sp / 128, units = B * MV_HEADS * nch;
    const int grid = units < 2 * h->num_cu ? units : 2 * h->num_cu;
    if (nch == 3 && x8) hipLaunchKernelGGL((attention_v2_kernel<2, 3, 1>), dim3(grid), dim3(256), ATT2_LDS_BYTES(2), h->w->stream, a, units);
    else if (nch == 3) hipLaunchKernelGGL((attention_v2_kernel<2, 3, 0>), dim3(grid), dim3(256), ATT2_LDS_BYTES(2), h->w->stream, a, units);
    else if (x8) hipLaunchKernelGGL((attention_v2_kernel<2, 4, 1>), dim3(grid), dim3(256), ATT2_LDS_BYTES(2), h->w->stream, a, units);
    else hipLaunchKernelGGL((attention_v2_kernel<2, 4, 0>), dim3(grid), dim3(256), ATT2_LDS_BYTES(2), h->w->stream, a, units);
  } else {
    return fail(h, MV_ERR_INVALID, "internal: attention at a padded length other than 64 .. 256 / 384 / 512");
  }
  return launch_check(h, "attention");
}

// the shortest sequence of a pass, from the host copy of its lengths
inline int pass_min_len(const int32_t* lens, int n) {
  int m = INT32_MAX;
  for (int i = 0; i < n; ++i) m = lens[i] < m ? lens[i] : m;
  return n > 0 ? m : 0;
}

// ---- encoder: ids (device) -> u (device, [B][512]); stops after n_layers (<0: all) ------------
// Two paths, chosen by the size of the pass (pp_selected):
//   * bench scale: the persistent GEMMs on the two-plane raw stream with the virtual LayerNorm (gemm_pp.h), five launches per
//     layer; compute dtype MV_F16X8 adds the fp8 correction sweep to each GEMM and the [lo8 | hi8] planes to each producer;
//   * small passes: one-tile-per-workgroup GEMMs (gemm.h) on an fp32 stream with explicit LayerNorm kernels.
// The last layer is pruned to the [CLS] rows when the pooler follows (cls_prune); `full` (debug taps) disables that and
// leaves the normalised fp32 stream of the last layer run in xres.
int encode_dev(mv_handle* h, const int32_t* d_ids, const int32_t* d_lens, int min_len, int B, int S_in, int n_layers, float* u_out,
               bool full = false, int pitch = 0) {  // min_len: the shortest sequence of the pass as the HOST knows it (pass_min_len; 0 = unknown)
  if (pitch <= 0) pitch = S_in;  // ints between the rows of d_ids
  const mv_config& c = h->cfg;
  const int Sp = padded_len(S_in);
  const int64_t M = (int64_t)B * Sp, Mpad = round_up(M, 256);
  if (S_in > c.max_pos) return fail(h, MV_ERR_INVALID, "sequence longer than max_pos");
  if (Mpad > h->cap_tokens) return fail(h, MV_ERR_CAPACITY, "B*S exceeds mv_config.max_tokens");
  if (n_layers < 0 || n_layers > c.layers) n_layers = c.layers;
  h->dbg_B = B;
  h->dbg_Sp = Sp;
  const bool big = pp_selected(h, Mpad);  // persistent GEMMs, raw stream as hi + a low part (x16 = hi; MV_F16: xlo, MV_F16X8: the lo8 plane of x8 + st_lo), virtual LayerNorm
  const bool x8 = h->precise;             // MV_F16X8: + fp8 correction sweeps (forces the persistent path, pp_selected)
  const bool prune = !full && h->cls_prune && u_out && n_layers == c.layers && n_layers > 0;
  // The [CLS]-row form (mv_handle::cls_aside): every persistent GEMM of this pass sweeps the weight-side correction term only (x8_terms = 1) and the
  // A-side term A_lo W_hi^T is formed for the B [CLS] rows alone: their low parts (2^11 x, fp16) gathered from the operand's lo plane (raw stream) or
  // lo8 plane (context, GELU output), one skinny fp16 GEMM [B x K] x [K x N], and the launch adds the result to those rows' accumulators
  // (gemm_pp.h GemmArgs::cls_corr).  Passes of padded length 256 / 512: a 256-row tile then belongs to ONE sequence, so the form of a sequence
  // depends on its own length alone (cls_tile_flags_kernel: sequences shorter than cls_min_len keep the both-terms form, tile by tile) and a row's
  // result stays independent of the batch it travels in.  Passes of padded length 192 / 384 (a tile there spans two sequences, a per-tile rule would mix the
  // forms inside a sequence): the form for the WHOLE pass when its shortest sequence has cls_min_len tokens — what a length-sorted sweep hands over by
  // construction (ModelMemory.sweep / Engine.bucketed_sweep: a pass at 192 holds 129 .. 192 tokens, at 384 257 .. 384) — else the both-terms form for the whole pass.
  const bool one_seq_tiles = Sp == 256 || Sp == 512;
  const bool whole_pass = (Sp == 192 || Sp == 384) && min_len >= h->cls_min_len;
  const bool cls_as = big && x8 && h->cls_aside && (one_seq_tiles || whole_pass);
  if (cls_as && one_seq_tiles) {
    const int ntile = (int)(Mpad / 256);
    hipLaunchKernelGGL(cls_tile_flags_kernel, dim3((unsigned)((ntile + 255) / 256)), dim3(256), 0, h->w->stream, d_lens, B, Sp, h->cls_min_len, ntile,
                       h->w->tile_both);
    if (int rc = launch_check(h, "cls_tile_flags")) return rc;
  }
  // Special rows (round 6): rows 0 and 1 of every sequence hold its [CLS] and its [SEP] token (embed_ln_kernel swaps the last token into row 1) — the token the
  // pooler reads and the two tokens trained BERT heads use as attention sinks, i.e. the rows whose roundings can reach the pooler un-averaged.  For them every
  // GEMM whose sweep carried the weight-side term only gets the A-side term from a skinny GEMM over the 2 B compact rows the PRODUCER's epilogue left in cls_lo
  // (no gather launch), and attention adds p[:, 0..1] V_lo[0..1].  The K and V blocks of the QKV projection take it in every pass of this compute dtype (they
  // never sweep the A-side term for all rows by default), the other three GEMMs where the [CLS]-row form is in force.
  const bool special = big && x8;
  auto row_term = [&](const half_t* A, const half_t* W, int N, int K) -> int {  // cls_corr [2 B][N] = A [2 B][K] W^T (both 2^11 x); A = st_lo (stream) or cls_lo (context, GELU output)
    ProfScope ps(h, KC_CLS_ROW_TERM);
    GemmArgs t{};
    t.M = (int)round_up(2 * B, 64); t.Mreal = 2 * B; t.S = 64; t.A = A; t.W = W; t.N = N; t.K = K; t.outf = h->w->cls_corr;
    t.GN = choose_gn(N / 64, 8);
    hipLaunchKernelGGL((gemm_ring_kernel<EPI_F32, 1, 1, 2, 2, 64, 4, 2>), dim3((unsigned)((t.M / 64) * (N / 64))), dim3(256), RING64_LDS, h->w->stream, t);
    return launch_check(h, "row term gemm_ring");
  };
  const unsigned ln_grid = (unsigned)((M + 3) / 4);
  {
    ProfScope ps(h, KC_EMBED_LN);
    if (big)
      hipLaunchKernelGGL(embed_ln_kernel<true>, dim3(ln_grid), dim3(256), 0, h->w->stream, d_ids, pitch, S_in, Sp, (int)M, c.vocab_size,
                         h->wemb, h->pemb, h->temb, h->embg, h->embb, c.ln_eps, h->w->xres, h->w->x16, h->w->lnstats,
                         x8 ? (half_t*)nullptr : h->w->xlo, x8 ? h->w->x8 : (uint8_t*)nullptr, h->x8_sat, special ? d_lens : (const int32_t*)nullptr,
                         special ? h->w->st_lo : (half_t*)nullptr);
    else
      hipLaunchKernelGGL(embed_ln_kernel<false>, dim3(ln_grid), dim3(256), 0, h->w->stream, d_ids, pitch, S_in, Sp, (int)M, c.vocab_size,
                         h->wemb, h->pemb, h->temb, h->embg, h->embb, c.ln_eps, h->w->xres, h->w->x16, (float*)nullptr,
                         (half_t*)nullptr, (uint8_t*)nullptr, (unsigned long long*)nullptr);
    if (int rc = launch_check(h, "embed_ln")) return rc;
  }
  // big: the LayerNorm whose statistics are pending in the vstats buffers — gamma / beta the next residual GEMM applies
  const float *pend_g = h->embg, *pend_b = h->embb;
  auto run_ln = [&](float* x32, half_t* x16, int rows, const float* g, const float* b) -> int {
    ProfScope ps(h, KC_LN);
    hipLaunchKernelGGL(ln_kernel<true>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, h->w->stream, x32, x16, rows, g, b, c.ln_eps,
                       (float*)nullptr);
    return launch_check(h, "layernorm");
  };
  auto final_ln = [&](const float* g, const float* b) -> int {  // two-plane raw stream -> normalised fp32 rows (pooler / debug taps)
    const size_t n4 = (size_t)M * MV_HIDDEN / 4;
    hipLaunchKernelGGL(hilo_to_f32_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, h->w->stream, h->w->x16, h->w->xlo, n4, h->w->xres,
                       special ? (const half_t*)h->w->st_lo : (const half_t*)nullptr, Sp, (const uint8_t*)h->w->x8);
    if (int rc = launch_check(h, "hilo_to_f32")) return rc;
    return run_ln(h->w->xres, h->w->x16, (int)M, g, b);
  };
  if (big && n_layers == 0) { if (int rc = final_ln(h->embg, h->embb)) return rc; }
  // big: st_in = vstats of the layer's input rows (embedding / previous FFN-2), st_mid = of the rows after the
  // attention-output projection; no statistics kernel in between (gemm_pp.h)
  float *st_in = h->w->lnstats, *st_mid = h->w->lnpart;
  for (int l = 0; l < n_layers; ++l) {
    const LayerW& w = h->L[l];
    const bool last = (l == n_layers - 1);
    GemmArgs g{};
    g.M = (int)Mpad; g.Mreal = (int)M; g.S = Sp; g.ln_eps = c.ln_eps; g.x8_sat = h->x8_sat;
    g.tile_both = (cls_as && one_seq_tiles) ? h->w->tile_both : nullptr;  // (whole_pass: no tile is short)
    g.q = h->w->q; g.k = h->w->k; g.vt = h->w->vt;
    const half_t* wqkv = big ? w.wqkv_f : w.wqkv;
    const float* bqkv = big ? w.bqkv_f : w.bqkv;
    if (last && prune) {
      // ---- last layer, [CLS] rows only: K and V of every token, everything else on B rows
      const int Bp = (int)round_up(B, 128);
      g.A = h->w->x16; g.W = wqkv + (size_t)MV_HIDDEN * MV_HIDDEN; g.bias = bqkv + MV_HIDDEN; g.N = 2 * MV_HIDDEN; g.K = MV_HIDDEN;
      g.col0 = MV_HIDDEN;
      if (big) {
        g.lnstats = st_in;
        if (x8) { g.A8 = h->w->x8; g.W8 = w.wqkv_f8 + (size_t)MV_HIDDEN * 2 * MV_HIDDEN; g.x8_scale = w.sc_qkv; g.x8_terms = 3; g.x8_aside_mask = h->qkv_aside_mask; }
        if (special) {  // K and V of the special rows: row term wherever a block sweeps the weight-side term only; V also as hi + lo (the [CLS] query itself is fp32: the tail below)
          if ((h->qkv_aside_mask & 6) != 6) {
            if (int rc = row_term(h->w->st_lo, g.W, g.N, g.K)) return rc;
            g.cls_corr = h->w->cls_corr;
          }
          g.vlo_sp = (h->short_vlo && Sp <= 128) ? nullptr : h->w->vlo_sp;
        }
        if (int rc = launch_pp<PP_QK>(h, KC_GEMM_KV_LAST, g)) return rc;
      } else if (int rc = launch_small<EPI_QKV>(h, KC_GEMM_KV_LAST, g)) return rc;
      ProfScope tail(h, KC_CLS_TAIL);
      const uint32_t keep_mask = h->prof_mask;
      h->prof_mask = 0;  // the tail is one profiled span; its inner launches carry no events of their own
      auto tail_rc = [&]() -> int {
        hipLaunchKernelGGL(cls_gather_kernel, dim3((B + 3) / 4), dim3(256), 0, h->w->stream, h->w->xres, h->w->x16, Sp, B,
                           big ? st_in : (const float*)nullptr, pend_g, pend_b, h->w->c32, h->w->c16, big ? 1 : 0,
                           (big && !x8) ? h->w->xlo : (const half_t*)nullptr, big ? 1 : 0, c.ln_eps, special ? (const half_t*)h->w->st_lo : (const half_t*)nullptr);
        if (int rc = launch_check(h, "cls_gather")) return rc;
        if (x8) {
          // MV_F16X8: the B [CLS] rows in full fp32 on the fp32-input matrix cores (their operand rounding would reach the
          // pooler un-attenuated): Q projection, single-query attention (fp16 K / V^T of the main path, fp32 context), output
          // projection + residual, LayerNorm, FFN, LayerNorm — the fp16 skinny GEMMs below are MV_F16's tail
          const unsigned gx = (unsigned)((B + 31) / 32);
          hipLaunchKernelGGL((dense768_kernel<2, MV_HIDDEN>), dim3(gx, MV_HIDDEN / 32), dim3(512), 0, h->w->stream, (const float*)h->w->c32,
                             (size_t)MV_HIDDEN, B, (const float*)w.wqT32, (const float*)w.bqkv, MV_HIDDEN, h->w->cq, (const float*)nullptr);
          if (int rc = launch_check(h, "cls q")) return rc;
          hipLaunchKernelGGL(attention_cls_kernel, dim3((B * MV_HEADS + 3) / 4), dim3(256), 0, h->w->stream, h->w->cq, h->w->k, h->w->vt,
                             d_lens, h->w->cctx, Sp, B * MV_HEADS, h->w->pooled, (const half_t*)g.vlo_sp);
          if (int rc = launch_check(h, "attention_cls")) return rc;
          hipLaunchKernelGGL((dense768_kernel<4, MV_HIDDEN>), dim3(gx, MV_HIDDEN / 32), dim3(512), 0, h->w->stream, (const float*)h->w->pooled,
                             (size_t)MV_HIDDEN, B, (const float*)w.woT32, (const float*)w.bo, MV_HIDDEN, h->w->c32, (const float*)h->w->c32);
          if (int rc = launch_check(h, "cls out")) return rc;
          if (int rc = run_ln(h->w->c32, h->w->c16, B, w.ln1g, w.ln1b)) return rc;
          hipLaunchKernelGGL((dense768_kernel<3, MV_HIDDEN>), dim3(gx, MV_INTER / 32), dim3(512), 0, h->w->stream, (const float*)h->w->c32,
                             (size_t)MV_HIDDEN, B, (const float*)w.w1T32, (const float*)w.b1, MV_INTER, h->w->ch32, (const float*)nullptr);
          if (int rc = launch_check(h, "cls ffn1")) return rc;
          hipLaunchKernelGGL((dense768_kernel<4, MV_INTER>), dim3(gx, MV_HIDDEN / 32), dim3(512), 0, h->w->stream, (const float*)h->w->ch32,
                             (size_t)MV_INTER, B, (const float*)w.w2T32, (const float*)w.b2, MV_HIDDEN, h->w->c32, (const float*)h->w->c32);
          if (int rc = launch_check(h, "cls ffn2")) return rc;
          if (int rc = run_ln(h->w->c32, h->w->c16, B, w.ln2g, w.ln2b)) return rc;
          return pool_head(h, h->w->c32, MV_HIDDEN, B, u_out);
        }
        GemmArgs t{};
        t.M = Bp; t.Mreal = B; t.S = 64;
        t.A = h->w->c16; t.W = w.wqkv; t.bias = w.bqkv; t.N = MV_HIDDEN; t.K = MV_HIDDEN; t.outf = h->w->cq;
        if (int rc = launch_small<EPI_F32>(h, KC_CLS_TAIL, t)) return rc;
        hipLaunchKernelGGL(attention_cls_kernel, dim3((B * MV_HEADS + 3) / 4), dim3(256), 0, h->w->stream, h->w->cq, h->w->k, h->w->vt,
                           d_lens, h->w->cctx, Sp, B * MV_HEADS);
        if (int rc = launch_check(h, "attention_cls")) return rc;
        t.A = h->w->cctx; t.W = w.wo; t.bias = w.bo; t.N = MV_HIDDEN; t.K = MV_HIDDEN; t.xres = h->w->c32; t.outf = nullptr;
        if (int rc = launch_small<EPI_RES>(h, KC_CLS_TAIL, t)) return rc;
        if (int rc = run_ln(h->w->c32, h->w->c16, B, w.ln1g, w.ln1b)) return rc;
        t.A = h->w->c16; t.W = w.w1; t.bias = w.b1; t.N = MV_INTER; t.K = MV_HIDDEN; t.out16 = h->w->ch16;
        if (int rc = launch_small<EPI_GELU>(h, KC_CLS_TAIL, t)) return rc;
        t.A = h->w->ch16; t.W = w.w2; t.bias = w.b2; t.N = MV_HIDDEN; t.K = MV_INTER; t.xres = h->w->c32;
        if (int rc = launch_small<EPI_RES>(h, KC_CLS_TAIL, t)) return rc;
        if (int rc = run_ln(h->w->c32, h->w->c16, B, w.ln2g, w.ln2b)) return rc;
        return pool_head(h, h->w->c32, MV_HIDDEN, B, u_out);
      }();
      h->prof_mask = keep_mask;
      return tail_rc;
    }
    if (big) {
      // K2: Q, K, V^T projection of the raw stream (LayerNorm folded into W'' / b')
      g.A = h->w->x16; g.W = wqkv; g.bias = bqkv; g.N = 3 * MV_HIDDEN; g.K = MV_HIDDEN; g.lnstats = st_in;
      if (x8) { g.A8 = h->w->x8; g.W8 = w.wqkv_f8; g.x8_scale = w.sc_qkv; g.x8_terms = 3; g.x8_aside_mask = h->qkv_aside_mask; }
      g.vt_lo = (x8 && h->short_vlo && Sp <= 128) ? h->w->vt_lo : nullptr;  // short passes: Q, K, V^T as hi + lo planes (launch_attention: the same predicate)
      g.q_lo = h->w->q_lo; g.k_lo = h->w->k_lo;
      // x8_terms stays 3 — a block of x8_aside_mask (Q by default) keeps its A-side term for EVERY row; the other blocks take it for the special rows from
      // the row term (the launch skips it in blocks that swept both terms: gemm_pp.h).  With diffuse attention K and V of one token are one key among S for
      // every query and the term buys nothing (round 5: model, four draws); with an attention sink on that token they reach every row un-averaged.
      if (special) {
        if (h->qkv_aside_mask != 7) {
          if (int rc = row_term(h->w->st_lo, g.W, g.N, g.K)) return rc;
          g.cls_corr = h->w->cls_corr;
        }
        g.vlo_sp = g.vt_lo ? nullptr : h->w->vlo_sp;
      }
      if (int rc = launch_pp<PP_QK>(h, KC_GEMM_QKV, g)) return rc;
      g.cls_corr = nullptr; g.vlo_sp = nullptr;
      // K3: attention (cls_as: + the context's special rows' low parts for the output projection's row term)
      if (int rc = launch_attention(h, d_lens, B, Sp, x8, cls_as)) return rc;
      // K4: attention output projection + bias + LayerNorm(residual), in place on the raw stream; + vstats of the new rows
      g.A = h->w->ctx; g.W = w.wo; g.bias = w.bo; g.N = MV_HIDDEN; g.K = MV_HIDDEN;
      g.lnstats = st_in; g.lng = pend_g; g.lnb = pend_b; g.lnpart = st_mid; g.out16 = h->w->x16; g.out16b = h->w->xlo;
      if (x8) { g.A8 = h->w->ctx8; g.W8 = w.wo8; g.x8_scale = w.sc_o; g.out8 = h->w->x8; g.x8_terms = 2; }
      if (cls_as) {
        if (int rc = row_term(h->w->cls_lo, g.W, g.N, g.K)) return rc;  // (the context's special low parts: launch_attention)
        g.cls_corr = h->w->cls_corr; g.x8_terms = 1;  // (cls_corr stays set for the rest of the layer: every GEMM's term goes through the same buffer)
        g.out8_hi_only = 0;  // FFN-1 sweeps the weight-side term only (hi8), but the lo8 plane IS the stream's low part: FFN-2 reads it back (gemm.h GemmArgs::out16b)
      }
      if (special) g.sp_lo_out = h->w->st_lo;  // the stream rows' special low parts, read back and rewritten in place: FFN-1's row term, and FFN-2's residual
      if (int rc = launch_pp<PP_RESLN3>(h, KC_GEMM_OUT, g)) return rc;
      pend_g = w.ln1g; pend_b = w.ln1b;
      // K5: FFN-1 + exact-erf GELU
      g.A = h->w->x16; g.W = w.w1_f; g.bias = w.b1_f; g.N = MV_INTER; g.K = MV_HIDDEN; g.lnstats = st_mid; g.out16 = h->w->h16;
      g.out16b = nullptr; g.lnpart = nullptr; g.sp_lo_out = nullptr;
      if (x8) { g.A8 = h->w->x8; g.W8 = w.w1_f8; g.x8_scale = w.sc_1; g.out8 = h->w->h8; g.x8_terms = 2; }
      if (cls_as) {
        if (int rc = row_term(h->w->st_lo, g.W, g.N, g.K)) return rc;
        g.x8_terms = 1;
        g.out8_hi_only = 1;  // h8 is FFN-2's A8: hi8 alone
        g.sp_lo_out = h->w->cls_lo;  // the GELU output's special low parts: FFN-2's row term
      }
      if (int rc = launch_pp<PP_GELU>(h, KC_GEMM_FFN1, g)) return rc;
      // K6: FFN-2 + bias + LayerNorm(residual)
      g.A = h->w->h16; g.W = w.w2; g.bias = w.b2; g.N = MV_HIDDEN; g.K = MV_INTER;
      g.lnstats = st_mid; g.lng = pend_g; g.lnb = pend_b; g.lnpart = st_in; g.out16 = h->w->x16; g.out16b = h->w->xlo;
      if (x8) { g.A8 = h->w->h8; g.W8 = w.w28; g.x8_scale = w.sc_2; g.out8 = h->w->x8; g.x8_terms = 2; }
      if (cls_as) {
        if (int rc = row_term(h->w->cls_lo, g.W, g.N, g.K)) return rc;
        g.x8_terms = 1;
        g.out8_hi_only = 0;  // the lo8 plane is the stream's low part as well as the A-side operand of the next QKV projection's Q block
      } else {
        g.cls_corr = nullptr;
      }
      if (special) g.sp_lo_out = h->w->st_lo;  // the next layer's QKV row term reads the new stream rows' special low parts in every pass of this compute dtype
      if (int rc = launch_pp<PP_RESLN3>(h, KC_GEMM_FFN2, g)) return rc;
      pend_g = w.ln2g; pend_b = w.ln2b;
      if (last) { if (int rc = final_ln(w.ln2g, w.ln2b)) return rc; }  // the pooler reads a normalised stream
    } else {
      g.A = h->w->x16; g.W = wqkv; g.bias = bqkv; g.N = 3 * MV_HIDDEN; g.K = MV_HIDDEN;
      if (int rc = launch_small<EPI_QKV>(h, KC_GEMM_QKV, g)) return rc;
      if (int rc = launch_attention(h, d_lens, B, Sp, false)) return rc;
      g.A = h->w->ctx; g.W = w.wo; g.bias = w.bo; g.N = MV_HIDDEN; g.K = MV_HIDDEN; g.xres = h->w->xres;
      if (int rc = launch_small<EPI_RES>(h, KC_GEMM_OUT, g)) return rc;
      if (int rc = run_ln(h->w->xres, h->w->x16, (int)M, w.ln1g, w.ln1b)) return rc;
      g.A = h->w->x16; g.W = w.w1; g.bias = w.b1; g.N = MV_INTER; g.K = MV_HIDDEN; g.out16 = h->w->h16;
      if (int rc = launch_small<EPI_GELU>(h, KC_GEMM_FFN1, g)) return rc;
      g.A = h->w->h16; g.W = w.w2; g.bias = w.b2; g.N = MV_HIDDEN; g.K = MV_INTER; g.xres = h->w->xres;
      if (int rc = launch_small<EPI_RES>(h, KC_GEMM_FFN2, g)) return rc;
      if (int rc = run_ln(h->w->xres, h->w->x16, (int)M, w.ln2g, w.ln2b)) return rc;
    }
  }
  if (u_out) {
    ProfScope ps(h, KC_POOL_HEAD);
    if (int rc = pool_head(h, h->w->xres, (size_t)Sp * MV_HIDDEN, B, u_out)) return rc;
  }
  return MV_OK;
}

// largest batch one encoder pass can take at padded length Sp
int max_rows_for(mv_handle* h, int S_in) {
  const int Sp = padded_len(S_in);
  int64_t r = (h->cap_tokens - 256) / Sp;
  if (r > h->cfg.max_batch) r = h->cfg.max_batch;
  return (int)r;
}

// K9 + K10 fused (match_topk.h): logits / probs / psame_out are optional full outputs; k >= 1 selects the best anchor
// (and, with topk_p / topk_idx, the k best).  4 issue reports per workgroup when that already fills the chip, else 1
// (the same bits either way).
int match_dev(mv_handle* h, const float* u_dev, int B, float* logits, float* probs, float* psame_out, int k, float* best_out,
              int32_t* idx_out, float* topk_p = nullptr, int32_t* topk_idx = nullptr) {
  const int G = h->n_anchors;
  if (G <= 0) return fail(h, MV_ERR_STATE, "anchor bank is empty (call mv_anchor_append / mv_anchor_set first)");
  MatchArgs a{};
  a.B = B; a.G = G; a.same_idx = h->cfg.same_idx; a.k = k;
  const bool small = G <= 128;                      // one 128-anchor chunk per workgroup (the pass is latency-bound at this size)
  const int GC = small ? 128 : 256;
  a.nchunk = (G + GC - 1) / GC;
  if ((int64_t)a.nchunk * k > 1024) return fail(h, MV_ERR_INVALID, "top-k: anchors / 256 * k must not exceed 1024");
  a.logits = logits; a.probs = probs; a.psame = psame_out;
  a.best = best_out; a.best_idx = idx_out; a.topk_p = topk_p; a.topk_idx = topk_idx;
  a.part_p = h->w->part_p; a.part_q = h->w->part_q; a.part_i = h->w->part_i;
  {
    ProfScope ps(h, KC_MATCH);
    const dim3 grid(small ? 1 : a.nchunk, (B + 3) / 4);
#define MV_MATCH(PD)                                                                                                                          \
    if (small && a.logits) hipLaunchKernelGGL((match_topk_kernel<2, 128, 64, 1, 2, PD>), grid, dim3(256), 0, h->w->stream, u_dev, h->anchors, h->Wm, a); \
    else if (small) hipLaunchKernelGGL((match_topk_kernel<2, 128, 64, 0, 2, PD>), grid, dim3(256), 0, h->w->stream, u_dev, h->anchors, h->Wm, a);        \
    else if (a.logits) hipLaunchKernelGGL((match_topk_kernel<2, 256, 32, 1, 2, PD>), grid, dim3(512), 0, h->w->stream, u_dev, h->anchors, h->Wm, a);     \
    else hipLaunchKernelGGL((match_topk_kernel<2, 256, 32, 0, 2, PD>), grid, dim3(512), 0, h->w->stream, u_dev, h->anchors, h->Wm, a)
    if (h->P == MV_PROJ) { MV_MATCH(MV_PROJ); } else { MV_MATCH(MV_HIDDEN); }
#undef MV_MATCH
    if (int rc = launch_check(h, "match_topk")) return rc;
  }
  if (a.nchunk > 1 && k > 0) {
    ProfScope ps(h, KC_TOPK);
    launch_topk_merge(a, h->w->stream);
    if (int rc = launch_check(h, "topk_merge")) return rc;
  }
  return MV_OK;
}

int sync_all(mv_handle* h) {
  for (int wi = 0; wi < h->n_alloc; ++wi) HIPCHK(h, hipStreamSynchronize(h->work[wi].stream));
  h->dual_pending = false;
  return MV_OK;
}

// Every entry point but the resident sweep works on set 0; a sweep may have left set 1 busy (it reads the anchor
// bank and the resident corpus): wait for it first.
int check_ready(mv_handle* h) {
  if (!h) return MV_ERR_INVALID;
  if (!h->finalized) return fail(h, MV_ERR_STATE, "weights not finalized (mv_finalize_weights)");
  if (h->dual_pending) {
    HIPCHK(h, hipStreamSynchronize(h->work[1].stream));
    h->dual_pending = false;
  }
  h->w = &h->work[0];
  return MV_OK;
}

// HF's embedding lookup raises on an id outside the table; the embedding kernel would clamp silently (a tokenizer /
// checkpoint vocabulary mismatch would then score garbage without a sign): reject such input at the boundary.
int check_ids(mv_handle* h, const int32_t* ids, int64_t n, const char* who) {
  const int32_t V = h->cfg.vocab_size;
  uint32_t bad = 0;
  for (int64_t i = 0; i < n; ++i) bad |= (uint32_t)(ids[i] < 0) | (uint32_t)(ids[i] >= V);
  if (bad) return fail(h, MV_ERR_INVALID, std::string(who) + ": token id outside [0, vocab_size) — tokenizer and checkpoint vocabularies differ?");
  return MV_OK;
}

const HostTensor* find(mv_handle* h, const std::string& k) {
  auto it = h->staged.find(k);
  return it == h->staged.end() ? nullptr : &it->second;
}

int need(mv_handle* h, const std::string& k, std::initializer_list<int64_t> shape, const HostTensor** out) {
  const HostTensor* t = find(h, k);
  if (!t) return fail(h, MV_ERR_MISSING_WEIGHT, "missing weight: " + k);
  std::vector<int64_t> s(shape);
  if (t->shape != s) {
    std::string got;
    for (auto d : t->shape) got += std::to_string(d) + ",";
    return fail(h, MV_ERR_INVALID, "bad shape for " + k + ": got [" + got + "]");
  }
  *out = t;
  return MV_OK;
}

int upload_f32(mv_handle* h, float** dst, const float* src, int64_t n) {
  if (int rc = dev_alloc(h, dst, n, false)) return rc;
  HIPCHK(h, hipMemcpyAsync(*dst, src, (size_t)n * 4, hipMemcpyHostToDevice, h->w->stream));
  HIPCHK(h, hipStreamSynchronize(h->w->stream));
  return MV_OK;
}
// Virtual LayerNorm weights (gemm_pp.h): W''[n][k] = W[n][k] gamma[k] - mean_k(W[n][.] gamma[.]),  b'[n] = b[n] + sum_k W[n][k] beta[k]
void fold_layernorm(const float* W, const float* b, const float* gamma, const float* beta, int64_t N, int64_t K,
                    std::vector<float>& Wf, std::vector<float>& bf) {
  Wf.resize((size_t)(N * K));
  bf.resize((size_t)N);
  for (int64_t n = 0; n < N; ++n) {
    double sum = 0.0, wb = 0.0;
    for (int64_t k = 0; k < K; ++k) {
      const double v = (double)W[n * K + k] * (double)gamma[k];
      sum += v;
      wb += (double)W[n * K + k] * (double)beta[k];
    }
    const double mean = sum / (double)K;
    for (int64_t k = 0; k < K; ++k) Wf[(size_t)(n * K + k)] = (float)((double)W[n * K + k] * (double)gamma[k] - mean);
    bf[(size_t)n] = (float)((double)b[n] + wb);
  }
}

// fp32 -> OCP e4m3fn bits (bias 7, 3 mantissa bits, subnormal step 2^-9, max 448, no infinities), round-to-nearest-even,
// saturating: the host-side twin of v_cvt_pk_fp8_f32 behind a clamp (common.h pack_fp8x4)
inline uint8_t f32_to_e4m3_bits(float f) {
  uint32_t x;
  std::memcpy(&x, &f, 4);
  const uint8_t sign = (uint8_t)((x >> 24) & 0x80u);
  x &= 0x7fffffffu;
  if (x > 0x7f800000u) return (uint8_t)(sign | 0x7fu);  // NaN
  float a;
  std::memcpy(&a, &x, 4);
  if (a >= 448.f) return (uint8_t)(sign | 0x7eu);        // saturate (0x7e = 448)
  if (a < 0.0009765625f) return sign;                    // < 2^-10: rounds to zero (2^-10 itself ties to even = 0)
  int e;
  (void)std::frexp(a, &e);                               // a = m 2^e, m in [0.5, 1)  ->  binade 2^(e-1)
  int be = e - 1;                                        // unbiased exponent
  if (be < -6) be = -6;                                  // subnormal range shares the exponent of the smallest normal
  const float q = std::ldexp(1.0f, be - 3);              // spacing of representable values in this binade
  const float r = std::nearbyint(a / q);                 // default rounding mode: to nearest, ties to even
  int mant = (int)r;                                     // 0..16 (8..16 for normals)
  int exp_field = be + 7;
  if (be == -6 && mant < 8) return (uint8_t)(sign | (uint8_t)mant);  // subnormal (exp field 0)
  if (mant == 16) { mant = 8; exp_field += 1; }
  if (exp_field > 15 || (exp_field == 15 && mant > 14)) return (uint8_t)(sign | 0x7eu);
  return (uint8_t)(sign | (uint8_t)(exp_field << 3) | (uint8_t)(mant - 8));
}

// MV_F16X8 planes of a weight matrix W [N][K] (gemm_pp.h): rows [hi8 | lo8] of 2 K bytes with hi8 = e4m3(fp16(W) 2^sw),
// lo8 = e4m3((W - fp16(W)) 2^(11 + sw)); sw = the largest shift that keeps max |W| inside e4m3's 448.  *scale_word = the E8M0
// byte of 2^-(11 + MV_X8_ACT_SHIFT + sw), replicated (the MFMA's scale operand of this GEMM's correction sweep).
void make_x8_weight_planes(const float* W, int64_t N, int64_t K, std::vector<uint8_t>& out, int* scale_word) {
  float mx = 0.f;
  for (int64_t i = 0; i < N * K; ++i) mx = std::fmax(mx, std::fabs(W[i]));
  int sw = 0;
  if (mx > 0.f) {
    sw = (int)std::floor(std::log2(448.0 / (double)mx));
    if (sw > 24) sw = 24;
    if (sw < -24) sw = -24;
  }
  const float sh = std::ldexp(1.0f, sw), sl = std::ldexp(1.0f, 11 + sw);
  out.resize((size_t)(N * 2 * K));
  for (int64_t n = 0; n < N; ++n) {
    uint8_t* row = out.data() + (size_t)(n * 2 * K);
    for (int64_t k = 0; k < K; ++k) {
      const float w = W[n * K + k], hi = f16_bits_to_f32(f32_to_f16_bits(w));
      row[k] = f32_to_e4m3_bits(hi * sh);
      row[K + k] = f32_to_e4m3_bits((w - hi) * sl);
    }
  }
  const int e8 = 127 - (11 + MV_X8_ACT_SHIFT + sw);
  *scale_word = e8 * 0x01010101;
}

int upload_x8_weight(mv_handle* h, uint8_t** dst, int* scale_word, const float* W, int64_t N, int64_t K) {
  std::vector<uint8_t> tmp;
  make_x8_weight_planes(W, N, K, tmp, scale_word);
  if (int rc = dev_alloc(h, dst, (int64_t)tmp.size(), false)) return rc;
  HIPCHK(h, hipMemcpyAsync(*dst, tmp.data(), tmp.size(), hipMemcpyHostToDevice, h->w->stream));
  HIPCHK(h, hipStreamSynchronize(h->w->stream));
  return MV_OK;
}

int upload_f16(mv_handle* h, half_t** dst, const float* src, int64_t n, float scale = 1.0f) {
  std::vector<uint16_t> tmp((size_t)n);
  for (int64_t i = 0; i < n; ++i) tmp[(size_t)i] = f32_to_f16_bits(src[i] * scale);
  if (int rc = dev_alloc(h, dst, n, false)) return rc;
  HIPCHK(h, hipMemcpyAsync(*dst, tmp.data(), (size_t)n * 2, hipMemcpyHostToDevice, h->w->stream));
  HIPCHK(h, hipStreamSynchronize(h->w->stream));
  return MV_OK;
}

}  // namespace

// =================================================================================================
extern "C" {

const char* mv_last_error(mv_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

const char* mv_kernel_class_name(int cls) {
  return (cls >= 0 && cls < MV_NUM_KERNEL_CLASSES) ? kKernelClassNames[cls] : "";
}

int mv_create(int device, const mv_config* cfg, mv_handle** out) try {
  if (!cfg || !out) return fail(nullptr, MV_ERR_INVALID, "null argument");
  if (cfg->hidden != MV_HIDDEN || cfg->heads != MV_HEADS || cfg->intermediate != MV_INTER ||
      (cfg->proj_dim != MV_PROJ && cfg->proj_dim != MV_HIDDEN))
    return fail(nullptr, MV_ERR_INVALID, "kernels are specialised to hidden=768, heads=12, intermediate=3072, proj_dim=512 (header) "
                                         "or 768 (use_header = False: no header)");
  if (cfg->layers < 0 || cfg->vocab_size <= 0 || cfg->max_pos <= 0 || cfg->max_pos > 512 || cfg->max_tokens <= 0 ||
      cfg->max_batch <= 0 || cfg->max_anchors <= 0 || cfg->type_vocab <= 0 || (cfg->same_idx != 0 && cfg->same_idx != 1))
    return fail(nullptr, MV_ERR_INVALID, "bad mv_config field");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail(nullptr, MV_ERR_HIP, std::string("no HIP device: ") + hipGetErrorString(e));
  if (device < 0 || device >= ndev) return fail(nullptr, MV_ERR_INVALID, "device index out of range");
  e = hipSetDevice(device);
  if (e != hipSuccess) return fail(nullptr, MV_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
  mv_handle* h = new (std::nothrow) mv_handle();
  if (!h) return fail(nullptr, MV_ERR_NOMEM, "out of host memory");
  struct Guard {  // an exception below (caught by this function's handler) must not leak the half-built handle
    mv_handle* h;
    ~Guard() { if (h) mv_destroy(h); }
  } guard{h};
  h->device = device;
  h->cfg = *cfg;
  h->P = cfg->proj_dim;
  // ---- environment switches (include/memvul_hip.h lists them).  Every one is parsed strictly: a value the library does not understand fails
  // mv_create with a message — a typo must never silently select other numerics (or another stream count) than the one asked for.
  auto env_int = [&](const char* name, int lo, int hi, int* out) -> bool {  // false = present and malformed (g_create_error set)
    const char* e = getenv(name);
    if (!e) return true;
    char* end = nullptr;
    const long v = strtol(e, &end, 10);
    if (end == e || *end != '\0' || v < lo || v > hi) {
      g_create_error = std::string(name) + "=\"" + e + "\": expected an integer in " + std::to_string(lo) + " .. " + std::to_string(hi);
      return false;
    }
    *out = (int)v;
    return true;
  };
  auto env_flag = [&](const char* name, bool* out) -> bool {
    int v = *out ? 1 : 0;
    if (!env_int(name, 0, 1, &v)) return false;
    *out = v != 0;
    return true;
  };
  {
    int ns = h->n_streams;
    if (!env_int("MEMVUL_STREAMS", 1, 2, &ns)) return MV_ERR_INVALID;  // (the guard destroys the handle)
    h->n_streams = h->n_alloc = ns;
  }
  for (int wi = 0; wi < h->n_alloc; ++wi) {
    e = hipStreamCreateWithFlags(&h->work[wi].stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      g_create_error = std::string("hipStreamCreate: ") + hipGetErrorString(e);
      return MV_ERR_HIP;  // the guard destroys the handle
    }
  }
  // dynamic LDS above 64 KiB needs an explicit opt-in — per device, so here and not behind a process-wide flag
  hipFuncSetAttribute((const void*)gemm_ring_kernel<EPI_F32, 1, 1, 2, 2, 64, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, RING64_LDS);
  hipFuncSetAttribute((const void*)gemm_ring_kernel<EPI_QKV, 1, 1, 2, 2, 64, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, RING64_LDS);
  hipFuncSetAttribute((const void*)gemm_ring_kernel<EPI_GELU, 1, 1, 2, 2, 64, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, RING64_LDS);
  hipFuncSetAttribute((const void*)gemm_ring_kernel<EPI_RES, 1, 1, 2, 2, 64, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, RING64_LDS);
  hipFuncSetAttribute((const void*)gemm_pp_kernel<PP_QK, 1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES_RAW);
  hipFuncSetAttribute((const void*)gemm_pp_kernel<PP_GELU, 1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES_RAW);
  hipFuncSetAttribute((const void*)gemm_pp_kernel<PP_RESLN3, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES);
  hipFuncSetAttribute((const void*)gemm_pp_kernel<PP_QK, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES_RAW);
  hipFuncSetAttribute((const void*)gemm_pp_kernel<PP_GELU, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES_RAW);
  hipFuncSetAttribute((const void*)gemm_pp_kernel<PP_RESLN3, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES);
#define MV_ATT_ATTR(NKB, NCH)                                                                                                     \
  hipFuncSetAttribute((const void*)attention_v2_kernel<NKB, NCH, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, ATT2_LDS_BYTES(NKB)); \
  hipFuncSetAttribute((const void*)attention_v2_kernel<NKB, NCH, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, ATT2_LDS_BYTES(NKB))
  MV_ATT_ATTR(1, 1); MV_ATT_ATTR(2, 1); MV_ATT_ATTR(3, 1); MV_ATT_ATTR(4, 1); MV_ATT_ATTR(2, 3); MV_ATT_ATTR(2, 4);
#undef MV_ATT_ATTR
  hipFuncSetAttribute((const void*)attention_v2_kernel<1, 1, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, ATT2_LDS_BYTES_VLO(1));
  hipFuncSetAttribute((const void*)attention_v2_kernel<2, 1, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, ATT2_LDS_BYTES_VLO(2));
  (void)hipGetLastError();
  if (!env_flag("MEMVUL_CLS_PRUNE", &h->cls_prune)) return MV_ERR_INVALID;
  if (const char* e = getenv("MEMVUL_QKV_ASIDE")) {
    h->qkv_aside_mask = 0;
    if (strcmp(e, "none")) {
      for (const char* c = e; *c; ++c) {
        const int bit = (*c == 'q' || *c == 'Q') ? 1 : (*c == 'k' || *c == 'K') ? 2 : (*c == 'v' || *c == 'V') ? 4 : 0;
        if (!bit) {
          g_create_error = std::string("MEMVUL_QKV_ASIDE=\"") + e + "\": expected a subset of \"qkv\", \"\" or \"none\"";
          return MV_ERR_INVALID;
        }
        h->qkv_aside_mask |= bit;
      }
    }
  }
  if (!env_flag("MEMVUL_CLS_ASIDE", &h->cls_aside)) return MV_ERR_INVALID;
  if (!env_int("MEMVUL_CLS_ASIDE_MIN_LEN", 1, 512, &h->cls_min_len)) return MV_ERR_INVALID;
  {
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0) h->num_cu = ncu;
  }
#ifdef MEMVUL_DEV_SWITCHES
  // Development A/B knobs: compiled only into libmemvul_hip_dev.so (memvul_amd/build.py dev=True; the GPU tests that force a kernel path at test
  // sizes and the A/B scripts load that build) — the product library does not read them.
  //   MEMVUL_GEMM_TILE  0 by pass size / 128 the small-pass kernels / 512 the persistent kernels forced
  //   MEMVUL_SHORT_VLO  0: passes of padded length <= 128 carry Q, K, V, P as ONE fp16 plane through attention (the A/B of attention_v2.h VLO)
  //   MEMVUL_NUM_CU     size the persistent grids for a share of the chip;  MEMVUL_RASTER 1: the A-stationary raster;  MEMVUL_GN_MAX 1 .. 12: raster group width cap
  {
    int gt = h->gemm_tile;
    if (!env_int("MEMVUL_GEMM_TILE", 0, 512, &gt)) return MV_ERR_INVALID;
    if (gt != 0 && gt != 128 && gt != 512) { g_create_error = "MEMVUL_GEMM_TILE: expected 0, 128 or 512"; return MV_ERR_INVALID; }
    h->gemm_tile = gt;
    if (!env_flag("MEMVUL_SHORT_VLO", &h->short_vlo)) return MV_ERR_INVALID;
    if (!env_int("MEMVUL_GN_MAX", 1, 12, &h->pp_gn_max)) return MV_ERR_INVALID;
    if (!env_int("MEMVUL_RASTER", 0, 1, &h->pp_raster)) return MV_ERR_INVALID;
    if (!env_int("MEMVUL_NUM_CU", 1, h->num_cu, &h->num_cu)) return MV_ERR_INVALID;
  }
#endif

  h->cap_tokens = round_up(cfg->max_tokens, 256) + 256;
  const int64_t T = h->cap_tokens;
  int rc = MV_OK;
  auto A = [&](int r) { if (rc == MV_OK) rc = r; };
  const int64_t BG = (int64_t)cfg->max_batch * cfg->max_anchors;
  const int64_t Bp = round_up(cfg->max_batch, 256);  // [CLS]-row buffers of the pruned last layer
  for (int wi = 0; wi < h->n_alloc; ++wi) {
    h->w = &h->work[wi];
    A(dev_alloc(h, &h->w->d_ids, T));
    A(dev_alloc(h, &h->w->d_lens, (int64_t)cfg->max_batch + 16));
    A(dev_alloc(h, &h->w->xres, T * MV_HIDDEN));
    A(dev_alloc(h, &h->w->x16, T * MV_HIDDEN));
    A(dev_alloc(h, &h->w->q, T * MV_HIDDEN));
    A(dev_alloc(h, &h->w->k, T * MV_HIDDEN));
    A(dev_alloc(h, &h->w->vt, T * MV_HIDDEN));
    A(dev_alloc(h, &h->w->ctx, T * MV_HIDDEN));
    A(dev_alloc(h, &h->w->h16, T * MV_INTER));
    A(dev_alloc(h, &h->w->lnstats, T * 6));
    A(dev_alloc(h, &h->w->lnpart, T * 6));
    A(dev_alloc(h, &h->w->c32, Bp * MV_HIDDEN));
    A(dev_alloc(h, &h->w->cq, Bp * MV_HIDDEN));
    A(dev_alloc(h, &h->w->c16, Bp * MV_HIDDEN));
    A(dev_alloc(h, &h->w->cctx, Bp * MV_HIDDEN));
    A(dev_alloc(h, &h->w->ch16, Bp * MV_INTER));
    A(dev_alloc(h, &h->w->u, (int64_t)cfg->max_batch * h->P));
    A(dev_alloc(h, &h->w->pooled, (int64_t)cfg->max_batch * MV_HIDDEN));
    A(dev_alloc(h, &h->w->u_in, (int64_t)cfg->max_batch * h->P));
    A(dev_alloc(h, &h->w->logits, BG * 2));
    A(dev_alloc(h, &h->w->probs, BG * 2));
    A(dev_alloc(h, &h->w->psame, BG));
    A(dev_alloc(h, &h->w->best, (int64_t)cfg->max_batch * 2));
    A(dev_alloc(h, &h->w->best_idx, cfg->max_batch));
    A(dev_alloc(h, &h->w->topk_p, (int64_t)cfg->max_batch * 64));
    A(dev_alloc(h, &h->w->topk_idx, (int64_t)cfg->max_batch * 64));
    {
      const int64_t nch = (cfg->max_anchors + 255) / 256;
      const int64_t per = nch > 1 ? (nch * MK_KMAX < 1024 ? nch * MK_KMAX : 1024) : 0;  // chunks x k <= 1024 (match_dev)
      A(dev_alloc(h, &h->w->part_p, (int64_t)cfg->max_batch * per));
      A(dev_alloc(h, &h->w->part_q, (int64_t)cfg->max_batch * per));
      A(dev_alloc(h, &h->w->part_i, (int64_t)cfg->max_batch * per));
    }
    if (rc == MV_OK && hipStreamSynchronize(h->w->stream) != hipSuccess) rc = MV_ERR_HIP;
  }
  h->w = &h->work[0];
  A(dev_alloc(h, &h->anchors, (int64_t)cfg->max_anchors * h->P));
  A(dev_alloc(h, &h->x8_sat, 1));  // (zeroed by dev_alloc)
  A(dev_alloc(h, &h->attn_conc, 4));
  if (rc == MV_OK && hipStreamSynchronize(h->w->stream) != hipSuccess) rc = MV_ERR_HIP;
  if (rc != MV_OK) {
    g_create_error = h->err.empty() ? "workspace allocation failed" : h->err;
    return rc;  // the guard destroys the handle
  }
  guard.h = nullptr;
  *out = h;
  return MV_OK;
} catch (...) { return on_exception(nullptr); }

void mv_destroy(mv_handle* h) {
  if (!h) return;
  hipSetDevice(h->device);
  mv_comm_destroy(h);
  for (auto& wk : h->work)
    if (wk.stream) hipStreamSynchronize(wk.stream);
  for (auto& r : h->recs) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
  for (auto e : h->free_events) hipEventDestroy(e);
  for (void* p : h->allocs) hipFree(p);
  for (void* p : h->pinned) hipHostFree(p);
  for (auto& wk : h->work)
    if (wk.stream) hipStreamDestroy(wk.stream);
  delete h;
}

int mv_sync(mv_handle* h) try {
  if (!h) return MV_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  return sync_all(h);
} catch (...) { return on_exception(h); }

int mv_load_tensor(mv_handle* h, const char* name, const void* host_ptr, int dtype, const int64_t* shape, int ndim) try {
  if (!h || !name || !host_ptr || !shape || ndim < 1 || ndim > 4) return fail(h, MV_ERR_INVALID, "mv_load_tensor: bad argument");
  if (h->finalized) return fail(h, MV_ERR_STATE, "weights already finalized");
  int64_t n = 1;
  for (int i = 0; i < ndim; ++i) {
    if (shape[i] < 0) return fail(h, MV_ERR_INVALID, "negative dimension");
    n *= shape[i];
  }
  HostTensor t;
  t.shape.assign(shape, shape + ndim);
  t.data.resize((size_t)n);
  if (dtype == MV_F32) std::memcpy(t.data.data(), host_ptr, (size_t)n * 4);
  else if (dtype == MV_F16) { const uint16_t* s = (const uint16_t*)host_ptr; for (int64_t i = 0; i < n; ++i) t.data[(size_t)i] = f16_bits_to_f32(s[i]); }
  else if (dtype == MV_BF16) { const uint16_t* s = (const uint16_t*)host_ptr; for (int64_t i = 0; i < n; ++i) t.data[(size_t)i] = bf16_bits_to_f32(s[i]); }
  else if (dtype == MV_I64 || dtype == MV_I32) return MV_OK;  // e.g. embeddings.position_ids: accepted, unused
  else return fail(h, MV_ERR_INVALID, "mv_load_tensor: unsupported dtype");
  h->staged[name] = std::move(t);
  return MV_OK;
} catch (...) { return on_exception(h); }

int mv_finalize_weights(mv_handle* h, int compute_dtype) try {
  if (!h) return MV_ERR_INVALID;
  if (h->finalized) return fail(h, MV_ERR_STATE, "weights already finalized");
  if (compute_dtype != MV_F16 && compute_dtype != MV_F16X8)
    return fail(h, MV_ERR_INVALID, "compute_dtype must be MV_F16 (fp16 MFMA operands, fp32 accumulation) or MV_F16X8 (+ fp8 correction "
                                   "sweeps); bf16 is a storage dtype of mv_load_tensor only (include/memvul_hip.h)");
  const bool precise = compute_dtype == MV_F16X8;
  HIPCHK(h, hipSetDevice(h->device));
  const mv_config& c = h->cfg;
  const std::string P = "_text_field_embedder.token_embedder_tokens.transformer_model.";
  const int64_t H = MV_HIDDEN, I = MV_INTER;
  const HostTensor *t = nullptr, *t2 = nullptr, *t3 = nullptr;
  int rc;
#define NEED(key, ...) if ((rc = need(h, key, {__VA_ARGS__}, &t)) != MV_OK) return rc
  NEED(P + "embeddings.word_embeddings.weight", c.vocab_size, H);
  if ((rc = upload_f32(h, &h->wemb, t->data.data(), (int64_t)c.vocab_size * H))) return rc;
  {
    const HostTensor* tp = find(h, P + "embeddings.position_embeddings.weight");
    if (!tp) return fail(h, MV_ERR_MISSING_WEIGHT, "missing weight: " + P + "embeddings.position_embeddings.weight");
    if (tp->shape.size() != 2 || tp->shape[1] != H || tp->shape[0] < c.max_pos)
      return fail(h, MV_ERR_INVALID, "bad shape for position_embeddings");
    if ((rc = upload_f32(h, &h->pemb, tp->data.data(), (int64_t)c.max_pos * H))) return rc;
  }
  NEED(P + "embeddings.token_type_embeddings.weight", c.type_vocab, H);
  if ((rc = upload_f32(h, &h->temb, t->data.data(), H))) return rc;  // row 0 only: type ids are all zero on this path
  NEED(P + "embeddings.LayerNorm.weight", H);
  if ((rc = upload_f32(h, &h->embg, t->data.data(), H))) return rc;
  NEED(P + "embeddings.LayerNorm.bias", H);
  if ((rc = upload_f32(h, &h->embb, t->data.data(), H))) return rc;
  h->L.resize(c.layers);
  for (int l = 0; l < c.layers; ++l) {
    const std::string q = P + "encoder.layer." + std::to_string(l) + ".";
    LayerW& w = h->L[l];
    std::vector<float> wqkv_host, bqkv_host;
    // packed QKV [2304][768]; 1/sqrt(64) folded into W_q, b_q (exact: power of two)
    if ((rc = need(h, q + "attention.self.query.weight", {H, H}, &t))) return rc;
    if ((rc = need(h, q + "attention.self.key.weight", {H, H}, &t2))) return rc;
    if ((rc = need(h, q + "attention.self.value.weight", {H, H}, &t3))) return rc;
    {
      std::vector<float> pack((size_t)(3 * H * H));
      for (int64_t i = 0; i < H * H; ++i) {
        pack[(size_t)i] = t->data[(size_t)i] * 0.125f;
        pack[(size_t)(H * H + i)] = t2->data[(size_t)i];
        pack[(size_t)(2 * H * H + i)] = t3->data[(size_t)i];
      }
      if ((rc = upload_f16(h, &w.wqkv, pack.data(), 3 * H * H))) return rc;
      wqkv_host = pack;
    }
    if ((rc = need(h, q + "attention.self.query.bias", {H}, &t))) return rc;
    if ((rc = need(h, q + "attention.self.key.bias", {H}, &t2))) return rc;
    if ((rc = need(h, q + "attention.self.value.bias", {H}, &t3))) return rc;
    {
      std::vector<float> pack((size_t)(3 * H));
      for (int64_t i = 0; i < H; ++i) {
        pack[(size_t)i] = t->data[(size_t)i] * 0.125f;
        pack[(size_t)(H + i)] = t2->data[(size_t)i];
        pack[(size_t)(2 * H + i)] = t3->data[(size_t)i];
      }
      if ((rc = upload_f32(h, &w.bqkv, pack.data(), 3 * H))) return rc;
      bqkv_host = pack;
    }
    {  // the LayerNorm in front of this layer's QKV projection: the embedding LayerNorm or the previous layer's output LayerNorm
      const std::string lnk = l == 0 ? P + "embeddings.LayerNorm." : P + "encoder.layer." + std::to_string(l - 1) + ".output.LayerNorm.";
      const HostTensor *tg = nullptr, *tb = nullptr;
      if ((rc = need(h, lnk + "weight", {H}, &tg))) return rc;
      if ((rc = need(h, lnk + "bias", {H}, &tb))) return rc;
      std::vector<float> Wf, bf;
      fold_layernorm(wqkv_host.data(), bqkv_host.data(), tg->data.data(), tb->data.data(), 3 * H, H, Wf, bf);
      if ((rc = upload_f16(h, &w.wqkv_f, Wf.data(), 3 * H * H))) return rc;
      if (precise && (rc = upload_x8_weight(h, &w.wqkv_f8, &w.sc_qkv, Wf.data(), 3 * H, H))) return rc;
      if ((rc = upload_f32(h, &w.bqkv_f, bf.data(), 3 * H))) return rc;
    }
    NEED(q + "attention.output.dense.weight", H, H);
    if ((rc = upload_f16(h, &w.wo, t->data.data(), H * H))) return rc;
    if (precise && (rc = upload_x8_weight(h, &w.wo8, &w.sc_o, t->data.data(), H, H))) return rc;
    NEED(q + "attention.output.dense.bias", H);
    if ((rc = upload_f32(h, &w.bo, t->data.data(), H))) return rc;
    NEED(q + "attention.output.LayerNorm.weight", H);
    if ((rc = upload_f32(h, &w.ln1g, t->data.data(), H))) return rc;
    NEED(q + "attention.output.LayerNorm.bias", H);
    if ((rc = upload_f32(h, &w.ln1b, t->data.data(), H))) return rc;
    NEED(q + "intermediate.dense.weight", I, H);
    if ((rc = upload_f16(h, &w.w1, t->data.data(), I * H))) return rc;
    NEED(q + "intermediate.dense.bias", I);
    if ((rc = upload_f32(h, &w.b1, t->data.data(), I))) return rc;
    {  // FFN-1 with the attention-output LayerNorm folded in
      const HostTensor *tw = nullptr, *tg = nullptr, *tb = nullptr;
      if ((rc = need(h, q + "intermediate.dense.weight", {I, H}, &tw))) return rc;
      if ((rc = need(h, q + "attention.output.LayerNorm.weight", {H}, &tg))) return rc;
      if ((rc = need(h, q + "attention.output.LayerNorm.bias", {H}, &tb))) return rc;
      std::vector<float> Wf, bf;
      fold_layernorm(tw->data.data(), t->data.data(), tg->data.data(), tb->data.data(), I, H, Wf, bf);
      if ((rc = upload_f16(h, &w.w1_f, Wf.data(), I * H))) return rc;
      if (precise && (rc = upload_x8_weight(h, &w.w1_f8, &w.sc_1, Wf.data(), I, H))) return rc;
      if ((rc = upload_f32(h, &w.b1_f, bf.data(), I))) return rc;
    }
    NEED(q + "output.dense.weight", H, I);
    if ((rc = upload_f16(h, &w.w2, t->data.data(), H * I))) return rc;
    if (precise && (rc = upload_x8_weight(h, &w.w28, &w.sc_2, t->data.data(), H, I))) return rc;
    NEED(q + "output.dense.bias", H);
    if ((rc = upload_f32(h, &w.b2, t->data.data(), H))) return rc;
    NEED(q + "output.LayerNorm.weight", H);
    if ((rc = upload_f32(h, &w.ln2g, t->data.data(), H))) return rc;
    NEED(q + "output.LayerNorm.bias", H);
    if ((rc = upload_f32(h, &w.ln2b, t->data.data(), H))) return rc;
  }
  if (precise && c.layers > 0) {  // fp32 [CLS] tail of the last layer: weights transposed to [k][n]
    const std::string q = P + "encoder.layer." + std::to_string(c.layers - 1) + ".";
    LayerW& w = h->L[c.layers - 1];
    auto up_T = [&](const std::string& key, int64_t N, int64_t K, float scale, float** dst) -> int {
      const HostTensor* tt = nullptr;
      if (int r = need(h, key, {N, K}, &tt)) return r;
      std::vector<float> tr((size_t)(N * K));
      for (int64_t n = 0; n < N; ++n) for (int64_t k = 0; k < K; ++k) tr[(size_t)(k * N + n)] = tt->data[(size_t)(n * K + k)] * scale;
      return upload_f32(h, dst, tr.data(), N * K);
    };
    if ((rc = up_T(q + "attention.self.query.weight", H, H, 0.125f, &w.wqT32))) return rc;  // 1/sqrt(64) folded like the packed QKV
    if ((rc = up_T(q + "attention.output.dense.weight", H, H, 1.0f, &w.woT32))) return rc;
    if ((rc = up_T(q + "intermediate.dense.weight", I, H, 1.0f, &w.w1T32))) return rc;
    if ((rc = up_T(q + "output.dense.weight", H, I, 1.0f, &w.w2T32))) return rc;
  }
  // pooler / header: transposed to [k][n] (fp32)
  NEED("_bert_pooler.pooler.dense.weight", H, H);
  {
    std::vector<float> tr((size_t)(H * H));
    for (int64_t n = 0; n < H; ++n) for (int64_t k = 0; k < H; ++k) tr[(size_t)(k * H + n)] = t->data[(size_t)(n * H + k)];
    if ((rc = upload_f32(h, &h->WpT, tr.data(), H * H))) return rc;
  }
  NEED("_bert_pooler.pooler.dense.bias", H);
  if ((rc = upload_f32(h, &h->bp, t->data.data(), H))) return rc;
  if (h->P == MV_PROJ) {  // use_header (model_memory.py:69-71); with proj_dim = 768 the model has no _projector_single
    NEED("_projector_single._linear_layers.0.weight", MV_PROJ, H);
    {
      std::vector<float> tr((size_t)(H * MV_PROJ));
      for (int64_t n = 0; n < MV_PROJ; ++n) for (int64_t k = 0; k < H; ++k) tr[(size_t)(k * MV_PROJ + n)] = t->data[(size_t)(n * H + k)];
      if ((rc = upload_f32(h, &h->WhT, tr.data(), H * MV_PROJ))) return rc;
    }
    NEED("_projector_single._linear_layers.0.bias", MV_PROJ);
    if ((rc = upload_f32(h, &h->bh, t->data.data(), MV_PROJ))) return rc;
  }
  NEED("_projector.weight", 2, 3 * (int64_t)h->P);
  if ((rc = upload_f32(h, &h->Wm, t->data.data(), 2 * 3 * (int64_t)h->P))) return rc;
#undef NEED
  if (!precise) {  // MV_F16: the lo fp16 plane of the two-plane raw stream (MV_F16X8 keeps the stream's low part in the lo8 plane of x8 + st_lo: gemm.h GemmArgs::out16b)
    for (int wi = 0; wi < h->n_alloc; ++wi) {
      Work* keep = h->w;
      h->w = &h->work[wi];
      rc = dev_alloc(h, &h->work[wi].xlo, h->cap_tokens * MV_HIDDEN);
      if (rc == MV_OK && hipStreamSynchronize(h->w->stream) != hipSuccess) rc = MV_ERR_HIP;
      h->w = keep;
      if (rc != MV_OK) return rc;
    }
  }
  if (precise) {  // fp8 planes [lo8 | hi8] of the three activations that are GEMM A operands
    if (h->gemm_tile == 128) return fail(h, MV_ERR_STATE, "MV_F16X8 runs on the persistent GEMM path: MEMVUL_GEMM_TILE=128 excludes it");
    for (int wi = 0; wi < h->n_alloc; ++wi) {
      Work* keep = h->w;
      h->w = &h->work[wi];
      rc = dev_alloc(h, &h->work[wi].x8, h->cap_tokens * 2 * MV_HIDDEN);
      if (rc == MV_OK) rc = dev_alloc(h, &h->work[wi].ctx8, h->cap_tokens * 2 * MV_HIDDEN);
      if (rc == MV_OK) rc = dev_alloc(h, &h->work[wi].h8, h->cap_tokens * 2 * MV_INTER);
      if (rc == MV_OK) rc = dev_alloc(h, &h->work[wi].ch32, (int64_t)round_up(h->cfg.max_batch, 256) * MV_INTER);
      if (rc == MV_OK) rc = dev_alloc(h, &h->work[wi].cls_lo, 2 * (int64_t)round_up(h->cfg.max_batch, 256) * MV_INTER);
      if (rc == MV_OK) rc = dev_alloc(h, &h->work[wi].cls_corr, 2 * (int64_t)round_up(h->cfg.max_batch, 256) * MV_INTER);
      if (rc == MV_OK) rc = dev_alloc(h, &h->work[wi].st_lo, 2 * (int64_t)round_up(h->cfg.max_batch, 256) * MV_HIDDEN);
      if (rc == MV_OK) rc = dev_alloc(h, &h->work[wi].vlo_sp, (int64_t)h->cfg.max_batch * MV_HEADS * MV_HEAD_DIM * 2);
      if (rc == MV_OK) rc = dev_alloc(h, &h->work[wi].tile_both, h->cap_tokens / 256 + 1);
      if (h->short_vlo) {  // second fp16 planes of V^T, Q, K: read only by passes of padded length <= 128 in this compute dtype (attention_v2.h VLO)
        if (rc == MV_OK) rc = dev_alloc(h, &h->work[wi].vt_lo, h->cap_tokens * MV_HIDDEN);
        if (rc == MV_OK) rc = dev_alloc(h, &h->work[wi].q_lo, h->cap_tokens * MV_HIDDEN);
        if (rc == MV_OK) rc = dev_alloc(h, &h->work[wi].k_lo, h->cap_tokens * MV_HIDDEN);
      }
      if (rc == MV_OK && hipStreamSynchronize(h->w->stream) != hipSuccess) rc = MV_ERR_HIP;
      h->w = keep;
      if (rc != MV_OK) return rc;
    }
  }
  h->precise = precise;
  h->staged.clear();
  h->compute_dtype = compute_dtype;
  h->finalized = true;
  return MV_OK;
} catch (...) { return on_exception(h); }

int mv_anchor_reset(mv_handle* h) try {
  if (!h) return MV_ERR_INVALID;
  h->n_anchors = 0;
  return MV_OK;
} catch (...) { return on_exception(h); }
int mv_anchor_count(mv_handle* h) { return h ? h->n_anchors : MV_ERR_INVALID; }

int mv_anchor_append(mv_handle* h, const int32_t* ids, const int32_t* lens, int n, int S) try {
  if (int rc = check_ready(h)) return rc;
  if (!ids || !lens || n <= 0 || S <= 0 || S > h->cfg.max_pos) return fail(h, MV_ERR_INVALID, "mv_anchor_append: bad argument");
  if (h->n_anchors + n > h->cfg.max_anchors) return fail(h, MV_ERR_CAPACITY, "anchor bank capacity (mv_config.max_anchors) exceeded");
  if (int rc = check_ids(h, ids, (int64_t)n * S, "mv_anchor_append")) return rc;
  HIPCHK(h, hipSetDevice(h->device));
  const int rows = max_rows_for(h, S);
  if (rows <= 0) return fail(h, MV_ERR_CAPACITY, "mv_config.max_tokens too small for one anchor of this length");
  for (int off = 0; off < n; off += rows) {
    const int nb = (n - off < rows) ? (n - off) : rows;
    HIPCHK(h, hipMemcpyAsync(h->w->d_ids, ids + (size_t)off * S, (size_t)nb * S * 4, hipMemcpyHostToDevice, h->w->stream));
    HIPCHK(h, hipMemcpyAsync(h->w->d_lens, lens + off, (size_t)nb * 4, hipMemcpyHostToDevice, h->w->stream));
    if (int rc = encode_dev(h, h->w->d_ids, h->w->d_lens, pass_min_len(lens + off, nb), nb, S, -1, h->anchors + (size_t)(h->n_anchors + off) * h->P)) return rc;
    HIPCHK(h, hipStreamSynchronize(h->w->stream));
  }
  h->n_anchors += n;
  return MV_OK;
} catch (...) { return on_exception(h); }

int mv_anchor_get(mv_handle* h, float* out) try {
  if (!h || !out) return MV_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  if (int rc = sync_all(h)) return rc;  // a sweep may still be appending / reading on the other stream
  h->w = &h->work[0];
  HIPCHK(h, hipMemcpyAsync(out, h->anchors, (size_t)h->n_anchors * h->P * 4, hipMemcpyDeviceToHost, h->w->stream));
  HIPCHK(h, hipStreamSynchronize(h->w->stream));
  return MV_OK;
} catch (...) { return on_exception(h); }

int mv_anchor_set(mv_handle* h, const float* v, int G) try {
  if (!h || !v || G <= 0) return fail(h, MV_ERR_INVALID, "mv_anchor_set: bad argument");
  if (G > h->cfg.max_anchors) return fail(h, MV_ERR_CAPACITY, "anchor bank capacity (mv_config.max_anchors) exceeded");
  HIPCHK(h, hipSetDevice(h->device));
  if (int rc = sync_all(h)) return rc;  // batches of a resident sweep in flight read the bank
  h->w = &h->work[0];
  HIPCHK(h, hipMemcpyAsync(h->anchors, v, (size_t)G * h->P * 4, hipMemcpyHostToDevice, h->w->stream));
  HIPCHK(h, hipStreamSynchronize(h->w->stream));
  h->n_anchors = G;
  return MV_OK;
} catch (...) { return on_exception(h); }

int mv_encode(mv_handle* h, const int32_t* ids, const int32_t* lens, int B, int S, float* embed) try {
  if (int rc = check_ready(h)) return rc;
  if (!ids || !lens || B <= 0 || S <= 0 || S > h->cfg.max_pos) return fail(h, MV_ERR_INVALID, "mv_encode: bad argument");
  if (int rc = check_ids(h, ids, (int64_t)B * S, "mv_encode")) return rc;
  HIPCHK(h, hipSetDevice(h->device));
  const int rows = max_rows_for(h, S);
  if (rows <= 0) return fail(h, MV_ERR_CAPACITY, "mv_config.max_tokens too small for this sequence length");
  for (int off = 0; off < B; off += rows) {
    const int nb = (B - off < rows) ? (B - off) : rows;
    HIPCHK(h, hipMemcpyAsync(h->w->d_ids, ids + (size_t)off * S, (size_t)nb * S * 4, hipMemcpyHostToDevice, h->w->stream));
    HIPCHK(h, hipMemcpyAsync(h->w->d_lens, lens + off, (size_t)nb * 4, hipMemcpyHostToDevice, h->w->stream));
    if (int rc = encode_dev(h, h->w->d_ids, h->w->d_lens, pass_min_len(lens + off, nb), nb, S, -1, h->w->u)) return rc;
    if (embed) HIPCHK(h, hipMemcpyAsync(embed + (size_t)off * h->P, h->w->u, (size_t)nb * h->P * 4, hipMemcpyDeviceToHost, h->w->stream));
    HIPCHK(h, hipStreamSynchronize(h->w->stream));
  }
  return MV_OK;
} catch (...) { return on_exception(h); }

int mv_forward(mv_handle* h, const int32_t* ids, const int32_t* lens, int B, int S, float* logits, float* probs, float* best,
               int32_t* best_idx, float* embed) try {
  if (int rc = check_ready(h)) return rc;
  if (!ids || !lens || B <= 0 || S <= 0 || S > h->cfg.max_pos) return fail(h, MV_ERR_INVALID, "mv_forward: bad argument");
  if (h->n_anchors <= 0) return fail(h, MV_ERR_STATE, "anchor bank is empty (call mv_anchor_append / mv_anchor_set first)");
  if (int rc = check_ids(h, ids, (int64_t)B * S, "mv_forward")) return rc;
  HIPCHK(h, hipSetDevice(h->device));
  const int rows = max_rows_for(h, S);
  if (rows <= 0) return fail(h, MV_ERR_CAPACITY, "mv_config.max_tokens too small for this sequence length");
  const int G = h->n_anchors;
  for (int off = 0; off < B; off += rows) {
    const int nb = (B - off < rows) ? (B - off) : rows;
    HIPCHK(h, hipMemcpyAsync(h->w->d_ids, ids + (size_t)off * S, (size_t)nb * S * 4, hipMemcpyHostToDevice, h->w->stream));
    HIPCHK(h, hipMemcpyAsync(h->w->d_lens, lens + off, (size_t)nb * 4, hipMemcpyHostToDevice, h->w->stream));
    if (int rc = encode_dev(h, h->w->d_ids, h->w->d_lens, pass_min_len(lens + off, nb), nb, S, -1, h->w->u)) return rc;
    // only the outputs the caller asked for leave the kernel (the best anchor always does)
    if (int rc = match_dev(h, h->w->u, nb, logits ? h->w->logits : nullptr, probs ? h->w->probs : nullptr, nullptr, 1, h->w->best,
                           h->w->best_idx)) return rc;
    const size_t bg = (size_t)nb * G;
    if (logits) HIPCHK(h, hipMemcpyAsync(logits + (size_t)off * G * 2, h->w->logits, bg * 8, hipMemcpyDeviceToHost, h->w->stream));
    if (probs) HIPCHK(h, hipMemcpyAsync(probs + (size_t)off * G * 2, h->w->probs, bg * 8, hipMemcpyDeviceToHost, h->w->stream));
    if (best) HIPCHK(h, hipMemcpyAsync(best + (size_t)off * 2, h->w->best, (size_t)nb * 8, hipMemcpyDeviceToHost, h->w->stream));
    if (best_idx) HIPCHK(h, hipMemcpyAsync(best_idx + off, h->w->best_idx, (size_t)nb * 4, hipMemcpyDeviceToHost, h->w->stream));
    if (embed) HIPCHK(h, hipMemcpyAsync(embed + (size_t)off * h->P, h->w->u, (size_t)nb * h->P * 4, hipMemcpyDeviceToHost, h->w->stream));
    HIPCHK(h, hipStreamSynchronize(h->w->stream));
  }
  return MV_OK;
} catch (...) { return on_exception(h); }

// mv_forward over a batch whose rows are ordered by length and cut into groups: group g = rows [group_end[g - 1], group_end[g]) runs as ONE pass at
// group_width[g] tokens per row (the rows' ids are read in place from the [B][S] upload: pitch S), all groups back to back on the handle's stream, one
// synchronisation at the end (binding.Engine.forward_by_length: a pad-to-longest batch of unsorted reports without its padding, one library call per batch).
int mv_forward_groups(mv_handle* h, const int32_t* ids, const int32_t* lens, int B, int S, int n_groups, const int32_t* group_end,
                      const int32_t* group_width, float* logits, float* probs, float* best, int32_t* best_idx, float* embed) try {
  if (int rc = check_ready(h)) return rc;
  if (!ids || !lens || !group_end || !group_width || B <= 0 || S <= 0 || S > h->cfg.max_pos || n_groups <= 0)
    return fail(h, MV_ERR_INVALID, "mv_forward_groups: bad argument");
  if (h->n_anchors <= 0) return fail(h, MV_ERR_STATE, "anchor bank is empty (call mv_anchor_append / mv_anchor_set first)");
  if (B > h->cfg.max_batch || (int64_t)B * S > h->cap_tokens) return fail(h, MV_ERR_CAPACITY, "mv_forward_groups: the batch exceeds mv_config.max_batch / max_tokens");
  int prev = 0;
  for (int g = 0; g < n_groups; ++g) {
    const int nb = group_end[g] - prev, w = group_width[g];
    if (nb <= 0 || w <= 0 || w > S) return fail(h, MV_ERR_INVALID, "mv_forward_groups: groups must be non-empty, in order, at most S tokens wide");
    if (nb > max_rows_for(h, w)) return fail(h, MV_ERR_CAPACITY, "mv_forward_groups: a group exceeds one pass (mv_config.max_tokens)");
    for (int i = prev; i < group_end[g]; ++i)
      if (lens[i] > w) return fail(h, MV_ERR_INVALID, "mv_forward_groups: a row is longer than its group's width");
    prev = group_end[g];
  }
  if (prev != B) return fail(h, MV_ERR_INVALID, "mv_forward_groups: the groups must cover the batch");
  if (int rc = check_ids(h, ids, (int64_t)B * S, "mv_forward_groups")) return rc;
  HIPCHK(h, hipSetDevice(h->device));
  const int G = h->n_anchors;
  HIPCHK(h, hipMemcpyAsync(h->w->d_ids, ids, (size_t)B * S * 4, hipMemcpyHostToDevice, h->w->stream));
  HIPCHK(h, hipMemcpyAsync(h->w->d_lens, lens, (size_t)B * 4, hipMemcpyHostToDevice, h->w->stream));
  prev = 0;
  for (int g = 0; g < n_groups; ++g) {
    const int nb = group_end[g] - prev;
    float* u = h->w->u + (size_t)prev * h->P;
    if (int rc = encode_dev(h, h->w->d_ids + (size_t)prev * S, h->w->d_lens + prev, pass_min_len(lens + prev, nb), nb, group_width[g], -1, u, false, S)) return rc;
    if (int rc = match_dev(h, u, nb, logits ? h->w->logits + (size_t)prev * G * 2 : nullptr, probs ? h->w->probs + (size_t)prev * G * 2 : nullptr, nullptr, 1,
                           h->w->best + (size_t)prev * 2, h->w->best_idx + prev)) return rc;
    prev = group_end[g];
  }
  const size_t bg = (size_t)B * G;
  if (logits) HIPCHK(h, hipMemcpyAsync(logits, h->w->logits, bg * 8, hipMemcpyDeviceToHost, h->w->stream));
  if (probs) HIPCHK(h, hipMemcpyAsync(probs, h->w->probs, bg * 8, hipMemcpyDeviceToHost, h->w->stream));
  if (best) HIPCHK(h, hipMemcpyAsync(best, h->w->best, (size_t)B * 8, hipMemcpyDeviceToHost, h->w->stream));
  if (best_idx) HIPCHK(h, hipMemcpyAsync(best_idx, h->w->best_idx, (size_t)B * 4, hipMemcpyDeviceToHost, h->w->stream));
  if (embed) HIPCHK(h, hipMemcpyAsync(embed, h->w->u, (size_t)B * h->P * 4, hipMemcpyDeviceToHost, h->w->stream));
  HIPCHK(h, hipStreamSynchronize(h->w->stream));
  return MV_OK;
} catch (...) { return on_exception(h); }

// mv_forward on a pad-to-longest batch of UNSORTED rows, all of binding.Engine.forward_by_length inside ONE call: the rows ordered (stably) by the padded length of
// their own token count, cut into groups (a group of fewer than min_tokens padded tokens travels with the next longer one), gathered on the host, scored by
// mv_forward_groups, and the results put back in the caller's row order.  One call = one release of the caller's interpreter lock per batch: next to two other
// Python threads every release cost the scoring thread ~10 ms of waiting (profiles/r06_*_e2e_dropin.txt).
int mv_forward_ragged(mv_handle* h, const int32_t* ids, const int32_t* lens, int B, int S, int min_tokens, float* logits, float* probs, float* best,
                      int32_t* best_idx, float* embed) try {
  if (int rc = check_ready(h)) return rc;
  if (!ids || !lens || B <= 0 || S <= 0 || S > h->cfg.max_pos || !best || !best_idx) return fail(h, MV_ERR_INVALID, "mv_forward_ragged: bad argument");
  if (B > h->cfg.max_batch || (int64_t)B * S > h->cap_tokens) return fail(h, MV_ERR_CAPACITY, "mv_forward_ragged: the batch exceeds mv_config.max_batch / max_tokens");
  const int G = h->n_anchors;
  std::vector<int> pl(B), order(B);
  for (int i = 0; i < B; ++i) {
    if (lens[i] > S) return fail(h, MV_ERR_INVALID, "mv_forward_ragged: a row is longer than S");
    pl[i] = padded_len(lens[i] < 1 ? 1 : lens[i]);
    order[i] = i;
  }
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return pl[a] < pl[b]; });
  std::vector<int32_t> ends, widths;
  int start = 0;
  for (int end = 1; end <= B; ++end) {
    if (end < B && pl[order[end]] == pl[order[end - 1]]) continue;  // inside a run of one padded length
    const int width = pl[order[end - 1]];
    if (end < B && (int64_t)(end - start) * width < min_tokens) continue;  // too small a pass: these rows travel with the next longer group
    ends.push_back(end);
    widths.push_back(width < S ? width : S);
    start = end;
  }
  auto& st = h->ragged;
  st.ids.resize((size_t)B * S);
  st.lens.resize(B);
  for (int i = 0; i < B; ++i) {
    std::memcpy(st.ids.data() + (size_t)i * S, ids + (size_t)order[i] * S, (size_t)S * 4);
    st.lens[i] = lens[order[i]];
  }
  if (logits) st.logits.resize((size_t)B * G * 2);
  if (probs) st.probs.resize((size_t)B * G * 2);
  st.best.resize((size_t)B * 2);
  st.idx.resize(B);
  if (embed) st.embed.resize((size_t)B * h->P);
  if (int rc = mv_forward_groups(h, st.ids.data(), st.lens.data(), B, S, (int)ends.size(), ends.data(), widths.data(), logits ? st.logits.data() : nullptr,
                                 probs ? st.probs.data() : nullptr, st.best.data(), st.idx.data(), embed ? st.embed.data() : nullptr)) return rc;
  const size_t g2 = (size_t)G * 2;
  for (int i = 0; i < B; ++i) {
    const size_t o = (size_t)order[i];
    if (logits) std::memcpy(logits + o * g2, st.logits.data() + (size_t)i * g2, g2 * 4);
    if (probs) std::memcpy(probs + o * g2, st.probs.data() + (size_t)i * g2, g2 * 4);
    best[o * 2] = st.best[(size_t)i * 2]; best[o * 2 + 1] = st.best[(size_t)i * 2 + 1];
    best_idx[o] = st.idx[i];
    if (embed) std::memcpy(embed + o * h->P, st.embed.data() + (size_t)i * h->P, (size_t)h->P * 4);
  }
  return MV_OK;
} catch (...) { return on_exception(h); }

// mv_forward_ragged in two halves, so that the caller can hand over batch k + 1 BEFORE it collects batch k: `begin` orders and groups the rows, gathers them into
// pinned memory, enqueues upload + passes + download on the stream of the next workspace set and returns a ticket without waiting; `end` waits for that stream and puts
// the results into the caller's arrays in the caller's row order.  At most one batch per workspace set (MEMVUL_STREAMS: 2) is in flight; tickets are collected in
// the order they were issued.  The GPU then never waits for the caller's Python between two batches (predict_memory.evaluate).
static int ragged_plan(mv_handle* h, const int32_t* lens, int B, int S, int min_tokens, std::vector<int>& order, std::vector<int32_t>& ends, std::vector<int32_t>& widths) {
  std::vector<int> pl(B);
  order.resize(B);
  for (int i = 0; i < B; ++i) {
    if (lens[i] > S) return fail(h, MV_ERR_INVALID, "mv_forward_ragged: a row is longer than S");
    pl[i] = padded_len(lens[i] < 1 ? 1 : lens[i]);
    order[i] = i;
  }
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return pl[a] < pl[b]; });
  int start = 0;
  for (int end = 1; end <= B; ++end) {
    if (end < B && pl[order[end]] == pl[order[end - 1]]) continue;  // inside a run of one padded length
    const int width = pl[order[end - 1]];
    if (end < B && (int64_t)(end - start) * width < min_tokens) continue;  // too small a pass: these rows travel with the next longer group
    if (end - start > max_rows_for(h, width < S ? width : S)) return fail(h, MV_ERR_CAPACITY, "mv_forward_ragged: a group exceeds one pass (mv_config.max_tokens)");
    ends.push_back(end);
    widths.push_back(width < S ? width : S);
    start = end;
  }
  return MV_OK;
}

int mv_forward_ragged_begin(mv_handle* h, const int32_t* ids, const int32_t* lens, int B, int S, int min_tokens, int want_logits, int want_probs, int want_embed,
                            int* ticket) try {
  if (int rc = check_ready(h)) return rc;
  if (!ids || !lens || !ticket || B <= 0 || S <= 0 || S > h->cfg.max_pos) return fail(h, MV_ERR_INVALID, "mv_forward_ragged_begin: bad argument");
  if (h->n_anchors <= 0) return fail(h, MV_ERR_STATE, "anchor bank is empty (call mv_anchor_append / mv_anchor_set first)");
  if (B > h->cfg.max_batch || (int64_t)B * S > h->cap_tokens) return fail(h, MV_ERR_CAPACITY, "mv_forward_ragged_begin: the batch exceeds mv_config.max_batch / max_tokens");
  const int slot = h->rnext % (h->n_alloc < 2 ? 1 : 2);
  auto& rs = h->rslot[slot];
  if (rs.busy) return fail(h, MV_ERR_STATE, "mv_forward_ragged_begin: the workspace set's previous batch has not been collected (mv_forward_ragged_end)");
  std::vector<int32_t> ends, widths;
  if (int rc = ragged_plan(h, lens, B, S, min_tokens, rs.order, ends, widths)) return rc;
  if (int rc = check_ids(h, ids, (int64_t)B * S, "mv_forward_ragged_begin")) return rc;
  HIPCHK(h, hipSetDevice(h->device));
  const int G = h->n_anchors;
  if (!rs.ids) {  // pinned staging of this slot, once
    const size_t mb = (size_t)h->cfg.max_batch, bg2 = mb * (size_t)h->cfg.max_anchors * 2;
    auto pin = [&](void** p, size_t bytes) -> int {
      if (hipHostMalloc(p, bytes, hipHostMallocDefault) != hipSuccess) return fail(h, MV_ERR_NOMEM, "hipHostMalloc failed (mv_forward_ragged_begin)");
      h->pinned.push_back(*p);
      return MV_OK;
    };
    int rc = pin((void**)&rs.ids, (size_t)h->cap_tokens * 4);
    if (!rc) rc = pin((void**)&rs.lens, mb * 4);
    if (!rc) rc = pin((void**)&rs.idx, mb * 4);
    if (!rc) rc = pin((void**)&rs.lg, bg2 * 4);
    if (!rc) rc = pin((void**)&rs.pr, bg2 * 4);
    if (!rc) rc = pin((void**)&rs.best, mb * 2 * 4);
    if (!rc) rc = pin((void**)&rs.emb, mb * (size_t)h->P * 4);
    if (rc) { rs.ids = nullptr; return rc; }
  }
  for (int i = 0; i < B; ++i) {
    std::memcpy(rs.ids + (size_t)i * S, ids + (size_t)rs.order[i] * S, (size_t)S * 4);
    rs.lens[i] = lens[rs.order[i]];
  }
  Work* keep = h->w;
  h->w = &h->work[slot];
  auto run = [&]() -> int {
    HIPCHK(h, hipMemcpyAsync(h->w->d_ids, rs.ids, (size_t)B * S * 4, hipMemcpyHostToDevice, h->w->stream));
    HIPCHK(h, hipMemcpyAsync(h->w->d_lens, rs.lens, (size_t)B * 4, hipMemcpyHostToDevice, h->w->stream));
    int prev = 0;
    for (size_t g = 0; g < ends.size(); ++g) {
      const int nb = ends[g] - prev;
      float* u = h->w->u + (size_t)prev * h->P;
      if (int rc = encode_dev(h, h->w->d_ids + (size_t)prev * S, h->w->d_lens + prev, pass_min_len(rs.lens + prev, nb), nb, widths[g], -1, u, false, S)) return rc;
      if (int rc = match_dev(h, u, nb, want_logits ? h->w->logits + (size_t)prev * G * 2 : nullptr, want_probs ? h->w->probs + (size_t)prev * G * 2 : nullptr, nullptr, 1,
                             h->w->best + (size_t)prev * 2, h->w->best_idx + prev)) return rc;
      prev = ends[g];
    }
    const size_t bg = (size_t)B * G;
    if (want_logits) HIPCHK(h, hipMemcpyAsync(rs.lg, h->w->logits, bg * 8, hipMemcpyDeviceToHost, h->w->stream));
    if (want_probs) HIPCHK(h, hipMemcpyAsync(rs.pr, h->w->probs, bg * 8, hipMemcpyDeviceToHost, h->w->stream));
    HIPCHK(h, hipMemcpyAsync(rs.best, h->w->best, (size_t)B * 8, hipMemcpyDeviceToHost, h->w->stream));
    HIPCHK(h, hipMemcpyAsync(rs.idx, h->w->best_idx, (size_t)B * 4, hipMemcpyDeviceToHost, h->w->stream));
    if (want_embed) HIPCHK(h, hipMemcpyAsync(rs.emb, h->w->u, (size_t)B * h->P * 4, hipMemcpyDeviceToHost, h->w->stream));
    return MV_OK;
  };
  const int rc = run();
  h->w = keep;
  if (rc != MV_OK) {
    hipStreamSynchronize(h->work[slot].stream);
    return rc;
  }
  rs.busy = true; rs.B = B; rs.G = G;
  rs.logits = want_logits != 0; rs.probs = want_probs != 0; rs.embed = want_embed != 0;
  *ticket = slot;
  h->rnext += 1;
  return MV_OK;
} catch (...) { return on_exception(h); }

int mv_forward_ragged_end(mv_handle* h, int ticket, float* logits, float* probs, float* best, int32_t* best_idx, float* embed) try {
  if (!h || ticket < 0 || ticket > 1 || !h->rslot[ticket].busy) return fail(h, MV_ERR_STATE, "mv_forward_ragged_end: no batch in flight under this ticket");
  auto& rs = h->rslot[ticket];
  HIPCHK(h, hipSetDevice(h->device));
  const hipError_t e = hipStreamSynchronize(h->work[ticket].stream);
  rs.busy = false;
  if (e != hipSuccess) return fail(h, MV_ERR_HIP, std::string("mv_forward_ragged_end: ") + hipGetErrorString(e));
  if (!best || !best_idx || (rs.logits && !logits) || (rs.probs && !probs) || (rs.embed && !embed))
    return fail(h, MV_ERR_INVALID, "mv_forward_ragged_end: an output the batch was started with is missing");
  const size_t g2 = (size_t)rs.G * 2;
  for (int i = 0; i < rs.B; ++i) {
    const size_t o = (size_t)rs.order[i];
    if (rs.logits) std::memcpy(logits + o * g2, rs.lg + (size_t)i * g2, g2 * 4);
    if (rs.probs) std::memcpy(probs + o * g2, rs.pr + (size_t)i * g2, g2 * 4);
    best[o * 2] = rs.best[(size_t)i * 2]; best[o * 2 + 1] = rs.best[(size_t)i * 2 + 1];
    best_idx[o] = rs.idx[i];
    if (rs.embed) std::memcpy(embed + o * h->P, rs.emb + (size_t)i * h->P, (size_t)h->P * 4);
  }
  return MV_OK;
} catch (...) { return on_exception(h); }

int mv_match(mv_handle* h, const float* u, int B, float* logits, float* probs, float* best, int32_t* best_idx) try {
  if (int rc = check_ready(h)) return rc;
  if (!u || B <= 0) return fail(h, MV_ERR_INVALID, "mv_match: bad argument");
  if (B > h->cfg.max_batch) return fail(h, MV_ERR_CAPACITY, "B exceeds mv_config.max_batch");
  HIPCHK(h, hipSetDevice(h->device));
  const int G = h->n_anchors;
  HIPCHK(h, hipMemcpyAsync(h->w->u_in, u, (size_t)B * h->P * 4, hipMemcpyHostToDevice, h->w->stream));
  if (int rc = match_dev(h, h->w->u_in, B, logits ? h->w->logits : nullptr, probs ? h->w->probs : nullptr, nullptr, 1, h->w->best,
                         h->w->best_idx)) return rc;
  const size_t bg = (size_t)B * G;
  if (logits) HIPCHK(h, hipMemcpyAsync(logits, h->w->logits, bg * 8, hipMemcpyDeviceToHost, h->w->stream));
  if (probs) HIPCHK(h, hipMemcpyAsync(probs, h->w->probs, bg * 8, hipMemcpyDeviceToHost, h->w->stream));
  if (best) HIPCHK(h, hipMemcpyAsync(best, h->w->best, (size_t)B * 8, hipMemcpyDeviceToHost, h->w->stream));
  if (best_idx) HIPCHK(h, hipMemcpyAsync(best_idx, h->w->best_idx, (size_t)B * 4, hipMemcpyDeviceToHost, h->w->stream));
  HIPCHK(h, hipStreamSynchronize(h->w->stream));
  return MV_OK;
} catch (...) { return on_exception(h); }

int mv_topk(mv_handle* h, const float* u, int B, int k, float* topk_p, int32_t* topk_idx) try {
  if (int rc = check_ready(h)) return rc;
  if (!u || B <= 0 || k <= 0 || k > 64 || !topk_p || !topk_idx) return fail(h, MV_ERR_INVALID, "mv_topk: bad argument (1 <= k <= 64)");
  if (B > h->cfg.max_batch) return fail(h, MV_ERR_CAPACITY, "B exceeds mv_config.max_batch");
  if (k > h->n_anchors) return fail(h, MV_ERR_INVALID, "k exceeds the number of anchors");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpyAsync(h->w->u_in, u, (size_t)B * h->P * 4, hipMemcpyHostToDevice, h->w->stream));
  // one fused pass: P(same) [B, G] never reaches HBM, only 8 B k bytes of results do
  if (int rc = match_dev(h, h->w->u_in, B, nullptr, nullptr, nullptr, k, nullptr, nullptr, h->w->topk_p, h->w->topk_idx)) return rc;
  HIPCHK(h, hipMemcpyAsync(topk_p, h->w->topk_p, (size_t)B * k * 4, hipMemcpyDeviceToHost, h->w->stream));
  HIPCHK(h, hipMemcpyAsync(topk_idx, h->w->topk_idx, (size_t)B * k * 4, hipMemcpyDeviceToHost, h->w->stream));
  HIPCHK(h, hipStreamSynchronize(h->w->stream));
  return MV_OK;
} catch (...) { return on_exception(h); }

// ---- resident corpus ---------------------------------------------------------------------------
int mv_corpus_upload(mv_handle* h, const int32_t* ids, const int32_t* lens, int64_t n, int S) try {
  if (int rc = check_ready(h)) return rc;
  if (!ids || !lens || n <= 0 || S <= 0 || S > h->cfg.max_pos) return fail(h, MV_ERR_INVALID, "mv_corpus_upload: bad argument");
  if (int rc = check_ids(h, ids, n * S, "mv_corpus_upload")) return rc;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->w->stream));
  dev_free(h, h->c_ids); dev_free(h, h->c_lens); dev_free(h, h->c_best); dev_free(h, h->c_idx); dev_free(h, h->c_psame);
  h->c_ids = nullptr; h->c_lens = nullptr; h->c_best = nullptr; h->c_idx = nullptr; h->c_psame = nullptr;
  h->c_psame_rows = 0;
  if (int rc = dev_alloc(h, &h->c_ids, n * S, false)) return rc;
  if (int rc = dev_alloc(h, &h->c_lens, n, false)) return rc;
  if (int rc = dev_alloc(h, &h->c_best, n * 2)) return rc;
  if (int rc = dev_alloc(h, &h->c_idx, n)) return rc;
  HIPCHK(h, hipMemcpyAsync(h->c_ids, ids, (size_t)n * S * 4, hipMemcpyHostToDevice, h->w->stream));
  HIPCHK(h, hipMemcpyAsync(h->c_lens, lens, (size_t)n * 4, hipMemcpyHostToDevice, h->w->stream));
  HIPCHK(h, hipStreamSynchronize(h->w->stream));
  h->c_lens_host.assign(lens, lens + n);
  h->c_n = n;
  h->c_S = S;
  return MV_OK;
} catch (...) { return on_exception(h); }

int mv_corpus_run(mv_handle* h, int64_t first, int64_t count, int batch, int keep_probs) try {
  return mv_corpus_run_len(h, first, count, batch, keep_probs, 0);
} catch (...) { return on_exception(h); }

int mv_corpus_run_len(mv_handle* h, int64_t first, int64_t count, int batch, int keep_probs, int s_eff) try {
  if (!h) return MV_ERR_INVALID;
  if (!h->finalized) return fail(h, MV_ERR_STATE, "weights not finalized (mv_finalize_weights)");
  if (!h->c_ids) return fail(h, MV_ERR_STATE, "no resident corpus (mv_corpus_upload)");
  if (first < 0 || count <= 0 || first + count > h->c_n || batch <= 0) return fail(h, MV_ERR_INVALID, "mv_corpus_run: bad range");
  if (h->n_anchors <= 0) return fail(h, MV_ERR_STATE, "anchor bank is empty");
  HIPCHK(h, hipSetDevice(h->device));
  h->w = &h->work[0];
  if (s_eff < 0 || s_eff > h->c_S) return fail(h, MV_ERR_INVALID, "mv_corpus_run_len: s_eff must be in [0, S of the resident corpus]");
  const int S_use = s_eff > 0 ? s_eff : h->c_S;  // tokens per row actually processed (rows longer than this must not be in the range)
  const int rows = max_rows_for(h, S_use);
  if (rows <= 0) return fail(h, MV_ERR_CAPACITY, "mv_config.max_tokens too small for one row of this length");
  // a batch larger than one pass holds is walked in passes of `rows` (as mv_forward / mv_encode do): a row's result
  // does not depend on the batch it travels in (bit-identical, tests/test_gpu_parity.py::test_full_batch_properties)
  if (batch > rows) batch = rows;
  const int G = h->n_anchors;
  if (keep_probs && (h->c_psame_rows != h->c_n || h->c_G != G)) {
    if (int rc = sync_all(h)) return rc;
    dev_free(h, h->c_psame);
    h->c_psame = nullptr;
    if (int rc = dev_alloc(h, &h->c_psame, h->c_n * G)) return rc;
    h->c_psame_rows = h->c_n;
    h->c_G = G;
  }
  // consecutive batches (also across calls) alternate between the two workspace sets / streams: two batches are in
  // flight at once; their results go to disjoint slices of the resident arrays
  int rc = MV_OK;
  for (int64_t off = first; off < first + count && rc == MV_OK; off += batch) {
    const int nb = (int)((first + count - off < batch) ? (first + count - off) : batch);
    h->w = &h->work[h->rr];
    if (h->n_streams == 2) {
      if (h->rr == 1) h->dual_pending = true;
      h->rr ^= 1;
    }
    rc = encode_dev(h, h->c_ids + (size_t)off * h->c_S, h->c_lens + off, pass_min_len(h->c_lens_host.data() + off, nb), nb, S_use, -1, h->w->u, false, h->c_S);
    if (rc != MV_OK) break;
    float* ps = keep_probs ? h->c_psame + (size_t)off * G : nullptr;  // P(same) [nb, G] only when the caller keeps it
    rc = match_dev(h, h->w->u, nb, nullptr, nullptr, ps, 1, h->c_best + (size_t)off * 2, h->c_idx + off);
  }
  h->w = &h->work[0];
  return rc;
} catch (...) { return on_exception(h); }

int mv_corpus_results(mv_handle* h, int64_t first, int64_t count, float* best, int32_t* best_idx, float* p_same) try {
  if (!h) return MV_ERR_INVALID;
  if (!h->c_ids) return fail(h, MV_ERR_STATE, "no resident corpus (mv_corpus_upload)");
  if (first < 0 || count <= 0 || first + count > h->c_n) return fail(h, MV_ERR_INVALID, "mv_corpus_results: bad range");
  HIPCHK(h, hipSetDevice(h->device));
  h->w = &h->work[0];
  if (int rc = sync_all(h)) return rc;
  if (best) HIPCHK(h, hipMemcpyAsync(best, h->c_best + (size_t)first * 2, (size_t)count * 8, hipMemcpyDeviceToHost, h->w->stream));
  if (best_idx) HIPCHK(h, hipMemcpyAsync(best_idx, h->c_idx + first, (size_t)count * 4, hipMemcpyDeviceToHost, h->w->stream));
  if (p_same) {
    if (!h->c_psame) return fail(h, MV_ERR_STATE, "P(same) was not kept (mv_corpus_run keep_probs=0)");
    HIPCHK(h, hipMemcpyAsync(p_same, h->c_psame + (size_t)first * h->c_G, (size_t)count * h->c_G * 4, hipMemcpyDeviceToHost, h->w->stream));
  }
  HIPCHK(h, hipStreamSynchronize(h->w->stream));
  return MV_OK;
} catch (...) { return on_exception(h); }

// ---- multi-GPU exchange: RCCL bound directly ---------------------------------------------------
// librccl.so is opened at run time (never linked).  The unique id is drawn by rank 0 (mv_comm_unique_id) and handed to every
// rank's mv_comm_init as BYTES: how they travel is the host's business (memvul_amd/distributed.py broadcasts them over its
// rendezvous socket — no id file in a shared temp directory, no single-node assumption).
int mv_comm_prepare(mv_handle* h) try {
  if (!h) return MV_ERR_INVALID;
  if (h->rccl_lib) return MV_OK;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    h->rccl_lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (h->rccl_lib) break;
  }
  if (!h->rccl_lib) return fail(h, MV_ERR_HIP, std::string("mv_comm_prepare: cannot open librccl.so: ") + dlerror());
#define RCCL_SYM(name)                                                                   \
  h->p_##name = (decltype(&name))dlsym(h->rccl_lib, #name);                            \
  if (!h->p_##name) { dlclose(h->rccl_lib); h->rccl_lib = nullptr; return fail(h, MV_ERR_HIP, "mv_comm_prepare: librccl.so lacks " #name); }
  RCCL_SYM(ncclGetUniqueId);
  RCCL_SYM(ncclCommInitRank);
  RCCL_SYM(ncclAllGather);
  RCCL_SYM(ncclCommDestroy);
  RCCL_SYM(ncclGetErrorString);
#undef RCCL_SYM
  h->p_ncclGetVersion = (decltype(&ncclGetVersion))dlsym(h->rccl_lib, "ncclGetVersion");
  h->p_ncclCommCount = (decltype(&ncclCommCount))dlsym(h->rccl_lib, "ncclCommCount");
  h->p_ncclCommUserRank = (decltype(&ncclCommUserRank))dlsym(h->rccl_lib, "ncclCommUserRank");
  return MV_OK;
} catch (...) { return on_exception(h); }

int mv_comm_unique_id(mv_handle* h, void* id_out, int capacity) try {
  if (!h || !id_out) return MV_ERR_INVALID;
  if (capacity < (int)sizeof(ncclUniqueId)) return fail(h, MV_ERR_INVALID, "mv_comm_unique_id: buffer smaller than ncclUniqueId (128 bytes)");
  if (int rc = mv_comm_prepare(h)) return rc;
  HIPCHK(h, hipSetDevice(h->device));
  ncclUniqueId id;
  ncclResult_t r = h->p_ncclGetUniqueId(&id);
  if (r != ncclSuccess) return fail(h, MV_ERR_HIP, std::string("ncclGetUniqueId: ") + h->p_ncclGetErrorString(r));
  std::memcpy(id_out, &id, sizeof(id));
  return (int)sizeof(id);
} catch (...) { return on_exception(h); }

int mv_comm_init(mv_handle* h, int rank, int world, const void* id, int id_bytes) try {
  if (!h || world < 1 || rank < 0 || rank >= world) return fail(h, MV_ERR_INVALID, "mv_comm_init: bad rank / world");
  if (h->comm) return fail(h, MV_ERR_STATE, "mv_comm_init: communicator already initialised");
  // rank / world are recorded only once the init has SUCCEEDED: a failed init leaves the handle in its one-rank state (the
  // gather is then a copy) instead of a world without a communicator
  if (world == 1 && !id) { h->comm_rank = 0; h->comm_world = 1; return MV_OK; }  // no transport needed (with an id: a real 1-rank communicator, the GPU-box test)
  if (!id || id_bytes != (int)sizeof(ncclUniqueId)) return fail(h, MV_ERR_INVALID, "mv_comm_init: the 128-byte unique id of rank 0 (mv_comm_unique_id) is required");
  if (int rc = mv_comm_prepare(h)) return rc;
  HIPCHK(h, hipSetDevice(h->device));
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  ncclResult_t r = h->p_ncclCommInitRank(&h->comm, world, uid, rank);
  if (r != ncclSuccess) { h->comm = nullptr; return fail(h, MV_ERR_HIP, std::string("ncclCommInitRank: ") + h->p_ncclGetErrorString(r)); }
  h->comm_rank = rank;
  h->comm_world = world;
  return MV_OK;
} catch (...) { return on_exception(h); }

// What the transport IS, as RCCL itself reports it: info[0] = ranks of the live communicator by ncclCommCount (0: no
// communicator — one rank, or the run is on another transport), info[1] = this rank in it by ncclCommUserRank, info[2] = the
// RCCL version code of ncclGetVersion (0 while librccl.so is not open), info[3] = the world mv_comm_allgather will gather over.
int mv_comm_info(mv_handle* h, int* info, int n) try {
  if (!h || !info || n < 4) return fail(h, MV_ERR_INVALID, "mv_comm_info: int[4] required");
  info[0] = info[1] = info[2] = 0;
  info[3] = h->comm_world;
  if (h->rccl_lib && h->p_ncclGetVersion) { int v = 0; if (h->p_ncclGetVersion(&v) == ncclSuccess) info[2] = v; }
  if (h->comm) {
    int c = -1, r = -1;
    if (h->p_ncclCommCount && h->p_ncclCommCount(h->comm, &c) == ncclSuccess) info[0] = c;
    if (h->p_ncclCommUserRank && h->p_ncclCommUserRank(h->comm, &r) == ncclSuccess) info[1] = r;
  }
  return MV_OK;
} catch (...) { return on_exception(h); }

// GPUs visible to this process (hipGetDeviceCount): 0 on a box without one (a HIP error there is "none", not a failure).
int mv_device_count(void) try {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
} catch (...) { return on_exception(nullptr); }

int mv_comm_allgather(mv_handle* h, const void* send, void* recv, int64_t bytes_per_rank) try {
  if (!h || !send || !recv || bytes_per_rank <= 0) return fail(h, MV_ERR_INVALID, "mv_comm_allgather: bad argument");
  if (h->comm_world == 1 && !h->comm) { std::memcpy(recv, send, (size_t)bytes_per_rank); return MV_OK; }
  if (!h->comm) return fail(h, MV_ERR_STATE, "mv_comm_allgather: mv_comm_init first");
  HIPCHK(h, hipSetDevice(h->device));
  if (int rc = sync_all(h)) return rc;
  hipStream_t st = h->work[0].stream;
  const int64_t total = bytes_per_rank * h->comm_world;
  if (h->comm_send_cap < bytes_per_rank) {
    if (h->comm_send) hipFree(h->comm_send);
    h->comm_send = nullptr; h->comm_send_cap = 0;
    HIPCHK(h, hipMalloc(&h->comm_send, (size_t)bytes_per_rank));
    h->comm_send_cap = bytes_per_rank;
  }
  if (h->comm_recv_cap < total) {
    if (h->comm_recv) hipFree(h->comm_recv);
    h->comm_recv = nullptr; h->comm_recv_cap = 0;
    HIPCHK(h, hipMalloc(&h->comm_recv, (size_t)total));
    h->comm_recv_cap = total;
  }
  HIPCHK(h, hipMemcpyAsync(h->comm_send, send, (size_t)bytes_per_rank, hipMemcpyHostToDevice, st));
  ncclResult_t r = h->p_ncclAllGather(h->comm_send, h->comm_recv, (size_t)bytes_per_rank, ncclChar, h->comm, st);
  if (r != ncclSuccess) return fail(h, MV_ERR_HIP, std::string("ncclAllGather: ") + h->p_ncclGetErrorString(r));
  HIPCHK(h, hipMemcpyAsync(recv, h->comm_recv, (size_t)total, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  return MV_OK;
} catch (...) { return on_exception(h); }

int mv_comm_destroy(mv_handle* h) try {
  if (!h) return MV_ERR_INVALID;
  (void)hipSetDevice(h->device);
  if (h->comm) { h->p_ncclCommDestroy(h->comm); h->comm = nullptr; }
  if (h->comm_send) { hipFree(h->comm_send); h->comm_send = nullptr; h->comm_send_cap = 0; }
  if (h->comm_recv) { hipFree(h->comm_recv); h->comm_recv = nullptr; h->comm_recv_cap = 0; }
  h->comm_world = 1; h->comm_rank = 0;
  return MV_OK;
} catch (...) { return on_exception(h); }

// ---- measurement / debug -----------------------------------------------------------------------
int mv_profile_enable(mv_handle* h, int on) try {
  if (!h) return MV_ERR_INVALID;
  h->prof = on != 0;
  return MV_OK;
} catch (...) { return on_exception(h); }

int mv_set_streams(mv_handle* h, int n) try {
  if (!h || (n != 1 && n != 2)) return fail(h, MV_ERR_INVALID, "mv_set_streams: 1 or 2");
  if (n == 2 && !h->work[1].stream) return fail(h, MV_ERR_STATE, "mv_set_streams: the second workspace set was not created (MEMVUL_STREAMS=1)");
  if (int rc = sync_all(h)) return rc;
  h->n_streams = n;
  h->rr = 0;
  return MV_OK;
} catch (...) { return on_exception(h); }

int mv_x8_saturation(mv_handle* h, int64_t* clamped, int reset) try {
  if (!h || !clamped) return fail(h, MV_ERR_INVALID, "mv_x8_saturation: bad argument");
  HIPCHK(h, hipSetDevice(h->device));
  if (int rc = sync_all(h)) return rc;
  unsigned long long v = 0;
  HIPCHK(h, hipMemcpy(&v, h->x8_sat, sizeof(v), hipMemcpyDeviceToHost));
  if (reset) HIPCHK(h, hipMemset(h->x8_sat, 0, sizeof(v)));
  *clamped = v > (unsigned long long)INT64_MAX ? INT64_MAX : (int64_t)v;
  return MV_OK;
} catch (...) { return on_exception(h); }

int mv_attention_concentration(mv_handle* h, float* max_collision, int64_t* items_over, int64_t* items_total, int reset) try {
  if (!h || !max_collision || !items_over || !items_total) return fail(h, MV_ERR_INVALID, "mv_attention_concentration: bad argument");
  HIPCHK(h, hipSetDevice(h->device));
  if (int rc = sync_all(h)) return rc;
  unsigned long long v[4] = {0, 0, 0, 0};
  HIPCHK(h, hipMemcpy(v, h->attn_conc, sizeof(v), hipMemcpyDeviceToHost));
  if (reset) HIPCHK(h, hipMemset(h->attn_conc, 0, sizeof(v)));
  const uint32_t bits = (uint32_t)v[0];
  std::memcpy(max_collision, &bits, 4);
  *items_over = v[1] > (unsigned long long)INT64_MAX ? INT64_MAX : (int64_t)v[1];
  *items_total = v[2] > (unsigned long long)INT64_MAX ? INT64_MAX : (int64_t)v[2];
  return MV_OK;
} catch (...) { return on_exception(h); }

int mv_profile_select(mv_handle* h, uint32_t class_mask) try {
  if (!h) return MV_ERR_INVALID;
  h->prof_mask = class_mask;
  return MV_OK;
} catch (...) { return on_exception(h); }

int mv_profile_read(mv_handle* h, double* ms, int64_t* launches, int n) try {
  if (!h || !ms || !launches || n < MV_NUM_KERNEL_CLASSES) return fail(h, MV_ERR_INVALID, "mv_profile_read: bad argument");
  if (int rc = sync_all(h)) return rc;
  for (int i = 0; i < n; ++i) { ms[i] = 0; launches[i] = 0; }
  for (auto& r : h->recs) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.e0, r.e1) == hipSuccess) { ms[r.cls] += t; launches[r.cls] += 1; }
    h->free_events.push_back(r.e0);
    h->free_events.push_back(r.e1);
  }
  h->recs.clear();
  return MV_OK;
} catch (...) { return on_exception(h); }

int mv_debug_encode(mv_handle* h, const int32_t* ids, const int32_t* lens, int B, int S, int n_layers) try {
  if (int rc = check_ready(h)) return rc;
  if (!ids || !lens || B <= 0 || S <= 0 || S > h->cfg.max_pos) return fail(h, MV_ERR_INVALID, "mv_debug_encode: bad argument");
  if (B > max_rows_for(h, S)) return fail(h, MV_ERR_CAPACITY, "mv_debug_encode: batch too large for one pass");
  if (int rc = check_ids(h, ids, (int64_t)B * S, "mv_debug_encode")) return rc;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpyAsync(h->w->d_ids, ids, (size_t)B * S * 4, hipMemcpyHostToDevice, h->w->stream));
  HIPCHK(h, hipMemcpyAsync(h->w->d_lens, lens, (size_t)B * 4, hipMemcpyHostToDevice, h->w->stream));
  if (int rc = encode_dev(h, h->w->d_ids, h->w->d_lens, pass_min_len(lens, B), B, S, n_layers < 0 ? h->cfg.layers : n_layers, h->w->u, /*full=*/true)) return rc;
  HIPCHK(h, hipStreamSynchronize(h->w->stream));
  return MV_OK;
} catch (...) { return on_exception(h); }

int mv_debug_read(mv_handle* h, int buffer, void* dst, int64_t bytes) try {
  if (!h || !dst || bytes <= 0) return MV_ERR_INVALID;
  const int64_t T = (int64_t)h->dbg_B * h->dbg_Sp;
  const void* src = nullptr;
  int64_t avail = 0;
  switch (buffer) {
    case 0: src = h->w->xres; avail = T * MV_HIDDEN * 4; break;
    case 1: src = h->w->x16; avail = T * MV_HIDDEN * 2; break;
    case 2: src = h->w->q; avail = T * MV_HIDDEN * 2; break;
    case 3: src = h->w->k; avail = T * MV_HIDDEN * 2; break;
    case 4: src = h->w->vt; avail = T * MV_HIDDEN * 2; break;
    case 5: src = h->w->ctx; avail = T * MV_HIDDEN * 2; break;
    case 6: src = h->w->h16; avail = T * MV_INTER * 2; break;
    case 7: src = h->w->u; avail = (int64_t)h->dbg_B * h->P * 4; break;
    default: return fail(h, MV_ERR_INVALID, "mv_debug_read: unknown buffer");
  }
  if (bytes > avail) return fail(h, MV_ERR_INVALID, "mv_debug_read: more bytes requested than the buffer holds");
  HIPCHK(h, hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, h->w->stream));
  HIPCHK(h, hipStreamSynchronize(h->w->stream));
  return MV_OK;
} catch (...) { return on_exception(h); }

int mv_test_gemm(mv_handle* h, int variant, int M, int N, int K, const uint16_t* A, const uint16_t* W, const float* bias,
                 float* C, int iters, float* ms) try {
  if (!h || !A || !W || M <= 0 || N <= 0 || K <= 0) return fail(h, MV_ERR_INVALID, "mv_test_gemm: bad argument");
  if (variant != 0 && variant != 19) return fail(h, MV_ERR_INVALID, "mv_test_gemm: variant 0 (128^2 tile) or 19 (64^2 ring)");
  if (variant == 0 && (M % 128 || N % 128 || K % 64)) return fail(h, MV_ERR_INVALID, "mv_test_gemm: M,N % 128 and K % 64 required");
  HIPCHK(h, hipSetDevice(h->device));
  half_t *dA = nullptr, *dW = nullptr;
  float *dB = nullptr, *dC = nullptr;
  int rc;
  if ((rc = dev_alloc(h, &dA, (int64_t)M * K, false))) return rc;
  if ((rc = dev_alloc(h, &dW, (int64_t)N * K, false))) return rc;
  if ((rc = dev_alloc(h, &dB, N))) return rc;
  if ((rc = dev_alloc(h, &dC, (int64_t)M * N))) return rc;
  HIPCHK(h, hipMemcpyAsync(dA, A, (size_t)M * K * 2, hipMemcpyHostToDevice, h->w->stream));
  HIPCHK(h, hipMemcpyAsync(dW, W, (size_t)N * K * 2, hipMemcpyHostToDevice, h->w->stream));
  if (bias) HIPCHK(h, hipMemcpyAsync(dB, bias, (size_t)N * 4, hipMemcpyHostToDevice, h->w->stream));
  GemmArgs g{};
  g.A = dA; g.W = dW; g.bias = dB; g.M = M; g.Mreal = M; g.N = N; g.K = K; g.outf = dC; g.S = 64;
  if (iters < 1) iters = 1;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto run = [&]() -> int { return variant == 0 ? launch_gemm128<EPI_F32>(h, KC_TEST_GEMM, g) : launch_ring64<EPI_F32>(h, KC_TEST_GEMM, g); };
  rc = run();  // warm-up / correctness launch
  if (rc == MV_OK) {
    hipEventRecord(e0, h->w->stream);
    for (int i = 0; i < iters && rc == MV_OK; ++i) rc = run();
    hipEventRecord(e1, h->w->stream);
  }
  hipError_t se = hipStreamSynchronize(h->w->stream);
  float t = 0.f;
  hipEventElapsedTime(&t, e0, e1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  if (ms) *ms = t / (float)iters;
  if (rc == MV_OK && se != hipSuccess) rc = fail(h, MV_ERR_HIP, std::string("test gemm: ") + hipGetErrorString(se));
  if (rc == MV_OK && C) {
    se = hipMemcpy(C, dC, (size_t)M * N * 4, hipMemcpyDeviceToHost);
    if (se != hipSuccess) rc = fail(h, MV_ERR_HIP, std::string("test gemm copy: ") + hipGetErrorString(se));
  }
  dev_free(h, dA); dev_free(h, dW); dev_free(h, dB); dev_free(h, dC);
  return rc;
} catch (...) { return on_exception(h); }

// The FFN-1 kernel of the persistent path (gemm_pp_kernel<PP_GELU, RAW>) on caller-provided fp32 operands with unit row
// statistics: out16 = fp16(gelu(A W^T + bias)) [M][N]; x8 != 0: the MV_F16X8 build (fp16 sweep + fp8 correction sweep) and, with
// out8, the [lo8 | hi8] planes of the output [M][2 N].  A / W are split into their planes on the host exactly as
// mv_finalize_weights does for weights (W) and as the producing epilogues do for activations (A: shift MV_X8_ACT_SHIFT).
int mv_test_gemm_pp(mv_handle* h, int x8, int M, int N, int K, const float* A, const float* W, const float* bias, uint16_t* out16,
                    uint8_t* out8, int iters, float* ms) try {
  if (!h || !A || !W || !bias || !out16 || M <= 0 || N <= 0 || K <= 0) return fail(h, MV_ERR_INVALID, "mv_test_gemm_pp: bad argument");
  if (M % 256 || N % 256 || K % 128 || K < 256 || N > MV_INTER)
    return fail(h, MV_ERR_INVALID, "mv_test_gemm_pp: M,N % 256, K % 128, K >= 256, N <= 3072 required");
  if (x8 == 2 && K % 256) return fail(h, MV_ERR_INVALID, "mv_test_gemm_pp: the weight-side-only sweep walks K / 128 K-tiles in pairs: K % 256 required");
  HIPCHK(h, hipSetDevice(h->device));
  std::vector<uint16_t> a16((size_t)M * K), w16((size_t)N * K);
  for (size_t i = 0; i < a16.size(); ++i) a16[i] = f32_to_f16_bits(A[i]);
  for (size_t i = 0; i < w16.size(); ++i) w16[i] = f32_to_f16_bits(W[i]);
  std::vector<uint8_t> a8, w8;
  int scale_word = 0;
  if (x8) {
    make_x8_weight_planes(W, N, K, w8, &scale_word);
    a8.resize((size_t)M * 2 * K);
    const float sh = std::ldexp(1.0f, MV_X8_ACT_SHIFT), sl = std::ldexp(1.0f, 11 + MV_X8_ACT_SHIFT);
    for (int64_t m = 0; m < M; ++m)
      for (int64_t k = 0; k < K; ++k) {
        const float v = A[m * K + k], hi = f16_bits_to_f32(a16[(size_t)(m * K + k)]);
        a8[(size_t)(m * 2 * K + k)] = f32_to_e4m3_bits((v - hi) * sl);   // [lo8 | hi8]
        a8[(size_t)(m * 2 * K + K + k)] = f32_to_e4m3_bits(v * sh);
      }
  }
  std::vector<float> st((size_t)M * 6, 0.f);
  for (int64_t m = 0; m < M; ++m) st[(size_t)m * 6 + 1] = (float)MV_HIDDEN;  // (sum, sumsq) = (0, 768): mean 0, rstd 1
  half_t *dA = nullptr, *dW = nullptr, *dO = nullptr;
  uint8_t *dA8 = nullptr, *dW8 = nullptr, *dO8 = nullptr;
  float *dB = nullptr, *dS = nullptr;
  int rc;
  if ((rc = dev_alloc(h, &dA, (int64_t)M * K, false))) return rc;
  if ((rc = dev_alloc(h, &dW, (int64_t)N * K, false))) return rc;
  if ((rc = dev_alloc(h, &dO, (int64_t)M * N))) return rc;
  if ((rc = dev_alloc(h, &dB, N, false))) return rc;
  if ((rc = dev_alloc(h, &dS, (int64_t)M * 6, false))) return rc;
  HIPCHK(h, hipMemcpyAsync(dA, a16.data(), a16.size() * 2, hipMemcpyHostToDevice, h->w->stream));
  HIPCHK(h, hipMemcpyAsync(dW, w16.data(), w16.size() * 2, hipMemcpyHostToDevice, h->w->stream));
  HIPCHK(h, hipMemcpyAsync(dB, bias, (size_t)N * 4, hipMemcpyHostToDevice, h->w->stream));
  HIPCHK(h, hipMemcpyAsync(dS, st.data(), st.size() * 4, hipMemcpyHostToDevice, h->w->stream));
  GemmArgs g{};
  g.A = dA; g.W = dW; g.bias = dB; g.M = M; g.Mreal = M; g.N = N; g.K = K; g.out16 = dO; g.S = 64; g.lnstats = dS; g.ln_eps = 0.f;
  if (x8) {
    if ((rc = dev_alloc(h, &dA8, (int64_t)a8.size(), false))) return rc;
    if ((rc = dev_alloc(h, &dW8, (int64_t)w8.size(), false))) return rc;
    if ((rc = dev_alloc(h, &dO8, (int64_t)M * 2 * N))) return rc;
    HIPCHK(h, hipMemcpyAsync(dA8, a8.data(), a8.size(), hipMemcpyHostToDevice, h->w->stream));
    HIPCHK(h, hipMemcpyAsync(dW8, w8.data(), w8.size(), hipMemcpyHostToDevice, h->w->stream));
    g.A8 = dA8; g.W8 = dW8; g.out8 = dO8; g.x8_scale = scale_word;
    g.x8_terms = (x8 == 2) ? 1 : 2;  // x8 = 2: the weight-side term only (the QKV projection's form)
  }
  if (iters < 1) iters = 1;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  rc = launch_pp<PP_GELU>(h, KC_TEST_GEMM, g);
  if (rc == MV_OK) {
    hipEventRecord(e0, h->w->stream);
    for (int i = 0; i < iters && rc == MV_OK; ++i) rc = launch_pp<PP_GELU>(h, KC_TEST_GEMM, g);
    hipEventRecord(e1, h->w->stream);
  }
  hipError_t se = hipStreamSynchronize(h->w->stream);
  float t = 0.f;
  hipEventElapsedTime(&t, e0, e1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  if (ms) *ms = t / (float)iters;
  if (rc == MV_OK && se != hipSuccess) rc = fail(h, MV_ERR_HIP, std::string("test gemm_pp: ") + hipGetErrorString(se));
  if (rc == MV_OK) {
    se = hipMemcpy(out16, dO, (size_t)M * N * 2, hipMemcpyDeviceToHost);
    if (se == hipSuccess && x8 && out8) se = hipMemcpy(out8, dO8, (size_t)M * 2 * N, hipMemcpyDeviceToHost);
    if (se != hipSuccess) rc = fail(h, MV_ERR_HIP, std::string("test gemm_pp copy: ") + hipGetErrorString(se));
  }
  dev_free(h, dA); dev_free(h, dW); dev_free(h, dO); dev_free(h, dB); dev_free(h, dS);
  dev_free(h, dA8); dev_free(h, dW8); dev_free(h, dO8);
  return rc;
} catch (...) { return on_exception(h); }

// host-side e4m3 encoder of the MV_F16X8 weight planes (no GPU needed): tests pin it to the oracle's rounding model
int mv_test_e4m3(const float* in, uint8_t* out, int64_t n) try {
  if (!in || !out || n < 0) return MV_ERR_INVALID;
  for (int64_t i = 0; i < n; ++i) out[i] = f32_to_e4m3_bits(in[i]);
  return MV_OK;
} catch (...) { return on_exception(nullptr); }

// ---- JSON-lines records of one batch (host only) --------------------------------------------------------------------------------------
// Python's repr(float) — what json.dumps prints for every probability of make_output_human_readable's records (model_memory.py:169-191 ->
// predict_memory.py:111) — restated: the shortest digit string that round-trips the double (std::to_chars, scientific) laid out by CPython's rule
// (PyOS_double_to_string 'r': exponent form when the decimal exponent is < -4 or >= 16, at least two exponent digits, ".0" after a whole number).
// 0.65 us per double in CPython, ~40 ns here; pinned to repr() on millions of values by tests/test_host_logic.py.
static inline char* py_repr_double(char* o, double v) {
  if (v == 0.0) {
    if (std::signbit(v)) *o++ = '-';
    *o++ = '0'; *o++ = '.'; *o++ = '0';
    return o;
  }
  char b[40];
  const auto r = std::to_chars(b, b + sizeof b, v, std::chars_format::scientific);  // [-]d[.ddd]e[+-]XX
  const char* p = b;
  if (*p == '-') *o++ = *p++;
  char dig[24];
  int nd = 0;
  dig[nd++] = *p++;
  if (*p == '.') {
    ++p;
    while (*p != 'e') dig[nd++] = *p++;
  }
  ++p;  // 'e'
  const bool eneg = *p == '-';
  ++p;
  int e = 0;
  while (p < r.ptr) e = e * 10 + (*p++ - '0');
  if (eneg) e = -e;
  if (e < -4 || e >= 16) {
    *o++ = dig[0];
    if (nd > 1) {
      *o++ = '.';
      for (int i = 1; i < nd; ++i) *o++ = dig[i];
    }
    *o++ = 'e';
    *o++ = e < 0 ? '-' : '+';
    const int ae = e < 0 ? -e : e;
    if (ae >= 100) *o++ = (char)('0' + ae / 100);
    *o++ = (char)('0' + (ae / 10) % 10);
    *o++ = (char)('0' + ae % 10);
  } else if (e < 0) {
    *o++ = '0'; *o++ = '.';
    for (int i = 0; i < -e - 1; ++i) *o++ = '0';
    for (int i = 0; i < nd; ++i) *o++ = dig[i];
  } else {
    for (int i = 0; i <= e; ++i) *o++ = i < nd ? dig[i] : '0';
    *o++ = '.';
    if (nd > e + 1) for (int i = e + 1; i < nd; ++i) *o++ = dig[i];
    else *o++ = '0';
  }
  return o;
}

// out = "[" + ", ".join(prefix_i + piece_0 + repr(p[i][0]) + piece_1 + repr(p[i][1]) + ... + row_suffix) + "]"
int mv_format_records(const char* prefixes, const int64_t* prefix_off, int64_t rows, const char* pieces, const int64_t* piece_off, int64_t cols,
                      const char* row_suffix, const double* p, char* out, int64_t cap, int64_t* written) try {
  if (!prefixes || !prefix_off || !pieces || !piece_off || !row_suffix || !p || !out || !written || rows < 0 || cols < 0) return MV_ERR_INVALID;
  const int64_t nsuf = (int64_t)std::strlen(row_suffix);
  const int64_t piece_bytes = piece_off[cols] - piece_off[0];
  char* o = out;
  char* const end = out + cap;
  if (end - o < 2) return MV_ERR_CAPACITY;
  *o++ = '[';
  for (int64_t i = 0; i < rows; ++i) {
    const int64_t np_ = prefix_off[i + 1] - prefix_off[i];
    if (end - o < np_ + piece_bytes + cols * 26 + nsuf + 4) return MV_ERR_CAPACITY;  // (a repr is at most 24 characters)
    if (i) { *o++ = ','; *o++ = ' '; }
    std::memcpy(o, prefixes + prefix_off[i], (size_t)np_);
    o += np_;
    const double* row = p + i * cols;
    for (int64_t c = 0; c < cols; ++c) {
      const int64_t n = piece_off[c + 1] - piece_off[c];
      std::memcpy(o, pieces + piece_off[c], (size_t)n);
      o += n;
      if (!std::isfinite(row[c])) return MV_ERR_INVALID;  // json spells these NaN / Infinity: the caller's Python path does
      o = py_repr_double(o, row[c]);
    }
    std::memcpy(o, row_suffix, (size_t)nsuf);
    o += nsuf;
  }
  *o++ = ']';
  *written = o - out;
  return MV_OK;
} catch (...) { return on_exception(nullptr); }

}  // extern "C"
