// Internal definitions shared by the HIP translation units of libpvsim_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <vector>

#include "../../include/pvsim.h"
#include "../../include/pvsim_diag.h"

#define PVS_EXPORT extern "C" __attribute__((visibility("default")))

namespace pvs {

void set_error(const char* fmt, ...);

#define PVS_FAIL(code, ...)        \
  do {                             \
    ::pvs::set_error(__VA_ARGS__); \
    return (code);                 \
  } while (0)

#define PVS_HIP(expr)                                                                    \
  do {                                                                                   \
    hipError_t e__ = (expr);                                                             \
    if (e__ != hipSuccess) {                                                             \
      ::pvs::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, \
                       __LINE__);                                                        \
      return e__ == hipErrorOutOfMemory ? PVS_ERR_OOM : PVS_ERR_NO_DEVICE;               \
    }                                                                                    \
  } while (0)

#define PVS_TRY(expr)                 \
  do {                                \
    int s__ = (expr);                 \
    if (s__ != PVS_OK) return s__;    \
  } while (0)

constexpr int WAVE = 64;

enum TimerSlot { T_ASSIGN = 0, T_AGGREGATE = 1, T_GEMM = 2, T_TOPK = 3, T_FPOST = 4, T_FMOM = 5, T_MISC = 6, T_RESCORE = 7 };

struct TimerRec {
  hipEvent_t a, b;
  int slot;
};

}  // namespace pvs

struct pvs_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool owns_stream = false;
  int num_cu = 256;
  // grow-only scratch areas (device)
  // 0 host-API input staging, 1 scratch (labels, tables, responsibilities), 2 host-API outputs / score panel,
  // 3 PCA projections, 4 materialised RootSIFT rows
  // 5 fp16 row copies (filtered top-k), 6 filtered top-k lists / candidates
  static constexpr int NWS = 7;
  void* ws[NWS] = {};
  size_t ws_bytes[NWS] = {};
  // cached tile lists of the similarity GEMM, one per GEMM model (cosine.hip): a context is one device + one stream, so the
  // list a launch reads can only be replaced by work queued behind it on the same stream
  struct GemmPlanSlot {
    int key[4] = {-1, -1, -1, -1};   // tiles_m, tiles_n, symmetric, resident workgroup slots
    int n_main = 0, n_tail = 0, splitk = 1;
    void* d_tiles = nullptr;
    size_t cap = 0;
  } gemm_plan[5];   // 0 exact fp32, 1 fp16 256x256, 2 fp16 128x128 two-level, 3 float64, 4 fp16 256x128 two-level
  // dynamic-LDS limits already raised on this context's device: kernel -> bytes
  std::map<const void*, int> lds_attr;
  // behaviour switches (pvs_set_option); defaults = the product path
  int opt[PVS_OPT_COUNT_] = {1, 0, 0, 0, 0};
  unsigned int* d_queue = nullptr;   // image queue head of the fused encode (persistent workgroups)
  unsigned long long* d_fused_stamps = nullptr;   // non-null: fused launches run the stamped diagnostic kernel (pvs_fused_profile)
  // timers
  bool timers_on = false;
  std::vector<pvs::TimerRec> pending;
  std::vector<hipEvent_t> event_pool;    // recycled timer events (destroyed with the context)
  // device blocks of destroyed table objects (codebooks, mixtures, projections), kept for the next one: a training loop creates
  // and destroys a codebook per iteration, and a hipFree can stall the host for tens of milliseconds (measured: 74 ms)
  std::multimap<size_t, void*> block_cache;      // capacity -> block
  std::map<void*, size_t> block_size;            // live blocks handed out by table_alloc
  size_t block_cache_bytes = 0;
  void* h_stage = nullptr;                        // pinned host staging block for small device -> host results (training statistics)
  size_t h_stage_bytes = 0;
  double t_total[PVS_TIMER_SLOTS] = {0};
  int64_t t_count[PVS_TIMER_SLOTS] = {0};
};

struct pvs_codebook {
  int K = 0, D = 0;
  int K_pad = 0, D_pad = 0;  // K_pad multiple of 32, D_pad multiple of 8 (zero / +inf padded)
  float* d_cent = nullptr;   // [K][D]     exact user layout (for the aggregate step)
  float* d_cpad = nullptr;   // [K_pad][D_pad] zero padded (MFMA operand)
  float* d_cnorm = nullptr;  // [K_pad]   ||c||^2, +inf on padded clusters
  // fp16 copy for the assignment prefilter (null when K_pad > 256, D > 128 or a centre leaves the fp16 range)
  void* d_c16 = nullptr;     // [2][K_pad][D_pad16] _Float16: hi = fl16(c 2^c16_shift), lo = fl16(c 2^c16_shift - hi); zero padded
  int D_pad16 = 0;           // multiple of 16
  int c16_shift = 0;         // the largest |c| 2^shift lies in [2^12, 2^13)
  float cmax = 0.f;          // max_k ||c_k||_2
  void* d_cnk = nullptr;     // [K_pad][4] _Float16: three pieces of -|c|^2/2 2^(c16_shift - cn_e1) (padded clusters: -65504, 0, 0), 0
  // tables of the fused encode (vlad_fused.hip; K_pad == 256 and D == 128 only)
  void* d_c16n = nullptr;    // [2][256][128] _Float16: the same hi | lo values in natural dim order, zero rows for padded clusters
  int cn_e1 = 0;             // the largest |c|^2/2 2^(c16_shift - cn_e1) lies in [2^12, 2^13)
};

struct pvs_gmm {
  int K = 0, D = 0;
  double* d_w = nullptr;     // [K]
  double* d_mu = nullptr;    // [K][D]
  double* d_cov = nullptr;   // [K][D]
  // derived, fp64 (sklearn/mixture/_gaussian_mixture.py:495-512)
  double* d_prec = nullptr;  // [K][D]  1/cov
  double* d_mup = nullptr;   // [K][D]  mu/cov
  double* d_inv_mu = nullptr; // [K][D]  1/(sqrt(w) sqrt(cov))      (Fisher normalisation, fisher_vector.py:118-120)
  double* d_inv_sg = nullptr; // [K][D]  1/(sqrt(2) sqrt(w) cov)
  double* d_const = nullptr; // [K]     -0.5*(D log 2pi + sum mu^2/cov) + sum log(1/sqrt(cov)) + log w
};

struct pvs_pca {
  int C = 0, Din = 0;
  float* d_comp = nullptr;   // [C][Din]
  float* d_off = nullptr;    // [C]  mean @ components^T
};

namespace pvs {

int ws_reserve(pvs_ctx* ctx, int which, size_t bytes, void** out);

// raise a kernel's dynamic-LDS limit once per context (and again when a launch needs more)
inline int ensure_lds(pvs_ctx* ctx, const void* fn, size_t bytes) {
  auto it = ctx->lds_attr.find(fn);
  if (it == ctx->lds_attr.end() || it->second < (int)bytes) {
    PVS_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    ctx->lds_attr[fn] = (int)bytes;
  }
  return PVS_OK;
}

struct ScopedTimer {
  pvs_ctx* ctx;
  int slot;
  hipEvent_t a = nullptr, b = nullptr;
  hipEvent_t take_event() {
    if (!ctx->event_pool.empty()) {
      hipEvent_t e = ctx->event_pool.back();
      ctx->event_pool.pop_back();
      return e;
    }
    hipEvent_t e = nullptr;
    return hipEventCreate(&e) == hipSuccess ? e : nullptr;
  }
  ScopedTimer(pvs_ctx* c, int s) : ctx(c), slot(s) {
    if (ctx->timers_on) {
      // a long-running caller that never reads the timers must not pile up events: fold in the finished ones (in order,
      // non-blocking) once a few hundred are pending
      // (events are recycled through ctx->event_pool: creating and destroying a pair per launch cost ~0.15 ms, and folding
      // 512 pairs at once was a 75-ms stall inside the timed region of a loop with many launches)
      if (ctx->pending.size() >= 128) {
        size_t done = 0;
        while (done < ctx->pending.size() && hipEventQuery(ctx->pending[done].b) == hipSuccess) {
          float ms = 0.f;
          if (hipEventElapsedTime(&ms, ctx->pending[done].a, ctx->pending[done].b) == hipSuccess) {
            ctx->t_total[ctx->pending[done].slot] += ms;
            ctx->t_count[ctx->pending[done].slot] += 1;
          }
          ctx->event_pool.push_back(ctx->pending[done].a);
          ctx->event_pool.push_back(ctx->pending[done].b);
          ++done;
        }
        ctx->pending.erase(ctx->pending.begin(), ctx->pending.begin() + done);
      }
      a = take_event();
      b = take_event();
      if (a && b) (void)hipEventRecord(a, ctx->stream);
      else a = nullptr;
    }
  }
  ~ScopedTimer() {
    if (a) {
      (void)hipEventRecord(b, ctx->stream);
      ctx->pending.push_back({a, b, slot});
    }
  }
};

// ---- launchers implemented in the kernel translation units (all enqueue on ctx->stream)
int launch_assign(pvs_ctx* ctx, const pvs_codebook* cb, const void* d_desc, int kind, int64_t total, int ld,
                  int32_t* d_labels, const float2** rowstat_out = nullptr);
int launch_vlad_aggregate(pvs_ctx* ctx, const pvs_codebook* cb, const void* d_desc, int kind, int ld,
                          const int64_t* d_offsets, int64_t n_images, const int32_t* d_labels,
                          const pvs_norm_params& prm, float* d_out, float* d_inv_norm, bool raw = false,
                          const float2* rowstat = nullptr, int64_t total_hint = 0);
bool vlad_fused_eligible(const pvs_codebook* cb, const void* d_desc, int kind, int ld, const float* d_out);
int launch_vlad_fused(pvs_ctx* ctx, const pvs_codebook* cb, const void* d_desc, int kind, int ld, const int64_t* d_offsets,
                      int64_t n_images, const pvs_norm_params& prm, float* d_out, int32_t* d_labels, float* d_inv_norm, bool raw = false);
int launch_row_inv_norms(pvs_ctx* ctx, const float* d_x, int64_t rows, int64_t L, float* d_inv);
int launch_cosine_f32(pvs_ctx* ctx, const float* A, int64_t M, const float* B, int64_t N, int64_t L,
                      const float* inva, const float* invb, float* out, int64_t ldo);
int launch_cosine_f32_dual(pvs_ctx* ctx, const float* A, int64_t M, const float* B, int64_t N, int64_t L, const float* inva,
                           const float* invb, float* out, int64_t ldo, float* out_t, int64_t ldt);
int launch_cosine_f16(pvs_ctx* ctx, const void* A, int64_t M, const void* B, int64_t N, int64_t L, const float* inva,
                      const float* invb, float* out, int64_t ldo);
int launch_f32_to_f16(pvs_ctx* ctx, const float* src, int64_t n, void* dst);
int launch_cosine_f16_bounded(pvs_ctx* ctx, const void* A, int64_t M, const void* B, int64_t N, int64_t L, const float* inva,
                              const float* invb, float* out, int64_t ldo);
// the plain exact path for device operands (api.hip): f32 GEMM panels + select over all database rows
int cosine_topk_exact(pvs_ctx* ctx, const float* d_Q, int64_t nq, const float* d_DB, int64_t N, int64_t L, const float* d_inv_q,
                      const float* d_inv_db, int k, int64_t* d_idx, float* d_val);
// exact top-k through an fp16 prefilter + exact re-scoring (filter.hip); returns PVS_ERR_UNSUPPORTED when the inputs
// do not qualify (the caller then runs the plain exact path)
int launch_cosine_topk_filtered(pvs_ctx* ctx, const float* Q, int64_t nq, const float* DB, int64_t N, int64_t L,
                                const float* invq, const float* invdb, int k, int64_t col_offset, int64_t* d_idx, float* d_val,
                                int64_t* h_stats);
int launch_cosine_f64(pvs_ctx* ctx, const double* A, int64_t M, const double* B, int64_t N, int64_t L, double* out);
int launch_row_inv_norms_f64(pvs_ctx* ctx, const double* d_x, int64_t rows, int64_t L, double* d_inv);
int launch_cosine_f64_dev(pvs_ctx* ctx, const double* A, int64_t M, const double* B, int64_t N, int64_t L, const double* inva,
                          const double* invb, double* out, int64_t ldo);
bool cosine_dense_rows_eligible(const float* Q, const float* DB, int64_t nq, int64_t L);
int launch_cosine_dense_rows(pvs_ctx* ctx, const float* Q, int64_t nq, const float* DB, int64_t N, int64_t L, const float* invq,
                             const float* invdb, float* scores, int64_t ld);
int launch_topk(pvs_ctx* ctx, const float* scores, int64_t nq, int64_t ncols, int64_t ld, int k,
                int64_t col_offset, int merge, int64_t* d_idx, float* d_val);
int launch_topk_merge(pvs_ctx* ctx, const int64_t* idx_lists, const float* val_lists, int n_lists, int64_t nq,
                      int k, int64_t* d_idx, float* d_val);
int launch_rank_f64(pvs_ctx* ctx, const double* scores, int64_t nq, int64_t ncols, int64_t ld, int k, int64_t* d_idx, double* d_val);
int launch_pca(pvs_ctx* ctx, const pvs_pca* p, const void* d_desc, int kind, int64_t total, float* d_out);
int launch_gmm_posterior(pvs_ctx* ctx, const pvs_gmm* g, const void* d_desc, int kind, int ld, int64_t total,
                         double* d_resp);
int launch_fisher(pvs_ctx* ctx, const pvs_gmm* g, const void* d_desc, int kind, int ld, const int64_t* d_offsets,
                  int64_t n_images, int64_t total, const pvs_norm_params& prm, void* d_out, int out_f64);

// training (learn.hip / fisher.hip); x are plain fp32 rows (launch_materialise output), ld = D
int launch_materialise(pvs_ctx* ctx, const void* d_desc, int kind, int64_t total, int D, float* d_out);
int launch_gmm_em_step(pvs_ctx* ctx, const pvs_gmm* g, const float* x, int ld, int64_t total, double* d_stats);
int launch_kmeans_step(pvs_ctx* ctx, const pvs_codebook* cb, const float* x, int64_t total, int32_t* d_labels,
                       const int32_t* d_prev_labels, double* d_stats, float* d_sqdist);
int launch_label_sums(pvs_ctx* ctx, const float* x, int64_t total, int D, const int32_t* d_labels, int K, int square, double* d_out);
int launch_gram(pvs_ctx* ctx, const float* x, int64_t total, int D, double* d_out);
int launch_seed_distances(pvs_ctx* ctx, const float* x, int64_t total, int D, const float* d_cand, int n_cand,
                          const float* d_mind, float* d_dist, double* d_pot);
int launch_seed_pick(pvs_ctx* ctx, const float* x, int64_t total, int D, const float* d_mind, const int64_t* d_blk,
                     const double* d_base, const double* d_target, int n_cand, int64_t* d_idx, float* d_cand);
int launch_min_update(pvs_ctx* ctx, float* d_mind, const float* d_dist, int64_t total, double* d_block_sums);
int launch_kmeanspp_run(pvs_ctx* ctx, const float* x, int64_t total, int D, int n_clusters, int trials, const double* d_uniform,
                        float* d_mind, float* d_dist, float* d_cand, double* d_block_sums, char* d_small, int64_t* d_indices);

}  // namespace pvs
