// Exact top-k retrieval through an fp16 prefilter (filtered top-k).
//
// The reference ranks every query against the whole database in fp32 (pyvisim/eval.py:37-43: cosine_similarity then
// argsort(-s)[:k]); the exact device path does the same with the f32 MFMA GEMM, which runs at the vector-FMA rate
// (157 TFLOP/s).  Only the k best scores of a row are returned, so this path
//   1. scores all pairs with fp16 operands on the f16 MFMA (16x the f32 rate) with a PROVEN error bound eps,
//   2. keeps, per query, every column whose approximate score is within 2 eps of the approximate k-th best
//      (a superset of the exact top-k, ties included),
//   3. re-scores the kept pairs with the exact fp32 recurrence of the f32 GEMM kernel, and
//   4. ranks them with the same (score desc, index asc) keys.
// The returned lists are bit-identical to those of the exact path -- scores included -- because step 3 reproduces the
// f32 MFMA accumulation bit for bit: v_mfma_f32_32x32x2_f32 adds its two products with two fused multiply-adds in
// lane-half order, so a score is the chain  acc = fma(a[k], b[k], acc)  over k in the order (8t+e, 8t+4+e), e < 4,
// in chains of 1024 k whose sums are added in order (gemm_mfma.hpp; verified on MI355X against a scalar recurrence,
// csrc/bench/gemm_variants.hip "chain").
//
// Error bound of step 1 for normalised scores (rows scaled by a power of two into the fp16 range first, so neither
// overflow nor the subnormal range matters): fp16 rounding of both operands  <= (2u + u^2) sum|a b| <= 2^-10 (1 + 2^-12)
// by Cauchy-Schwarz; fp32 accumulation in chains of 1024 k plus the chain sums  <= (1088 + L/1024) 2^-23; entries
// flushed below the fp16 normal range  <= 2^-25 sqrt(L).  eps(L) adds these; the margin is 2 eps.
#include <algorithm>
#include <cmath>
#include <vector>

#include "common.hpp"

namespace pvs {

constexpr int FLT_PASS = 16;   // candidates re-scored together by one workgroup
constexpr int FLT_SK = 32;     // k-values per chain slab staged in LDS
constexpr int FLT_THREADS = FLT_PASS * 32;

static double filter_eps(int64_t L) {
  // (+ the roundings of the per-segment scaling and accumulation for rows longer than 32768)
  return 9.77e-4 * (1.0 + 1.0 / 4096.0) + (1088.0 + (double)L / 1024.0 + 2.0 * (double)((L + 32767) / 32768)) * 1.1920929e-7 +
         2.98e-8 * std::sqrt((double)L);
}

// ---- rows -> fp16, each row scaled by a power of two so that its largest magnitude lies in [2^14, 2^15)
// stats[0] += rows whose values are not finite or whose scale falls outside 2^+-30 (the caller then takes the exact path)
__global__ __launch_bounds__(256) void filter_rows_to_f16_kernel(const float* __restrict__ x, int64_t rows, int64_t L,
                                                                 const float* __restrict__ inv_in, _Float16* __restrict__ out,
                                                                 float* __restrict__ inv_out, unsigned long long* __restrict__ stats) {
  __shared__ float red[4];
  __shared__ int bad[4];
  const int64_t row = blockIdx.x;
  const float4* xr = reinterpret_cast<const float4*>(x + row * L);
  const int64_t n4 = L / 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float amax = 0.f;
  int nf = 0;
  for (int64_t i = threadIdx.x; i < n4; i += 256) {
    const float4 v = xr[i];
    const float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    nf |= !((fabsf(v.x) + fabsf(v.y)) + (fabsf(v.z) + fabsf(v.w)) <= 3.0e38f) ? 1 : 0;   // NaN and inf propagate through the sum
    amax = fmaxf(amax, m);
  }
  for (int m = 32; m >= 1; m >>= 1) {
    amax = fmaxf(amax, __shfl_xor(amax, m, 64));
    nf |= __shfl_xor(nf, m, 64);
  }
  if (lane == 0) { red[wave] = amax; bad[wave] = nf; }
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  nf = bad[0] | bad[1] | bad[2] | bad[3];
  int e = 15;
  if (amax > 0.f) (void)frexpf(amax, &e);      // amax = m 2^e, m in [0.5, 1)
  int sh = 15 - e;                              // scaled maximum in [2^14, 2^15)
  if (sh > 30 || sh < -30) { nf = 1; sh = sh > 30 ? 30 : -30; }
  const float scale = ldexpf(1.f, sh);
  _Float16* orow = out + row * L;
  for (int64_t i = threadIdx.x; i < n4; i += 256) {
    const float4 v = xr[i];
    union { _Float16 h[4]; uint2 u; } o;
    o.h[0] = (_Float16)(v.x * scale); o.h[1] = (_Float16)(v.y * scale);
    o.h[2] = (_Float16)(v.z * scale); o.h[3] = (_Float16)(v.w * scale);
    reinterpret_cast<uint2*>(orow)[i] = o.u;
  }
  if (threadIdx.x == 0) {
    inv_out[row] = (inv_in ? inv_in[row] : 1.f) * ldexpf(1.f, -sh);
    if (nf) atomicAdd(&stats[0], 1ull);
  }
}

// ---- candidates of a query: columns with approximate score >= (approximate k-th best) - margin, in column order.
// One wave per query.  Slot s of query q lives at list s / k, entry s % k of the [n_lists][nq][k] layout that the
// top-k merge kernel reads.  count[q] may exceed cap (overflow: that query is redone by the exact path).
__global__ __launch_bounds__(256) void filter_collect_kernel(const float* __restrict__ S, int64_t nq, int64_t ncols, int64_t ld,
                                                             const int64_t* __restrict__ aidx, const float* __restrict__ aval,
                                                             int k, float margin, int cap, int64_t col_offset, int first,
                                                             int64_t* __restrict__ cand_idx, int* __restrict__ count,
                                                             unsigned long long* __restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= nq) return;
  const bool full = aidx[q * k + k - 1] >= 0;                       // fewer than k columns: everything is a candidate
  const float thr = full ? aval[q * k + k - 1] - margin : -INFINITY;
  // database panels after the first append to the query's list; the threshold is the k-th best seen SO FAR, which can only
  // be lower than the final one, so the test stays a superset test
  int base = first ? 0 : count[q];
  const int base0 = base;
  for (int64_t c0 = 0; c0 < ncols; c0 += 64) {
    const int64_t c = c0 + lane;
    const bool pred = c < ncols && S[q * ld + c] >= thr;
    const unsigned long long mask = __ballot(pred);
    if (pred) {
      const int slot = base + __popcll(mask & ((1ull << lane) - 1ull));
      if (slot < cap) cand_idx[((int64_t)(slot / k) * nq + q) * k + slot % k] = col_offset + c;
    }
    base += __popcll(mask);
  }
  if (lane == 0) {
    count[q] = base;
    atomicAdd(&stats[2], (unsigned long long)(base - base0));
    if (base > cap && base0 <= cap) atomicAdd(&stats[1], 1ull);   // counted once per query
  }
}

// ---- exact re-scoring: workgroup = one query, FLT_PASS candidates at a time; lane (candidate, c) runs the fma chain of
// the k-chunks c, c + 32, ... exactly as the f32 GEMM kernel does (see the header), chunk sums added in chunk order.
struct RescoreArgs {
  const float* Q;
  const float* DB;
  int64_t nq, L;
  const float* invq;
  const float* invdb;
  const int64_t* cand_idx;  // [n_lists][nq][k]
  float* cand_val;
  const int* count;
  int cap, k;
  int64_t col_offset;
  int64_t dense_n, dense_ld;   // DENSE: every database row is a "candidate": workgroup (g, q) scores rows [16 g, 16 g + 16) of query q
};                             //        into cand_val[q * dense_ld + row] (cand_idx, count, cap, k unused)

// DENSE: the same chains for ALL rows of the database, one or two queries at a time -- a single query against a resident index is
// a 1 x N row of scores: through the 128 x 128 MFMA tiles that is 127 wasted rows per tile (0.44 ms of matrix work at 8189 x 32768
// for 1.07 GB of reads); here it is one pass over the database at HBM speed, and the scores are the MFMA kernel's, bit for bit.
template <bool DENSE>
__global__ __launch_bounds__(FLT_THREADS, 4) void filter_rescore_kernel(RescoreArgs a) {
  __shared__ float a_s[32][FLT_SK + 1];
  __shared__ float b_s[FLT_PASS][32][FLT_SK + 1];
  __shared__ float csum[FLT_PASS][32];
  const int tid = threadIdx.x;
  const int cand = tid >> 5, c = tid & 31;
  const int64_t q = DENSE ? blockIdx.y : blockIdx.x;
  const int64_t dense0 = DENSE ? (int64_t)blockIdx.x * FLT_PASS : 0;
  const int cnt = DENSE ? (int)min((int64_t)FLT_PASS, a.dense_n - dense0) : a.count[q];
  if (!DENSE && cnt > a.cap) return;   // overflow: the exact path redoes this query
  const int64_t L = a.L;
  const int64_t nchunks = (L + 1023) / 1024;
  const float* qrow = a.Q + q * L;
  for (int p0 = 0; p0 < cnt; p0 += FLT_PASS) {
    const int slot = p0 + cand;
    const bool valid = slot < cnt;
    const int64_t off = DENSE ? q * a.dense_ld + dense0 + slot : ((int64_t)(slot / a.k) * a.nq + q) * a.k + slot % a.k;
    const int64_t j = !valid ? 0 : (DENSE ? dense0 + slot : a.cand_idx[off] - a.col_offset);
    const int npass = min(FLT_PASS, cnt - p0);
    float tot = 0.f;
    // staging roles: thread t < 256 loads float4 #t of the query slab; every thread loads 8 float4 of the candidate slabs
    float4 ra, rb[8];
    const float* brow[8];   // candidate row of staging slot u (candidate 2u + (tid >> 8)); null = no such candidate
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int cd = 2 * u + (tid >> 8);
      brow[u] = nullptr;
      if (cd < npass) {
        const int sl = p0 + cd;
        brow[u] = a.DB + (DENSE ? dense0 + sl : a.cand_idx[((int64_t)(sl / a.k) * a.nq + q) * a.k + sl % a.k] - a.col_offset) * L;
      }
    }
    auto fetch = [&](int64_t r0, int s) {
      {
        const int cc = tid >> 3, w = tid & 7;
        const int64_t kk = (r0 + cc) * 1024 + (int64_t)s * FLT_SK + 4 * w;
        ra = (tid < 256 && kk < L) ? *reinterpret_cast<const float4*>(qrow + kk) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int rem = tid & 255, cc = rem >> 3, w = rem & 7;
        const int64_t kk = (r0 + cc) * 1024 + (int64_t)s * FLT_SK + 4 * w;
        rb[u] = (brow[u] != nullptr && kk < L) ? *reinterpret_cast<const float4*>(brow[u] + kk) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    auto stash = [&]() {
      if (tid < 256) {
        const int cc = tid >> 3, w = tid & 7;
        a_s[cc][4 * w] = ra.x; a_s[cc][4 * w + 1] = ra.y; a_s[cc][4 * w + 2] = ra.z; a_s[cc][4 * w + 3] = ra.w;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = tid + FLT_THREADS * u;
        const int cd = idx >> 8, rem = idx & 255, cc = rem >> 3, w = rem & 7;
        b_s[cd][cc][4 * w] = rb[u].x; b_s[cd][cc][4 * w + 1] = rb[u].y; b_s[cd][cc][4 * w + 2] = rb[u].z; b_s[cd][cc][4 * w + 3] = rb[u].w;
      }
    };
    for (int64_t r0 = 0; r0 < nchunks; r0 += 32) {
      float acc = 0.f;
      fetch(r0, 0);
      for (int s = 0; s < 1024 / FLT_SK; ++s) {
        __syncthreads();           // the previous slab has been consumed
        stash();
        __syncthreads();
        if (s + 1 < 1024 / FLT_SK) fetch(r0, s + 1);   // in flight during the chain below
#pragma unroll
        for (int b8 = 0; b8 < FLT_SK; b8 += 8)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc = fmaf(a_s[c][b8 + e], b_s[cand][c][b8 + e], acc);
            acc = fmaf(a_s[c][b8 + 4 + e], b_s[cand][c][b8 + 4 + e], acc);
          }
      }
      csum[cand][c] = acc;
      __syncthreads();
      if (c == 0) {
        const int lim = (int)min((int64_t)32, nchunks - r0);
        for (int cc = 0; cc < lim; ++cc) tot += csum[cand][cc];
      }
    }
    if (c == 0 && valid) {
      const float sa = a.invq ? a.invq[q] : 1.f, sb = a.invdb ? a.invdb[j] : 1.f;
      a.cand_val[off] = (0.f + tot) * (sa * sb);
    }
    __syncthreads();
  }
}

__global__ void filter_gather_rows_kernel(const float* __restrict__ src, const int64_t* __restrict__ rows, int64_t L, float* __restrict__ dst,
                                          const float* __restrict__ inv_src, float* __restrict__ inv_dst) {
  const int64_t r = blockIdx.x;
  const int64_t srow = rows[r];
  for (int64_t i = threadIdx.x; i < L; i += blockDim.x) dst[r * L + i] = src[srow * L + i];
  if (threadIdx.x == 0) inv_dst[r] = inv_src ? inv_src[srow] : 1.f;
}

__global__ void filter_scatter_lists_kernel(const int64_t* __restrict__ idx_src, const float* __restrict__ val_src,
                                            const int64_t* __restrict__ rows, int k, int64_t* __restrict__ idx_dst, float* __restrict__ val_dst) {
  const int64_t r = blockIdx.x;
  for (int i = threadIdx.x; i < k; i += blockDim.x) {
    idx_dst[rows[r] * k + i] = idx_src[r * k + i];
    val_dst[rows[r] * k + i] = val_src[r * k + i];
  }
}


// One or two queries against all N rows: scores[q][j] by the defined recurrence, one pass over the database (see filter_rescore_kernel<true>)
bool cosine_dense_rows_eligible(const float* Q, const float* DB, int64_t nq, int64_t L) {
  return nq >= 1 && nq <= 2 && L % 8 == 0 && L >= 8 && L <= (int64_t)8 * 1024 * 1024 && reinterpret_cast<uintptr_t>(Q) % 16 == 0 &&
         reinterpret_cast<uintptr_t>(DB) % 16 == 0;
}
int launch_cosine_dense_rows(pvs_ctx* ctx, const float* Q, int64_t nq, const float* DB, int64_t N, int64_t L, const float* invq,
                             const float* invdb, float* scores, int64_t ld) {
  if (!cosine_dense_rows_eligible(Q, DB, nq, L)) PVS_FAIL(PVS_ERR_UNSUPPORTED, "dense row scoring: shape does not qualify");
  if (N <= 0) return PVS_OK;
  const int64_t groups = (N + FLT_PASS - 1) / FLT_PASS;
  if (groups > 0x7fffffffLL) PVS_FAIL(PVS_ERR_UNSUPPORTED, "dense row scoring: too many rows");
  RescoreArgs ra{Q, DB, nq, L, invq, invdb, nullptr, scores, nullptr, 0, 1, 0, N, ld};
  ScopedTimer tm(ctx, T_GEMM);
  hipLaunchKernelGGL(filter_rescore_kernel<true>, dim3((unsigned)groups, (unsigned)nq), dim3(FLT_THREADS), 0, ctx->stream, ra);
  PVS_HIP(hipGetLastError());
  return PVS_OK;
}

int launch_cosine_topk_filtered(pvs_ctx* ctx, const float* Q, int64_t nq, const float* DB, int64_t N, int64_t L,
                                const float* invq, const float* invdb, int k, int64_t col_offset, int64_t* d_idx, float* d_val,
                                int64_t* h_stats) {
  // qualify: the exact re-scoring reproduces the f32 MFMA kernel, so the exact path must itself take that kernel
  if (col_offset != 0) PVS_FAIL(PVS_ERR_UNSUPPORTED, "filtered top-k: col_offset must be 0");
  if (nq <= 0 || N <= 0) PVS_FAIL(PVS_ERR_UNSUPPORTED, "filtered top-k: empty input");
  if (L > (int64_t)8 * 1024 * 1024) PVS_FAIL(PVS_ERR_UNSUPPORTED, "filtered top-k: rows too long");
  if (L % 8 != 0 || L < 8 || reinterpret_cast<uintptr_t>(Q) % 16 || reinterpret_cast<uintptr_t>(DB) % 16)
    PVS_FAIL(PVS_ERR_UNSUPPORTED, "filtered top-k needs 16-B aligned rows with L %% 8 == 0");
  if (k < 1 || k > 128) PVS_FAIL(PVS_ERR_UNSUPPORTED, "filtered top-k: k <= 128");
  const int64_t NC = std::min<int64_t>(N, 32768);       // database rows per score panel
  const bool multi = N > NC;
  const bool same = (Q == DB) && (nq == N) && (invq == invdb);
  const bool square = same && !multi && (int64_t)nq * N <= ((int64_t)1 << 28);   // one symmetric launch covers everything
  // candidate slots: a multiple of k; over several database panels the running threshold admits more columns early on
  const int cap_lists = ((multi ? 16 : 4) * k + (multi ? 128 : 64) + k - 1) / k;
  const int cap = cap_lists * k;
  const int64_t QT = square ? nq : std::min<int64_t>(nq, std::max<int64_t>(256, ((int64_t)1 << 28) / NC));
  const double eps = filter_eps(L);
  const float margin = (float)(2.0 * eps);

  // ---- workspace
  const size_t q16_b = ((size_t)nq * L * 2 + 255) / 256 * 256, db16_b = same ? 0 : ((size_t)N * L * 2 + 255) / 256 * 256;
  char* w5 = nullptr;
  if (ws_reserve(ctx, 5, q16_b + db16_b, reinterpret_cast<void**>(&w5)) != PVS_OK) {
    (void)hipGetLastError();   // no room for the fp16 copies: the plain exact path needs none
    PVS_FAIL(PVS_ERR_UNSUPPORTED, "filtered top-k: no device memory for the fp16 copies (%zu bytes)", q16_b + db16_b);
  }
  _Float16* q16 = reinterpret_cast<_Float16*>(w5);
  _Float16* db16 = same ? q16 : reinterpret_cast<_Float16*>(w5 + q16_b);
  auto al = [](size_t b) { return (b + 255) / 256 * 256; };
  const size_t o_invq = 0, o_invd = o_invq + al((size_t)nq * 4), o_stats = o_invd + al((size_t)N * 4),
               o_aidx = o_stats + 256, o_aval = o_aidx + al((size_t)QT * k * 8), o_cidx = o_aval + al((size_t)QT * k * 4),
               o_cval = o_cidx + al((size_t)QT * cap * 8), o_cnt = o_cval + al((size_t)QT * cap * 4), o_end = o_cnt + al((size_t)QT * 4);
  char* w6 = nullptr;
  PVS_TRY(ws_reserve(ctx, 6, o_end, reinterpret_cast<void**>(&w6)));
  float* invq16 = reinterpret_cast<float*>(w6 + o_invq);
  float* invd16 = same ? invq16 : reinterpret_cast<float*>(w6 + o_invd);
  unsigned long long* stats = reinterpret_cast<unsigned long long*>(w6 + o_stats);
  int64_t* aidx = reinterpret_cast<int64_t*>(w6 + o_aidx);
  float* aval = reinterpret_cast<float*>(w6 + o_aval);
  int64_t* cidx = reinterpret_cast<int64_t*>(w6 + o_cidx);
  float* cval = reinterpret_cast<float*>(w6 + o_cval);
  int* cnt = reinterpret_cast<int*>(w6 + o_cnt);
  float* panel = nullptr;
  PVS_TRY(ws_reserve(ctx, 2, (size_t)QT * NC * sizeof(float), reinterpret_cast<void**>(&panel)));
  PVS_HIP(hipMemsetAsync(stats, 0, 32, ctx->stream));

  // ---- 1. scaled fp16 rows
  {
    ScopedTimer tm(ctx, T_MISC);
    hipLaunchKernelGGL(filter_rows_to_f16_kernel, dim3((unsigned)nq), dim3(256), 0, ctx->stream, Q, nq, L, invq, q16, invq16, stats);
    if (!same)
      hipLaunchKernelGGL(filter_rows_to_f16_kernel, dim3((unsigned)N), dim3(256), 0, ctx->stream, DB, N, L, invdb, db16, invd16, stats);
    PVS_HIP(hipGetLastError());
  }
  unsigned long long hs[4] = {0, 0, 0, 0};
  PVS_HIP(hipMemcpyAsync(hs, stats, 8, hipMemcpyDeviceToHost, ctx->stream));
  PVS_HIP(hipStreamSynchronize(ctx->stream));
  if (hs[0]) PVS_FAIL(PVS_ERR_UNSUPPORTED, "filtered top-k: %llu rows hold non-finite values or an extreme dynamic range", hs[0]);

  std::vector<int> h_cnt;
  for (int64_t q0 = 0; q0 < nq; q0 += QT) {
    const int64_t qn = std::min(QT, nq - q0);
    PVS_TRY(ws_reserve(ctx, 2, (size_t)QT * NC * sizeof(float), reinterpret_cast<void**>(&panel)));   // (the exact fallback shares the slot)
    PVS_HIP(hipMemsetAsync(cidx, 0xff, (size_t)qn * cap * 8, ctx->stream));
    for (int64_t c0 = 0; c0 < N; c0 += NC) {
      const int64_t cn = std::min(NC, N - c0);
      // ---- 2. bounded-error approximate scores, 3. approximate k-th best (running over the database panels)
      PVS_TRY(launch_cosine_f16_bounded(ctx, q16 + q0 * L, qn, db16 + c0 * L, cn, L, invq16 + q0, invd16 + c0, panel, cn));
      PVS_TRY(launch_topk(ctx, panel, qn, cn, cn, k, c0, c0 > 0 ? 1 : 0, aidx, aval));
      // ---- 4. candidates of this panel
      ScopedTimer tm(ctx, T_TOPK);
      hipLaunchKernelGGL(filter_collect_kernel, dim3((unsigned)((qn + 3) / 4)), dim3(256), 0, ctx->stream, panel, qn, cn, cn, aidx, aval, k,
                         margin, cap, c0, c0 == 0 ? 1 : 0, cidx, cnt, stats);
    }
    // scores too crowded for the filter (more than a tenth of the queries have more candidates than slots, e.g. nearly
    // orthogonal rows whose best scores sit inside the error margin of the bulk): the all-pairs path is the faster exact one
    PVS_HIP(hipMemcpyAsync(hs, stats, 24, hipMemcpyDeviceToHost, ctx->stream));
    PVS_HIP(hipStreamSynchronize(ctx->stream));
    if (hs[1] * 10 > (unsigned long long)qn) PVS_FAIL(PVS_ERR_UNSUPPORTED, "filtered top-k: %llu of %lld queries exceed the candidate slots", hs[1], (long long)qn);
    // ---- 5. exact scores of the candidates
    {
      ScopedTimer tm(ctx, T_RESCORE);
      RescoreArgs ra{Q + q0 * L, DB, qn, L, invq ? invq + q0 : nullptr, invdb, cidx, cval, cnt, cap, k, 0};
      hipLaunchKernelGGL(filter_rescore_kernel<false>, dim3((unsigned)qn), dim3(FLT_THREADS), 0, ctx->stream, ra);
    }
    PVS_HIP(hipGetLastError());
    // ---- 6. rank the candidates (same keys as the exact path); database indices get the caller's offset afterwards
    PVS_TRY(launch_topk_merge(ctx, cidx, cval, cap_lists, qn, k, d_idx + q0 * k, d_val + q0 * k));
    // ---- 7. queries with more candidates than slots: exact path
    PVS_HIP(hipMemcpyAsync(hs, stats, 24, hipMemcpyDeviceToHost, ctx->stream));
    PVS_HIP(hipStreamSynchronize(ctx->stream));
    if (hs[1]) {
      h_cnt.resize((size_t)qn);
      PVS_HIP(hipMemcpy(h_cnt.data(), cnt, (size_t)qn * 4, hipMemcpyDeviceToHost));
      std::vector<int64_t> rows;
      for (int64_t i = 0; i < qn; ++i)
        if (h_cnt[(size_t)i] > cap) rows.push_back(i);
      const int64_t nr = (int64_t)rows.size();
      const int64_t rb = std::max<int64_t>(1, std::min<int64_t>(nr, ((int64_t)1 << 28) / N));   // rows per exact batch (score rows <= 1 GiB)
      char* w4 = nullptr;
      const size_t rows_b = al((size_t)nr * 8), g_b = al((size_t)rb * L * 4), gi_b = al((size_t)rb * 4), li_b = al((size_t)rb * k * 8);
      PVS_TRY(ws_reserve(ctx, 3, rows_b + g_b + gi_b + li_b + al((size_t)rb * k * 4), reinterpret_cast<void**>(&w4)));
      int64_t* d_rows = reinterpret_cast<int64_t*>(w4);
      float* gq = reinterpret_cast<float*>(w4 + rows_b);
      float* gi = reinterpret_cast<float*>(w4 + rows_b + g_b);
      int64_t* li = reinterpret_cast<int64_t*>(w4 + rows_b + g_b + gi_b);
      float* lv = reinterpret_cast<float*>(w4 + rows_b + g_b + gi_b + li_b);
      PVS_HIP(hipMemcpyAsync(d_rows, rows.data(), (size_t)nr * 8, hipMemcpyHostToDevice, ctx->stream));
      for (int64_t r0 = 0; r0 < nr; r0 += rb) {
        const int64_t rn = std::min(rb, nr - r0);
        hipLaunchKernelGGL(filter_gather_rows_kernel, dim3((unsigned)rn), dim3(256), 0, ctx->stream, Q + q0 * L, d_rows + r0, L, gq,
                           invq ? invq + q0 : nullptr, gi);
        PVS_HIP(hipGetLastError());
        PVS_TRY(cosine_topk_exact(ctx, gq, rn, DB, N, L, gi, invdb, k, li, lv));   // panel + select over all database rows
        hipLaunchKernelGGL(filter_scatter_lists_kernel, dim3((unsigned)rn), dim3(64), 0, ctx->stream, li, lv, d_rows + r0, k,
                           d_idx + q0 * k, d_val + q0 * k);
      }
      PVS_HIP(hipGetLastError());
      PVS_HIP(hipStreamSynchronize(ctx->stream));   // `rows` is host memory going out of scope
      PVS_HIP(hipMemsetAsync(stats + 1, 0, 8, ctx->stream));
      if (h_stats) h_stats[1] += nr;
    }
    if (h_stats) h_stats[2] = (int64_t)hs[2];
  }
  if (h_stats) { h_stats[0] = 1; h_stats[3] = (int64_t)cap; }
  return PVS_OK;
}

}  // namespace pvs
