// K7 top-k: per query row, the first k entries of np.argsort(-scores)  (pyvisim/eval.py:37-43,75-80,131-132).
//
// Order is total and deterministic: (score descending, index ascending); NaN scores rank last (as NumPy's
// sort puts NaN at the end of -scores).  The reference uses a non-stable argsort, so its tie order is
// unspecified; on tie-free rows the lists are identical.
//
// One workgroup per row.  Every candidate becomes a 64-bit key  mono(score) << 32 | ~index  (all keys of a
// row are distinct, larger key = better).  A panel is consumed in chunks of 8192 columns held in registers
// together with the running list; the k best are found by an MSB-first 8-bit radix SELECT (LDS histogram,
// early exit when the pivot bin is taken whole), compacted, and only the final list is sorted (bitonic).
#include <algorithm>
#include <cstdlib>

#include "common.hpp"

namespace pvs {

constexpr int TK_THREADS = 256;
constexpr int TK_ITEMS = 32;                      // columns per thread per chunk
constexpr int TK_CHUNK = TK_THREADS * TK_ITEMS;   // 8192
constexpr int TK_KMAX = 1024;
constexpr int TK_RUN = TK_KMAX / TK_THREADS;      // running-list keys per thread

__device__ __forceinline__ uint32_t mono_f32(float f) {
  if (f != f) return 1u;                           // NaN: below every number (-inf maps to 0x007fffff)
  f += 0.0f;                                       // -0 -> +0 (equal scores must tie)
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unmono_f32(uint32_t m) {
  if (m == 1u) return __uint_as_float(0x7fc00000u);
  return __uint_as_float((m & 0x80000000u) ? (m & 0x7fffffffu) : ~m);
}
__device__ __forceinline__ uint64_t make_key(float v, uint32_t idx) {
  return ((uint64_t)mono_f32(v) << 32) | (uint64_t)(~idx);
}

struct TopkArgs {
  const float* scores;      // panel mode: [nq][ld];   list mode: val lists [n_lists][nq][k]
  const int64_t* idx_lists; // list mode only: [n_lists][nq][k]
  int64_t nq, ncols, ld;
  int k;
  int64_t col_offset;
  int merge;
  int n_lists;              // > 0 selects list mode (ncols = n_lists * k)
  int64_t* idx;             // [nq][out_ld]; this launch writes columns [out_off, out_off + k)
  float* val;
  int64_t out_ld;
  int out_off;              // > 0: paging -- only candidates strictly worse than entry out_off-1 are eligible
};

// block-wide inclusive scan of one int per thread (256 threads); tmp: LDS int[4]
__device__ __forceinline__ int block_incl_scan(int v, int* tmp, int lane, int wave) {
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(incl, d, 64);
    if (lane >= d) incl += o;
  }
  if (lane == 63) tmp[wave] = incl;
  __syncthreads();
  int off = 0;
  for (int w = 0; w < wave; ++w) off += tmp[w];
  __syncthreads();
  return incl + off;
}

__global__ __launch_bounds__(TK_THREADS) void topk_kernel(TopkArgs a) {
  __shared__ uint64_t run[2][TK_KMAX];
  __shared__ int hist[256];
  __shared__ int stmp[4];
  __shared__ int s_bin, s_above, s_cnt;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t q = blockIdx.x;
  const int k = a.k;
  int cur = 0;
  int run_count = 0;
  int64_t* const oidx = a.idx + q * a.out_ld + a.out_off;
  float* const oval = a.val + q * a.out_ld + a.out_off;
  // paging (ranking deeper than TK_KMAX): everything at or above the previous page's last key is already placed
  uint64_t upper = ~0ull;
  if (a.out_off > 0) {
    const int64_t pid = oidx[-1];
    upper = pid >= 0 ? make_key(oval[-1], (uint32_t)pid) : 0ull;   // previous page not full: nothing is left
  }

  // ---- running list from a previous panel
  if (a.merge) {
    int mine = 0;
    for (int r = tid; r < k; r += TK_THREADS) mine += oidx[r] >= 0;
    const int incl = block_incl_scan(mine, stmp, lane, wave);
    int pos = incl - mine;
    for (int r = tid; r < k; r += TK_THREADS) {
      const int64_t id = oidx[r];
      if (id >= 0) run[cur][pos++] = make_key(oval[r], (uint32_t)id);
    }
    if (tid == TK_THREADS - 1) s_cnt = incl;
    __syncthreads();
    run_count = s_cnt;
    __syncthreads();
  }

  for (int64_t c0 = 0; c0 < a.ncols; c0 += TK_CHUNK) {
    uint64_t key[TK_ITEMS + TK_RUN];
    int nvalid = 0;
#pragma unroll
    for (int it = 0; it < TK_ITEMS; ++it) {
      const int64_t c = c0 + (int64_t)it * TK_THREADS + tid;
      key[it] = 0ull;
      if (c < a.ncols) {
        if (a.n_lists > 0) {
          const int64_t l = c / k, r = c - l * k;
          const int64_t off = (l * a.nq + q) * k + r;
          const int64_t id = a.idx_lists[off];
          if (id >= 0) { key[it] = make_key(a.scores[off], (uint32_t)id); ++nvalid; }
        } else {
          const uint64_t kk = make_key(a.scores[q * a.ld + c], (uint32_t)(a.col_offset + c));
          if (kk < upper) { key[it] = kk; ++nvalid; }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < TK_RUN; ++r) {
      const int p = r * TK_THREADS + tid;
      key[TK_ITEMS + r] = p < run_count ? run[cur][p] : 0ull;
    }
    const int total = block_incl_scan(nvalid, stmp, lane, wave);  // inclusive; last thread holds the sum
    if (tid == TK_THREADS - 1) s_cnt = total + run_count;
    __syncthreads();
    const int T = s_cnt;
    __syncthreads();

    uint64_t prefix = 0ull, mask = 0ull;
    if (T > k) {
      int need = k;
      for (int pass = 0; pass < 8; ++pass) {
        const int shift = 56 - 8 * pass;
        hist[tid] = 0;
        __syncthreads();
#pragma unroll
        for (int it = 0; it < TK_ITEMS + TK_RUN; ++it)
          if ((key[it] & mask) == prefix && key[it] != 0ull) atomicAdd(&hist[(int)((key[it] >> shift) & 0xffull)], 1);
        __syncthreads();
        const int hv = hist[255 - tid];                  // thread t owns bin 255-t: scan from the top bin down
        const int incl = block_incl_scan(hv, stmp, lane, wave);
        if (incl >= need && incl - hv < need) { s_bin = 255 - tid; s_above = incl - hv; }
        __syncthreads();
        const int bin = s_bin;
        need -= s_above;
        prefix |= (uint64_t)bin << shift;
        mask |= 0xffull << shift;
        const bool whole = hist[bin] == need;            // pivot bin is taken whole: selection is decided
        __syncthreads();
        if (whole) break;
      }
    }
    // ---- compact the selected keys into the other buffer (T <= k: everything valid is selected)
    if (tid == 0) s_cnt = 0;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < TK_ITEMS + TK_RUN; ++it) {
      if (key[it] != 0ull && (key[it] & mask) >= prefix) {
        const int p = atomicAdd(&s_cnt, 1);
        if (p < TK_KMAX) run[cur ^ 1][p] = key[it];
      }
    }
    __syncthreads();
    run_count = min(s_cnt, TK_KMAX);  // == k unless the inputs held duplicate (score, index) pairs
    cur ^= 1;
    __syncthreads();
  }

  // ---- sort the final list descending (bitonic, padded with 0 = worst)
  int P = 1;
  while (P < run_count) P <<= 1;
  for (int p = run_count + tid; p < P; p += TK_THREADS) run[cur][p] = 0ull;
  __syncthreads();
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (P >> 1); t += TK_THREADS) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool desc = ((lo & size) == 0);
        const uint64_t x = run[cur][lo], y = run[cur][hi];
        if ((x < y) == desc) { run[cur][lo] = y; run[cur][hi] = x; }
      }
      __syncthreads();
    }
  }
  for (int r = tid; r < k; r += TK_THREADS) {
    if (r < run_count) {  // run_count may exceed k only with duplicate inputs; the sort keeps the best first
      const uint64_t kk = run[cur][r];
      oidx[r] = (int64_t)(uint32_t)(~(uint32_t)(kk & 0xffffffffull));
      oval[r] = unmono_f32((uint32_t)(kk >> 32));
    } else {
      oidx[r] = -1;
      oval[r] = -INFINITY;
    }
  }
}


// Small k (<= 16), panel mode: k rounds of "take the best remaining key of the row".  Every thread keeps the best of its
// keys in a register; a round is one wave butterfly + a four-way LDS merge, and only the thread that owned the winner rescans
// its 32 keys.  Same keys, same total order as the radix select above (so the lists are identical), a third of its time at
// k = 5: the select's histogram passes and block scans cost more than ranking 8192 keys five times.
constexpr int TKS_KMAX = 16;

__global__ __launch_bounds__(TK_THREADS) void topk_small_kernel(TopkArgs a) {
  __shared__ uint64_t wbest[4];
  __shared__ uint64_t run[TKS_KMAX];
  __shared__ uint64_t outk[TKS_KMAX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t q = blockIdx.x;
  const int k = a.k;
  int64_t* const oidx = a.idx + q * a.out_ld;
  float* const oval = a.val + q * a.out_ld;
  int run_count = 0;
  if (a.merge) {   // running list from earlier panels: extra candidates of the first chunk
    if (tid < k) run[tid] = oidx[tid] >= 0 ? make_key(oval[tid], (uint32_t)oidx[tid]) : 0ull;
    run_count = k;
  }
  __syncthreads();
  for (int64_t c0 = 0; c0 < a.ncols || (c0 == 0 && run_count > 0); c0 += TK_CHUNK) {
    uint64_t key[TK_ITEMS];
#pragma unroll
    for (int it = 0; it < TK_ITEMS; ++it) {
      const int64_t c = c0 + (int64_t)it * TK_THREADS + tid;
      key[it] = c < a.ncols ? make_key(a.scores[q * a.ld + c], (uint32_t)(a.col_offset + c)) : 0ull;
    }
    uint64_t extra = tid < run_count ? run[tid] : 0ull;   // key 0 = "nothing" (every real key is > 0)
    __syncthreads();                                       // run[] has been read by everyone
    for (int r = 0; r < k; ++r) {
      uint64_t mine = extra;
#pragma unroll
      for (int it = 0; it < TK_ITEMS; ++it) mine = key[it] > mine ? key[it] : mine;
      uint64_t best = mine;
      for (int m = 32; m >= 1; m >>= 1) {
        const uint64_t o = __shfl_xor(best, m, 64);
        best = o > best ? o : best;
      }
      if (lane == 0) wbest[wave] = best;
      __syncthreads();
      uint64_t b = wbest[0];
      b = wbest[1] > b ? wbest[1] : b;
      b = wbest[2] > b ? wbest[2] : b;
      b = wbest[3] > b ? wbest[3] : b;
      if (tid == 0) outk[r] = b;
      if (b != 0ull && mine == b) {          // keys are distinct: exactly one thread owns the winner; it retires that key
        if (extra == b) extra = 0ull;
#pragma unroll
        for (int it = 0; it < TK_ITEMS; ++it) key[it] = key[it] == b ? 0ull : key[it];
      }
      __syncthreads();                        // wbest is rewritten next round
    }
    if (tid < k) run[tid] = outk[tid];        // descending: the list so far
    run_count = k;
    __syncthreads();
  }
  if (tid < k) {
    const uint64_t kk = run_count > 0 ? run[tid] : 0ull;
    if (kk != 0ull) {
      oidx[tid] = (int64_t)(~(uint32_t)kk);
      oval[tid] = unmono_f32((uint32_t)(kk >> 32));
    } else {
      oidx[tid] = -1;
      oval[tid] = -INFINITY;
    }
  }
}

// Small k (<= 16), panels of many columns: THRESHOLD FILTER in front of the ranking, ONE WAVE PER ROW (no workgroup barrier).
// Once k keys are known to exist, their k-th best score T is a lower bound of the final k-th best, so a column can only
// matter if its score is >= T (ties included: the 64-bit key decides among equal scores).  A chunk of 2048 columns is 32
// values per lane: one compare each, the few survivors are compacted (ballot + prefix count) into the wave's candidate list
// in LDS, and that list is ranked once at the end -- a bitonic sort across the 64 lanes, registers only.  T comes from the
// running list of the previous panels (its k-th entry) or, on a row's first chunk, from the chunk itself: the k-th largest
// of the 64 lane maxima is reached by k columns.  Rows whose scores are so tied that the survivors do not fit the list (an
// all-zero query) fall back to k rounds of "best remaining key" over the chunk.  Same keys, same total order as the two
// kernels above: the lists are theirs, bit for bit.  The rounds kernel is instruction-bound on 8192 x 32768 panels (1.2 ms per
// GB); this one streams the panel.
constexpr int TKW_WAVES = 4;                 // rows per workgroup
constexpr int TKW_LIST = 256;                // candidate list capacity per wave (keys)
constexpr int TKW_CHUNK = 64 * TK_ITEMS;     // 2048 columns

// descending bitonic sort across the 64 lanes of a wave (registers only): lane i ends up with the i-th largest value
template <typename T>
__device__ __forceinline__ T wave_sort_desc(T v, int lane) {
#pragma unroll
  for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const T o = __shfl_xor(v, stride, 64);
      const bool keep_max = ((lane & stride) == 0) == ((lane & size) == 0);
      v = keep_max ? (o > v ? o : v) : (o < v ? o : v);
    }
  }
  return v;
}
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
  for (int m = 32; m >= 1; m >>= 1) {
    const uint64_t o = __shfl_xor(v, m, 64);
    v = o > v ? o : v;
  }
  return v;
}

// VEC: rows are 16-B aligned (ld % 4 == 0): 16-B loads, lane l holds columns c0 + 4 (64 j + l) + e, j = 0..7, e = 0..3
template <bool VEC>
__global__ __launch_bounds__(64 * TKW_WAVES) void topk_wave_kernel(TopkArgs a) {
  __shared__ uint64_t cand_all[TKW_WAVES][TKW_LIST];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t q = (int64_t)blockIdx.x * TKW_WAVES + wave;
  if (q >= a.nq) return;                     // whole waves leave: nothing below synchronises across waves
  uint64_t* const cand = cand_all[wave];
  const int k = a.k;
  int64_t* const oidx = a.idx + q * a.out_ld;
  float* const oval = a.val + q * a.out_ld;
  const float* const row = a.scores + q * a.ld;

  // the list: cand[0..n), unsorted; T = a score (order-preserving 32-bit image) that at least k known keys reach, 0 = none yet
  int n = 0;
  uint32_t T = 0u;
  if (a.merge) {
    const uint64_t key = (lane < k && oidx[lane] >= 0) ? make_key(oval[lane], (uint32_t)oidx[lane]) : 0ull;
    const unsigned long long bm = __ballot(key != 0ull);
    if (key != 0ull) cand[__popcll(bm & ((1ull << lane) - 1ull))] = key;
    n = __popcll(bm);
    if (oidx[k - 1] >= 0) T = mono_f32(oval[k - 1]);   // the running list is sorted: its last entry is the k-th best so far
  }
  // list (n <= TKW_LIST) -> its best min(n, k) keys in cand[0..), sorted; tightens T
  auto list_topk = [&]() {
    uint64_t key[TKW_LIST / 64];
#pragma unroll
    for (int it = 0; it < TKW_LIST / 64; ++it) key[it] = (it * 64 + lane) < n ? cand[it * 64 + lane] : 0ull;
    uint64_t keep = 0ull;                    // lane r keeps the r-th winner
    for (int r = 0; r < k; ++r) {
      uint64_t mine = 0ull;
#pragma unroll
      for (int it = 0; it < TKW_LIST / 64; ++it) mine = key[it] > mine ? key[it] : mine;
      const uint64_t b = wave_max_u64(mine);
      if (lane == r) keep = b;
      if (b != 0ull && mine == b) {
#pragma unroll
        for (int it = 0; it < TKW_LIST / 64; ++it) key[it] = key[it] == b ? 0ull : key[it];
      }
    }
    if (lane < k) cand[lane] = keep;
    n = n < k ? n : k;
    if (n == k) T = (uint32_t)(__shfl(keep, k - 1, 64) >> 32);
  };

  for (int64_t c0 = 0; c0 < a.ncols; c0 += TKW_CHUNK) {
    uint32_t m[TK_ITEMS];     // order-preserving images of this lane's 32 scores (0 = no column)
    auto col = [&](int it) -> int64_t {      // column of element `it` (increasing in `it` for both layouts)
      return VEC ? c0 + ((int64_t)(it >> 2) * 64 + lane) * 4 + (it & 3) : c0 + (int64_t)it * 64 + lane;
    };
    if constexpr (VEC) {
#pragma unroll
      for (int j = 0; j < TK_ITEMS / 4; ++j) {
        const int64_t c = c0 + ((int64_t)j * 64 + lane) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c + 3 < a.ncols) v = *reinterpret_cast<const float4*>(row + c);
        else {
          if (c + 0 < a.ncols) v.x = row[c + 0];
          if (c + 1 < a.ncols) v.y = row[c + 1];
          if (c + 2 < a.ncols) v.z = row[c + 2];
        }
        m[4 * j + 0] = c + 0 < a.ncols ? mono_f32(v.x) : 0u;
        m[4 * j + 1] = c + 1 < a.ncols ? mono_f32(v.y) : 0u;
        m[4 * j + 2] = c + 2 < a.ncols ? mono_f32(v.z) : 0u;
        m[4 * j + 3] = c + 3 < a.ncols ? mono_f32(v.w) : 0u;
      }
    } else {
#pragma unroll
      for (int it = 0; it < TK_ITEMS; ++it) {
        const int64_t c = col(it);
        m[it] = c < a.ncols ? mono_f32(row[c]) : 0u;
      }
    }
    if (T == 0u) {
      // no threshold yet: the k-th largest of the 64 lane maxima is reached by k columns of this chunk (0: fewer than k columns)
      uint32_t mx = 0u;
#pragma unroll
      for (int it = 0; it < TK_ITEMS; ++it) mx = m[it] > mx ? m[it] : mx;
      T = __shfl(wave_sort_desc(mx, lane), k - 1, 64);
    }
    const uint32_t thr = T == 0u ? 1u : T;
    const int n0 = n;
    bool overflow = false;
#pragma unroll
    for (int it = 0; it < TK_ITEMS; ++it) {
      const bool pass = m[it] >= thr;
      const unsigned long long bm = __ballot(pass);
      if (bm != 0ull) {                      // wave-uniform
        const int cnt = __popcll(bm);
        if (n + cnt <= TKW_LIST) {
          if (pass) cand[n + __popcll(bm & ((1ull << lane) - 1ull))] = ((uint64_t)m[it] << 32) | (uint64_t)(~(uint32_t)(a.col_offset + col(it)));
        } else {
          overflow = true;
        }
        n += cnt;
      }
    }
    if (!overflow) {
      if (n > TKW_LIST / 2) list_topk();
      continue;
    }
    // too many survivors (heavily tied scores): drop this chunk's appends, keep the best k of the list, then k rounds of "best
    // remaining key" over the chunk's survivors and that list.  A lane's best is its largest score, the first (lowest column)
    // among equal ones.
    n = n0;
    list_topk();
    uint64_t lk = lane < n ? cand[lane] : 0ull;
    uint64_t keep = 0ull;
    for (int r = 0; r < k; ++r) {
      uint32_t mx = 0u;
      int at = -1;
#pragma unroll
      for (int it = 0; it < TK_ITEMS; ++it)
        if (m[it] >= thr && m[it] > mx) { mx = m[it]; at = it; }
      uint64_t mine = at >= 0 ? ((uint64_t)mx << 32) | (uint64_t)(~(uint32_t)(a.col_offset + col(at))) : 0ull;
      const bool from_list = lk > mine;
      mine = from_list ? lk : mine;
      const uint64_t b = wave_max_u64(mine);
      if (lane == r) keep = b;
      if (b != 0ull && mine == b) {
        if (from_list) lk = 0ull;
        else {
#pragma unroll
          for (int it = 0; it < TK_ITEMS; ++it) m[it] = it == at ? 0u : m[it];
        }
      }
    }
    if (lane < k) cand[lane] = keep;
    n = __popcll(__ballot(lane < k && keep != 0ull));
    if (n == k) T = (uint32_t)(__shfl(keep, k - 1, 64) >> 32);
  }
  // ---- rank the list: <= 64 keys by a bitonic sort across the lanes
  if (n > 64) list_topk();
  const uint64_t kk = wave_sort_desc(lane < n ? cand[lane] : 0ull, lane);
  if (lane < k) {
    if (kk != 0ull) {
      oidx[lane] = (int64_t)(~(uint32_t)kk);
      oval[lane] = unmono_f32((uint32_t)(kk >> 32));
    } else {
      oidx[lane] = -1;
      oval[lane] = -INFINITY;
    }
  }
}

static int launch_topk_impl(pvs_ctx* ctx, const TopkArgs& a) {
  if (a.nq <= 0) return PVS_OK;
  if (a.k < 1 || a.k > TK_KMAX) PVS_FAIL(PVS_ERR_UNSUPPORTED, "top-k: k must be in [1, %d] (got %d)", TK_KMAX, a.k);
  if (a.nq > 0x7fffffffLL) PVS_FAIL(PVS_ERR_UNSUPPORTED, "top-k: too many query rows for one launch");
  if (a.col_offset + a.ncols > 0xfffffffeLL) PVS_FAIL(PVS_ERR_UNSUPPORTED, "top-k: column index exceeds 32 bits");
  ScopedTimer tm(ctx, T_TOPK);
  const int variant = ctx->opt[PVS_OPT_TOPK_SELECT_ONLY];   // 0 chosen here, 1 radix select, 2 rounds, 3 threshold filter + rounds
  const bool small = a.k <= TKS_KMAX && a.n_lists == 0 && a.out_off == 0 && a.out_ld == a.k && variant != 1;
  if (small && (variant == 3 || (variant == 0 && a.ncols >= 4096))) {
    const bool vec = a.ld % 4 == 0 && reinterpret_cast<uintptr_t>(a.scores) % 16 == 0;
    const unsigned grid = (unsigned)((a.nq + TKW_WAVES - 1) / TKW_WAVES);
    if (vec) hipLaunchKernelGGL(topk_wave_kernel<true>, dim3(grid), dim3(64 * TKW_WAVES), 0, ctx->stream, a);
    else hipLaunchKernelGGL(topk_wave_kernel<false>, dim3(grid), dim3(64 * TKW_WAVES), 0, ctx->stream, a);
  }
  else if (small)
    hipLaunchKernelGGL(topk_small_kernel, dim3((unsigned)a.nq), dim3(TK_THREADS), 0, ctx->stream, a);
  else
    hipLaunchKernelGGL(topk_kernel, dim3((unsigned)a.nq), dim3(TK_THREADS), 0, ctx->stream, a);
  PVS_HIP(hipGetLastError());
  return PVS_OK;
}

int launch_topk(pvs_ctx* ctx, const float* scores, int64_t nq, int64_t ncols, int64_t ld, int k, int64_t col_offset,
                int merge, int64_t* d_idx, float* d_val) {
  if (k <= TK_KMAX) {
    TopkArgs a{scores, nullptr, nq, ncols, ld, k, col_offset, merge, 0, d_idx, d_val, k, 0};
    return launch_topk_impl(ctx, a);
  }
  // deep ranking (e.g. top_k_map(k=None) = full argsort): pages of TK_KMAX, each page selects the best keys that
  // are strictly worse than the previous page's last key.  Needs the whole score row in this panel.
  if (merge) PVS_FAIL(PVS_ERR_UNSUPPORTED, "top-k deeper than %d cannot merge across panels", TK_KMAX);
  for (int off = 0; off < k; off += TK_KMAX) {
    TopkArgs a{scores, nullptr, nq, ncols, ld, std::min(TK_KMAX, k - off), col_offset, 0, 0, d_idx, d_val, k, off};
    PVS_TRY(launch_topk_impl(ctx, a));
  }
  return PVS_OK;
}

int launch_topk_merge(pvs_ctx* ctx, const int64_t* idx_lists, const float* val_lists, int n_lists, int64_t nq, int k,
                      int64_t* d_idx, float* d_val) {
  if (n_lists < 1) PVS_FAIL(PVS_ERR_INVALID, "top-k merge: n_lists must be >= 1");
  if (k > TK_KMAX) PVS_FAIL(PVS_ERR_UNSUPPORTED, "top-k merge: k must be <= %d", TK_KMAX);
  TopkArgs a{val_lists, idx_lists, nq, (int64_t)n_lists * k, 0, k, 0, 0, n_lists, d_idx, d_val, k, 0};
  return launch_topk_impl(ctx, a);
}

// ------------------------------------------------------------------------------------------------- fp64 ranking
// The reference ranks in float64 whenever an operand is float64 (Fisher encodings; pyvisim/_utils.py:312-330 returns the
// dtype sklearn's cosine_similarity gives, eval.py then argsorts it).  Not a throughput path: one workgroup per query row,
// the row's keys (order-preserving 64-bit image of the score, NaN last, -0 = +0) sorted in LDS by a bitonic network with the
// index as tie break -- the same total order as the fp32 kernels: (score descending, index ascending).
constexpr int R64_THREADS = 256;
constexpr int R64_CHUNK = 8192;

__device__ __forceinline__ uint64_t mono_f64(double f) {
  if (f != f) return 1ull;
  f += 0.0;
  const uint64_t u = (uint64_t)__double_as_longlong(f);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double unmono_f64(uint64_t m) {
  if (m == 1ull) return __longlong_as_double(0x7ff8000000000000ll);
  return __longlong_as_double((long long)((m >> 63) ? (m & 0x7fffffffffffffffull) : ~m));
}

// sorts key[0..n2) / id[0..n2) (n2 a power of two) so that better entries come first
__device__ __forceinline__ void bitonic_desc(uint64_t* key, uint32_t* id, int n2) {
  for (int sz = 2; sz <= n2; sz <<= 1) {
    for (int st = sz >> 1; st >= 1; st >>= 1) {
      __syncthreads();
      for (int t = threadIdx.x; t < (n2 >> 1); t += R64_THREADS) {
        const int lo = ((t / st) * (st << 1)) + (t % st), hi = lo + st;
        const bool desc = ((lo & sz) == 0);            // direction of this bitonic block
        const uint64_t ka = key[lo], kb = key[hi];
        const uint32_t ia = id[lo], ib = id[hi];
        const bool a_first = ka > kb || (ka == kb && ia < ib);
        if (a_first != desc) { key[lo] = kb; key[hi] = ka; id[lo] = ib; id[hi] = ia; }
      }
    }
  }
  __syncthreads();
}

// One page of the ranking: the best `k` entries of the row that come strictly AFTER the entry at position off - 1 of the list
// already written (off = 0: no bound), stored at positions [off, off + k) of a list of length ktotal.  Deep rankings
// (top_k_map(k=None) is a full argsort in the reference, eval.py:78) page through the complete row R64_PAGE entries at a time.
__global__ __launch_bounds__(R64_THREADS) void rank_f64_kernel(const double* __restrict__ scores, int64_t ncols, int64_t ld, int k,
                                                               int ktotal, int off, int64_t* __restrict__ oidx,
                                                               double* __restrict__ oval) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t* key = reinterpret_cast<uint64_t*>(smem);               // [R64_CHUNK]
  uint32_t* id = reinterpret_cast<uint32_t*>(key + R64_CHUNK);     // [R64_CHUNK]
  const int64_t q = blockIdx.x;
  const double* row = scores + q * ld;
  uint64_t bkey = ~0ull;
  uint32_t bid = 0;
  const bool bounded = off > 0;
  if (bounded) {
    const int64_t pi = oidx[q * ktotal + off - 1];
    if (pi < 0) {                                                  // the previous page already ran out of columns
      for (int t = threadIdx.x; t < k; t += R64_THREADS) {
        oidx[q * ktotal + off + t] = -1;
        oval[q * ktotal + off + t] = -INFINITY;
      }
      return;
    }
    bid = (uint32_t)pi;
    bkey = mono_f64(row[pi]);
  }
  int have = 0;   // entries of the running list, kept sorted in key[0..have)
  for (int64_t c0 = 0; c0 < ncols;) {
    // the running list stays in front; the rest of the buffer takes the next columns
    const int room = R64_CHUNK - have;
    const int take = (int)((ncols - c0) < room ? (ncols - c0) : room);
    int n2 = 1;
    while (n2 < have + take) n2 <<= 1;
    for (int t = threadIdx.x; t < n2 - have; t += R64_THREADS) {
      bool in = t < take;
      uint64_t kk = 0ull;                                           // 0: below NaN's key, never selected before a real entry
      if (in) {
        kk = mono_f64(row[c0 + t]);
        // entries at or before the bound in the total order (key descending, index ascending) belong to earlier pages
        if (bounded && (kk > bkey || (kk == bkey && (uint32_t)(c0 + t) <= bid))) { in = false; kk = 0ull; }
      }
      key[have + t] = kk;
      id[have + t] = in ? (uint32_t)(c0 + t) : 0xffffffffu;
    }
    bitonic_desc(key, id, n2);
    const int tot = have + take;
    have = tot < k ? tot : k;
    c0 += take;
    __syncthreads();
  }
  for (int t = threadIdx.x; t < k; t += R64_THREADS) {
    const bool in = t < have && id[t] != 0xffffffffu;
    oidx[q * ktotal + off + t] = in ? (int64_t)id[t] : -1;
    oval[q * ktotal + off + t] = in ? unmono_f64(key[t]) : -INFINITY;
  }
}

// Shallow float64 rankings (k <= 16) of complete rows: the one-wave-per-row threshold filter of topk_wave_kernel on 64-bit score
// images -- a lane holds 16 scores of a 1024-column chunk, survivors of `score >= T` go to the wave's list as (image, column)
// pairs, the list is ranked at the end by a bitonic sort across the lanes with the column as tie break.  Same total order as
// rank_f64_kernel (score descending, index ascending, NaN last), so the same lists; that kernel sorts 8192 keys in LDS per
// row whatever k is (8.6 ms for 8189 x 8189 scores, k = 5).
constexpr int R64W_LIST = 256, R64W_ITEMS = 16, R64W_CHUNK = 64 * R64W_ITEMS;

struct Key96 {
  uint64_t m;
  uint32_t i;
};
__device__ __forceinline__ bool key96_better(const Key96& a, const Key96& b) { return a.m > b.m || (a.m == b.m && a.i < b.i); }
__device__ __forceinline__ Key96 wave_sort_desc96(Key96 v, int lane) {
#pragma unroll
  for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      Key96 o;
      o.m = __shfl_xor(v.m, stride, 64);
      o.i = __shfl_xor(v.i, stride, 64);
      const bool keep_best = ((lane & stride) == 0) == ((lane & size) == 0);
      const bool o_better = key96_better(o, v);
      if (keep_best == o_better) v = o;
    }
  }
  return v;
}

__global__ __launch_bounds__(64 * TKW_WAVES) void rank_f64_wave_kernel(const double* __restrict__ scores, int64_t nq, int64_t ncols, int64_t ld,
                                                                       int k, int64_t* __restrict__ oidx, double* __restrict__ oval) {
  __shared__ uint64_t cm_all[TKW_WAVES][R64W_LIST];
  __shared__ uint32_t ci_all[TKW_WAVES][R64W_LIST];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t q = (int64_t)blockIdx.x * TKW_WAVES + wave;
  if (q >= nq) return;
  uint64_t* const cm = cm_all[wave];
  uint32_t* const ci = ci_all[wave];
  const double* const row = scores + q * ld;
  int n = 0;
  uint64_t T = 0ull;   // 0: no threshold yet (every real image is >= 1)
  // list (n <= R64W_LIST) -> its best min(n, k) entries in cm / ci [0..), sorted; tightens T
  auto list_topk = [&]() {
    Key96 key[R64W_LIST / 64];
#pragma unroll
    for (int it = 0; it < R64W_LIST / 64; ++it) {
      const int p = it * 64 + lane;
      key[it].m = p < n ? cm[p] : 0ull;
      key[it].i = p < n ? ci[p] : 0xffffffffu;
    }
    Key96 keep{0ull, 0xffffffffu};
    for (int r = 0; r < k; ++r) {
      Key96 mine{0ull, 0xffffffffu};
#pragma unroll
      for (int it = 0; it < R64W_LIST / 64; ++it)
        if (key96_better(key[it], mine)) mine = key[it];
      Key96 b = mine;
      for (int sh = 32; sh >= 1; sh >>= 1) {
        Key96 o;
        o.m = __shfl_xor(b.m, sh, 64);
        o.i = __shfl_xor(b.i, sh, 64);
        if (key96_better(o, b)) b = o;
      }
      if (lane == r) keep = b;
      if (b.m != 0ull && mine.m == b.m && mine.i == b.i) {
#pragma unroll
        for (int it = 0; it < R64W_LIST / 64; ++it)
          if (key[it].m == b.m && key[it].i == b.i) key[it] = Key96{0ull, 0xffffffffu};
      }
    }
    if (lane < k) { cm[lane] = keep.m; ci[lane] = keep.i; }
    n = n < k ? n : k;
    if (n == k) T = __shfl(keep.m, k - 1, 64);
  };
  for (int64_t c0 = 0; c0 < ncols; c0 += R64W_CHUNK) {
    uint64_t m[R64W_ITEMS];
#pragma unroll
    for (int it = 0; it < R64W_ITEMS; ++it) {
      const int64_t c = c0 + (int64_t)it * 64 + lane;
      m[it] = c < ncols ? mono_f64(row[c]) : 0ull;
    }
    if (T == 0ull) {   // the k-th largest of the 64 lane maxima is reached by k columns of this chunk (0: fewer than k columns)
      uint64_t mx = 0ull;
#pragma unroll
      for (int it = 0; it < R64W_ITEMS; ++it) mx = m[it] > mx ? m[it] : mx;
      T = __shfl(wave_sort_desc(mx, lane), k - 1, 64);
    }
    const uint64_t thr = T == 0ull ? 1ull : T;
    const int n0 = n;
    bool overflow = false;
#pragma unroll
    for (int it = 0; it < R64W_ITEMS; ++it) {
      const bool pass = m[it] >= thr;
      const unsigned long long bm = __ballot(pass);
      if (bm != 0ull) {
        const int cnt = __popcll(bm);
        if (n + cnt <= R64W_LIST) {
          if (pass) {
            const int p = n + __popcll(bm & ((1ull << lane) - 1ull));
            cm[p] = m[it];
            ci[p] = (uint32_t)(c0 + (int64_t)it * 64 + lane);
          }
        } else {
          overflow = true;
        }
        n += cnt;
      }
    }
    if (!overflow) {
      if (n > R64W_LIST / 2) list_topk();
      continue;
    }
    // heavily tied scores: best k of the list, then k rounds over the chunk's survivors and that list
    n = n0;
    list_topk();
    Key96 lk{lane < n ? cm[lane] : 0ull, lane < n ? ci[lane] : 0xffffffffu};
    Key96 keep{0ull, 0xffffffffu};
    for (int r = 0; r < k; ++r) {
      Key96 mine{0ull, 0xffffffffu};
      int at = -1;
#pragma unroll
      for (int it = 0; it < R64W_ITEMS; ++it)
        if (m[it] >= thr && m[it] > mine.m) { mine.m = m[it]; at = it; }   // strict '>': the lowest column among equal scores
      if (at >= 0) mine.i = (uint32_t)(c0 + (int64_t)at * 64 + lane);
      const bool from_list = key96_better(lk, mine);
      if (from_list) mine = lk;
      Key96 b = mine;
      for (int sh = 32; sh >= 1; sh >>= 1) {
        Key96 o;
        o.m = __shfl_xor(b.m, sh, 64);
        o.i = __shfl_xor(b.i, sh, 64);
        if (key96_better(o, b)) b = o;
      }
      if (lane == r) keep = b;
      if (b.m != 0ull && mine.m == b.m && mine.i == b.i) {
        if (from_list) lk = Key96{0ull, 0xffffffffu};
        else {
#pragma unroll
          for (int it = 0; it < R64W_ITEMS; ++it) m[it] = it == at ? 0ull : m[it];
        }
      }
    }
    if (lane < k) { cm[lane] = keep.m; ci[lane] = keep.i; }
    n = __popcll(__ballot(lane < k && keep.m != 0ull));
    if (n == k) T = __shfl(keep.m, k - 1, 64);
  }
  if (n > 64) list_topk();
  const Key96 kk = wave_sort_desc96(Key96{lane < n ? cm[lane] : 0ull, lane < n ? ci[lane] : 0xffffffffu}, lane);
  if (lane < k) {
    const bool in = kk.m != 0ull;
    oidx[q * k + lane] = in ? (int64_t)kk.i : -1;
    oval[q * k + lane] = in ? unmono_f64(kk.m) : -INFINITY;
  }
}

constexpr int R64_PAGE = R64_CHUNK / 2;

int launch_rank_f64(pvs_ctx* ctx, const double* scores, int64_t nq, int64_t ncols, int64_t ld, int k, int64_t* d_idx, double* d_val) {
  if (nq <= 0 || k <= 0) return PVS_OK;
  if (ncols >= ((int64_t)1 << 32) - 1) PVS_FAIL(PVS_ERR_UNSUPPORTED, "float64 ranking: too many columns");
  const size_t lds = (size_t)R64_CHUNK * 12;
  PVS_TRY(ensure_lds(ctx, reinterpret_cast<const void*>(rank_f64_kernel), lds));
  ScopedTimer tm(ctx, T_TOPK);
  if (k <= TKS_KMAX && ncols >= 2 * R64W_CHUNK && ctx->opt[PVS_OPT_TOPK_SELECT_ONLY] != 1) {
    hipLaunchKernelGGL(rank_f64_wave_kernel, dim3((unsigned)((nq + TKW_WAVES - 1) / TKW_WAVES)), dim3(64 * TKW_WAVES), 0, ctx->stream, scores, nq,
                       ncols, ld, k, d_idx, d_val);
  } else if (k <= R64_PAGE || ncols <= R64_CHUNK) {
    hipLaunchKernelGGL(rank_f64_kernel, dim3((unsigned)nq), dim3(R64_THREADS), lds, ctx->stream, scores, ncols, ld, k, k, 0, d_idx, d_val);
  } else {
    // deeper than one page over more columns than the LDS buffer holds: page through the complete rows
    for (int off = 0; off < k; off += R64_PAGE)
      hipLaunchKernelGGL(rank_f64_kernel, dim3((unsigned)nq), dim3(R64_THREADS), lds, ctx->stream, scores, ncols, ld,
                         std::min(R64_PAGE, k - off), k, off, d_idx, d_val);
  }
  PVS_HIP(hipGetLastError());
  return PVS_OK;
}

}  // namespace pvs
