// VLAD encode on gfx950: K1 assign (KMeans.predict) and K2+K3 aggregate + normalise.
//
// Reference semantics (paths relative to the reference root):
//   K1  pyvisim/encoders/vlad.py:95 -> sklearn/cluster/_k_means_lloyd.pyx:168-218
//         label_i = argmin_j (||c_j||^2 - 2 x_i.c_j), fp32, strict '<' => first minimum wins
//   K2  vlad.py:98-104   V[label_i] += (x_i - c_label_i)  sequentially in descriptor order, fp32
//   K3  vlad.py:106-111  sign(V)|V|^p ; per-cluster row norm + eps ; divide ; k-major flatten
//
// K1 is exact-fp32 GEMM shaped (2*n*K*D flop): v_mfma_f32_32x32x2_f32, centroid block resident in LDS,
// descriptor rows streamed HBM -> registers.  K2/K3 is HBM/latency bound: per image a stable counting
// sort of the labels in LDS, then one lane-group per cluster sums its descriptors IN DESCRIPTOR ORDER
// (bit-identical to the reference's loop given equal labels; no float atomics, run-to-run reproducible).
#include "common.hpp"
#include "desc_load.hpp"

namespace pvs {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ======================================================================================= K1 assign
struct AssignArgs {
  const void* X;
  int64_t total;
  int D, ld;
  const float* Cpad;   // [K_pad][D_pad], zero padded
  const float* cnorm;  // [K_pad], +inf on padded clusters
  int K_pad, D_pad;
  int32_t* labels;
  // second pass of the prefiltered assignment: workgroup b labels the descriptors of list b (counts live on the device)
  const int64_t* rows;              // [gridDim][rows_cap]
  const unsigned long long* nrows;  // [gridDim]
  int64_t rows_cap;
};

constexpr int ASSIGN_THREADS = 512;                 // 8 waves, 2 per SIMD
constexpr int ASSIGN_ROWS = (ASSIGN_THREADS / 64) * 32;  // 256 descriptors per workgroup step
constexpr int ASSIGN_DCHUNK = 128;                  // dims resident in LDS / registers at a time

// Operand roles: MFMA "A" = centroids (rows i = cluster within a 32-tile, from LDS),
//                MFMA "B" = descriptors (cols j = lane & 31, from registers).
// D[i][j] lands with col j on the lane and 16 rows in registers, so the argmin over clusters is
// lane-local except for one exchange between the two half-waves.
// K-slot mapping: lane (., h = lane>>5) feeds dims 8t+4h .. 8t+4h+3 to the four k-steps of step t; the
// same permutation is applied to both operands, so the sum is over all dims.
template <int NT, int KIND, bool VEC>
__global__ __launch_bounds__(ASSIGN_THREADS, 2) void assign_kernel(AssignArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int CB = 32 * NT;
  const int cw_max = a.D_pad < ASSIGN_DCHUNK ? a.D_pad : ASSIGN_DCHUNK;
  const int stride = cw_max + 4;  // +16 B per row: ds_read_b128 of 16 consecutive rows is conflict-free
  float* lds_c = reinterpret_cast<float*>(smem);
  float* lds_n = lds_c + CB * stride;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int ncb = a.K_pad / CB;
  const int nsc = (a.D_pad + ASSIGN_DCHUNK - 1) / ASSIGN_DCHUNK;
  const bool restage = (ncb * nsc) > 1;
  const int64_t n_items = a.rows ? (int64_t)a.nrows[blockIdx.x] : a.total;
  const int64_t nblocks = (n_items + ASSIGN_ROWS - 1) / ASSIGN_ROWS;
  const int64_t* const my_rows = a.rows ? a.rows + (int64_t)blockIdx.x * a.rows_cap : nullptr;

  auto stage = [&](int cb, int sc) {
    const int cw = min(ASSIGN_DCHUNK, a.D_pad - sc * ASSIGN_DCHUNK);
    const int c4n = cw >> 2;
    for (int idx = threadIdx.x; idx < CB * c4n; idx += ASSIGN_THREADS) {
      const int r = idx / c4n, c4 = idx - r * c4n;
      const float4 v = *reinterpret_cast<const float4*>(a.Cpad + (int64_t)(cb * CB + r) * a.D_pad +
                                                        sc * ASSIGN_DCHUNK + 4 * c4);
      *reinterpret_cast<float4*>(lds_c + r * stride + 4 * c4) = v;
    }
    for (int idx = threadIdx.x; idx < CB; idx += ASSIGN_THREADS) lds_n[idx] = a.cnorm[cb * CB + idx];
  };

  if (!restage) {
    stage(0, 0);
    __syncthreads();
  }

  // list mode: this workgroup walks its own list; otherwise the grid strides over all descriptors
  for (int64_t blk = my_rows ? 0 : blockIdx.x; blk < nblocks; blk += my_rows ? 1 : gridDim.x) {
    const int64_t item = blk * ASSIGN_ROWS + wave * 32 + j;
    const bool rvalid = item < n_items;
    const int64_t row = (my_rows && rvalid) ? my_rows[item] : item;
    float best = INFINITY;
    int bidx = 0;

    for (int cb = 0; cb < ncb; ++cb) {
      f32x16 acc[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

      for (int sc = 0; sc < nsc; ++sc) {
        if (restage) {
          __syncthreads();
          stage(cb, sc);
          __syncthreads();
        }
        const int dc = sc * ASSIGN_DCHUNK;
        const int nt = min(ASSIGN_DCHUNK, a.D_pad - dc) >> 3;

        // ---- descriptor fragment: 16 x float4 = this lane's half of a 128-dim slab of its row
        float4 xb[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          xb[t] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (t < nt) {
            const int d = dc + 8 * t + 4 * h;
            if constexpr (VEC) {
              if (rvalid && d < a.D) xb[t] = load4<KIND>(a.X, row, a.ld, d);
            } else {
              if (rvalid) {
                if (d + 0 < a.D) xb[t].x = load1<KIND>(a.X, row, a.ld, d + 0);
                if (d + 1 < a.D) xb[t].y = load1<KIND>(a.X, row, a.ld, d + 1);
                if (d + 2 < a.D) xb[t].z = load1<KIND>(a.X, row, a.ld, d + 2);
                if (d + 3 < a.D) xb[t].w = load1<KIND>(a.X, row, a.ld, d + 3);
              }
            }
          }
        }
        if constexpr (DescTraits<KIND>::rootsift) {
          // host guarantees D <= 128 (one slab) for the RootSIFT kinds
          float s = 0.f;
#pragma unroll
          for (int t = 0; t < 16; ++t) s += (xb[t].x + xb[t].y) + (xb[t].z + xb[t].w);
          s += __shfl_xor(s, 32, 64);
          const RootsiftRow<KIND> rr(s);
#pragma unroll
          for (int t = 0; t < 16; ++t) {
            xb[t].x = rr(xb[t].x);
            xb[t].y = rr(xb[t].y);
            xb[t].z = rr(xb[t].z);
            xb[t].w = rr(xb[t].w);
          }
        }

        // ---- distance tile: NT x (32 clusters x 32 descriptors), exact fp32 MFMA
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          if (t < nt) {
#pragma unroll
            for (int tile = 0; tile < NT; ++tile) {
              const float4 c4 =
                  *reinterpret_cast<const float4*>(lds_c + (32 * tile + j) * stride + 8 * t + 4 * h);
              acc[tile] = __builtin_amdgcn_mfma_f32_32x32x2f32(c4.x, xb[t].x, acc[tile], 0, 0, 0);
              acc[tile] = __builtin_amdgcn_mfma_f32_32x32x2f32(c4.y, xb[t].y, acc[tile], 0, 0, 0);
              acc[tile] = __builtin_amdgcn_mfma_f32_32x32x2f32(c4.z, xb[t].z, acc[tile], 0, 0, 0);
              acc[tile] = __builtin_amdgcn_mfma_f32_32x32x2f32(c4.w, xb[t].w, acc[tile], 0, 0, 0);
            }
          }
        }
      }  // sc

      // ---- lane-local argmin in ascending cluster order (strict '<' keeps the first minimum)
#pragma unroll
      for (int tile = 0; tile < NT; ++tile) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int r0 = 32 * tile + 8 * g + 4 * h;  // C/D layout: row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
          const float4 cn = *reinterpret_cast<const float4*>(lds_n + r0);
          const float v0 = fmaf(-2.f, acc[tile][4 * g + 0], cn.x);
          const float v1 = fmaf(-2.f, acc[tile][4 * g + 1], cn.y);
          const float v2 = fmaf(-2.f, acc[tile][4 * g + 2], cn.z);
          const float v3 = fmaf(-2.f, acc[tile][4 * g + 3], cn.w);
          const int kb = cb * CB + r0;
          if (v0 < best) { best = v0; bidx = kb + 0; }
          if (v1 < best) { best = v1; bidx = kb + 1; }
          if (v2 < best) { best = v2; bidx = kb + 2; }
          if (v3 < best) { best = v3; bidx = kb + 3; }
        }
      }
    }  // cb

    // the two half-waves hold interleaved cluster subsets of the same descriptor
    const float oval = __shfl_xor(best, 32, 64);
    const int oidx = __shfl_xor(bidx, 32, 64);
    if (oval < best || (oval == best && oidx < bidx)) bidx = oidx;
    if (h == 0 && rvalid) a.labels[row] = bidx;
  }
}

// ----------------------------------------------------------------------------------------- K1 prefilter (fp16 MFMA)
// The exact kernel above runs at the f32 MFMA rate (vector-FMA speed).  Most descriptors have a clear nearest centre,
// so a first pass evaluates  v16 = |c|^2 - 2 x.c  on the f16 MFMA with both operands split into two fp16 halves
// (x = xh + xl, c = ch + cl after a power-of-two scaling into the fp16 range; products ch.xh + ch.xl + cl.xh, fp32
// accumulate: 3 MFMAs at 16x the f32 rate) under a proven error bound, and settles every descriptor whose best v16 is
// more than 2 eps below all others -- there the exact kernel's strict-'<' argmin is the same cluster.  The remaining
// descriptors (near ties, non-finite values) are listed per workgroup and labelled by the exact kernel itself in a second
// launch, so the labels are those of the exact kernel for every input.
//   |dot16 - dot32| <= [ 2 2^-22 (operand split) + 2^-22 (dropped cl.xl) + 400 2^-23 (fp32 accumulation of 384 products)
//                       + 128 2^-24 (the exact kernel's own accumulation) + 1e-9 (flush below the fp16 normal range) ] |x||c|
//   eps = 2 |dot16 - dot32| + 2^-22 (|c|^2 + 2 |x||c|)   (the final fma of both kernels)
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));

struct Assign16Args {
  const void* X;
  int64_t total;
  int D, ld;
  const _Float16* C16;  // [2][K_pad][D_pad16]  hi | lo
  const float* cnorm;   // [K_pad], +inf on padded clusters
  int K_pad, D_pad16;
  int c_shift;
  float cmax;
  const _Float16* cnk;  // [K_pad][4]: three exact fp16 pieces of -|c|^2/2 2^(c_shift - cn_e1) (padded clusters: -65504, 0, 0), 0
  int cn_e1, K;
  int32_t* labels;
  int64_t* amb_rows;             // [gridDim][cap]: descriptors left to the exact kernel, one list per workgroup
  unsigned long long* amb_count; // [gridDim]
  int64_t cap;
  float2* rowstat;               // uint8 rows: (row sum + 1e-7, its reciprocal) per descriptor for the aggregate pass, or null
  unsigned long long* stamps;    // diagnostic build (DIAG) only: [16] cycle totals over all workgroups, see pvs_fused_profile
  int nprod;                     // fp16 products per (row, cluster): 3 (product path), 2 or 1 (measurement variants)
  int shape16;                   // != 0: assign16x_kernel (v_mfma_f32_16x16x32_f16) where its shape qualifies (measurement variant)
};

// STEPS: the number of 16-dim k-steps when it is known at compile time (8 for D_pad16 = 128), 0 = read it from the arguments.
// With STEPS > 0 the whole cluster loop of a row block is straight-line code: table fragments are fetched from LDS one step
// ahead of the MFMAs that use them, and the selection over one pair of tiles runs under the MFMAs of the next pair.
// NP: fp16 products per (row, cluster): 3 = ch.xh + ch.xl + cl.xh (the product path), 2 = ch.xh + ch.xl (cl.xh dropped), 1 = ch.xh
// only.  Fewer products widen the proven margin by 2^-11 |x||c| each, so more rows are left to the exact kernel -- the labels stay
// the exact kernel's either way.  NP < 3 exists to MEASURE that trade (pvs_set_option(PVS_OPT_ASSIGN_PREFILTER, 2 | 3), STEPS > 0 only).
template <int NT, int KIND, bool VEC, int STEPS, bool DIAG = false, int NP = 3>
__global__ __launch_bounds__(ASSIGN_THREADS, 2) void assign16_kernel(Assign16Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ unsigned int s_count;
  constexpr int CB = 32 * NT;
  constexpr int stride = 128 + 8;     // halfs, fixed (D <= 128): compile-time fragment offsets; +16 B per row keeps the 16-B reads conflict-free
  _Float16* lds_h = reinterpret_cast<_Float16*>(smem);
  _Float16* lds_l = lds_h + CB * stride;
  float* lds_n = reinterpret_cast<float*>(lds_l + CB * stride);
  _Float16* lds_k = reinterpret_cast<_Float16*>(lds_n + CB);    // [CB][4]: the -|c|^2/2 pieces (STEPS > 0 only)
  // uint8 rows (LUT): sqrt(raw) for raw = 0..255 as an fp16 pair (hi | lo << 16).  RootSIFT's element is sqrt(raw) / sqrt(d) with ONE
  // factor per row, so the prefilter works on T = sqrt(raw) -- a table lookup and two byte permutes per element, no arithmetic
  // at all -- and the row's factor sqrt(d) moves into the -|c|^2/2 step:  x.c - |c|^2/2 = (T.c - |c|^2 sqrt(d) / 2) / sqrt(d).
  constexpr bool LUT = KIND == PVS_DESC_U8_ROOTSIFT && STEPS > 0 && VEC;
  uint32_t* lds_t = reinterpret_cast<uint32_t*>(lds_k + 4 * CB);   // [256]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int nt = a.D_pad16 >> 4;      // k-steps of 16 dims
  const int64_t nblocks = (a.total + ASSIGN_ROWS - 1) / ASSIGN_ROWS;

  for (int idx = threadIdx.x; idx < 2 * CB * (a.D_pad16 >> 3); idx += ASSIGN_THREADS) {
    const int per = a.D_pad16 >> 3;
    const int r = idx / per, c8 = idx - r * per;      // r < 2 CB: hi rows then lo rows
    *reinterpret_cast<uint4*>(lds_h + r * stride + 8 * c8) = *reinterpret_cast<const uint4*>(a.C16 + (int64_t)r * a.D_pad16 + 8 * c8);
  }
  for (int idx = threadIdx.x; idx < CB; idx += ASSIGN_THREADS) lds_n[idx] = a.cnorm[idx];
  if constexpr (STEPS > 0)
    for (int idx = threadIdx.x; idx < CB; idx += ASSIGN_THREADS)
      *reinterpret_cast<uint2*>(lds_k + 4 * idx) = *reinterpret_cast<const uint2*>(a.cnk + 4 * idx);
  if constexpr (LUT) {
    if (threadIdx.x < 256) {
      const float sv = sqrtf((float)threadIdx.x);
      const _Float16 hi = (_Float16)sv, lo = (_Float16)(sv - (float)hi);
      lds_t[threadIdx.x] = (uint32_t)__builtin_bit_cast(unsigned short, hi) | ((uint32_t)__builtin_bit_cast(unsigned short, lo) << 16);
    }
  }
  if (threadIdx.x == 0) s_count = 0u;
  __syncthreads();
  const float sqrt_d = sqrtf((float)a.D);
  int64_t* const my_rows = a.amb_rows + (int64_t)blockIdx.x * a.cap;

  // STEPS > 0 (launched only for D = 128) with float rows: the rows of the NEXT block are requested before this block's MFMA
  // phase and land under it.  Loads are unconditional: a row past the end reads the last row instead, its result is not stored.
  constexpr bool PREFETCH = STEPS > 0 && VEC;
  float xf[8][8];
  uint32_t xw[8][2];       // LUT: the raw bytes of this lane's half row
  auto request_rows = [&](int64_t blk_) {
    int64_t r = blk_ * ASSIGN_ROWS + wave * 32 + j;
    r = r < a.total ? r : a.total - 1;
    if constexpr (LUT) {
      const uint8_t* pr = static_cast<const uint8_t*>(a.X) + r * a.ld + 4 * h;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        xw[t][0] = *reinterpret_cast<const uint32_t*>(pr + 16 * t);
        xw[t][1] = *reinterpret_cast<const uint32_t*>(pr + 16 * t + 8);
      }
      return;
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float4 v0 = load4<KIND>(a.X, r, a.ld, 16 * t + 4 * h);
      const float4 v1 = load4<KIND>(a.X, r, a.ld, 16 * t + 4 * h + 8);
      xf[t][0] = v0.x; xf[t][1] = v0.y; xf[t][2] = v0.z; xf[t][3] = v0.w;
      xf[t][4] = v1.x; xf[t][5] = v1.y; xf[t][6] = v1.z; xf[t][7] = v1.w;
    }
  };
  if constexpr (PREFETCH) {
    if (blockIdx.x < nblocks) request_rows(blockIdx.x);
  }

  // STEPS > 0: ping-pong.  The 8 waves are two groups (waves 0-3 / 4-7: one wave of each per SIMD).  A row block is
  //   [conversion (+ the previous block's tail) | barrier | MFMA loop | barrier]
  // and group 1 runs one barrier behind group 0, so that on every SIMD one wave feeds the matrix pipe while the other does its
  // vector work (conversion: fp32 multiplies / subtracts that do block the pipe when they come from the SAME phase on both
  // waves; left to drift, the two waves of a SIMD spent a third of the time converting together with the pipe idle: 69 % busy).
  // All waves of a workgroup run the same number of blocks, so the barrier counts match.
  constexpr bool PP = STEPS > 0;
  const bool pp_g1 = wave >= ASSIGN_THREADS / 128;
  if constexpr (PP) {
    if (pp_g1) __builtin_amdgcn_s_barrier();
  }
  unsigned long long dg[6] = {0, 0, 0, 0, 0, 0}, dt0 = 0, dt1 = 0, dt2 = 0, dt3 = 0, dt4 = 0;
  for (int64_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
    if constexpr (DIAG) dt0 = __builtin_amdgcn_s_memtime();
    const int64_t row = blk * ASSIGN_ROWS + wave * 32 + j;
    const bool rvalid = row < a.total;
    // ---- this lane's half of the row: of every 16 dims the four at 4h and the four at 8 + 4h (the lane pair (j, 0), (j, 1)
    // reads 32 contiguous bytes per load; the fp16 tables are stored in the same order)
    if constexpr (!PREFETCH) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
#pragma unroll
      for (int q = 0; q < 8; ++q) xf[t][q] = 0.f;
      if (t < nt && rvalid) {
        const int d = 16 * t + 4 * h;
        if constexpr (VEC) {
          if (d < a.D) {
            const float4 v = load4<KIND>(a.X, row, a.ld, d);
            xf[t][0] = v.x; xf[t][1] = v.y; xf[t][2] = v.z; xf[t][3] = v.w;
          }
          if (d + 8 < a.D) {
            const float4 v = load4<KIND>(a.X, row, a.ld, d + 8);
            xf[t][4] = v.x; xf[t][5] = v.y; xf[t][6] = v.z; xf[t][7] = v.w;
          }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (d + q < a.D) xf[t][q] = load1<KIND>(a.X, row, a.ld, d + q);
            if (d + 8 + q < a.D) xf[t][4 + q] = load1<KIND>(a.X, row, a.ld, d + 8 + q);
          }
        }
      }
    }
    }
    float nx;
    int x_shift = 0;
    bool finite;
    f16x8_t xh[8], xl[8];
    _Float16 lut_f1 = (_Float16)0.f, lut_f2 = (_Float16)0.f;   // LUT: sqrt(d) 2^cn_e1 as two fp16 pieces (the row side of the -|c|^2/2 step)
    float lut_sd = 1.f;                                        // LUT: sqrt(d): this row's scores and margins are in units of 1 / sqrt(d)
    if constexpr (LUT) {
      unsigned ssum = 0;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        ssum = __builtin_amdgcn_sad_u8(xw[t][0], 0u, ssum);
        ssum = __builtin_amdgcn_sad_u8(xw[t][1], 0u, ssum);
      }
      ssum += __shfl_xor(ssum, 32, 64);
      const float sf = (float)ssum, dd = sf + 1e-7f;
      if (a.rowstat != nullptr && h == 0 && rvalid) a.rowstat[row] = make_float2(dd, 1.0f / dd);   // = RootsiftRow(s), bit for bit
      lut_sd = sqrtf(dd);
      const float ff = ldexpf(lut_sd, a.cn_e1);
      lut_f1 = (_Float16)ff;
      lut_f2 = (_Float16)(ff - (float)lut_f1);
      // f1 and f2 must be normal fp16 numbers (f2 ~ 2^-11 f1): 2^-3 <= f <= 2^15; all-zero rows (d = 1e-7) go to the exact kernel
      finite = ff >= 0.125f && ff <= 32768.f;
      nx = sqrtf(sf) * 1.0001f;              // |T| = sqrt(sum of the raw row), exactly
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        uint32_t hw[4], lw[4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int pq = 0; pq < 2; ++pq) {
            const uint32_t w = xw[t][i];
            const uint32_t e0 = lds_t[(w >> (16 * pq)) & 0xffu], e1 = lds_t[(w >> (16 * pq + 8)) & 0xffu];
            hw[2 * i + pq] = __builtin_amdgcn_perm(e1, e0, 0x05040100u);   // hi(e0) | hi(e1) << 16
            lw[2 * i + pq] = __builtin_amdgcn_perm(e1, e0, 0x07060302u);   // lo(e0) | lo(e1) << 16
          }
        xh[t] = __builtin_bit_cast(f16x8_t, make_uint4(hw[0], hw[1], hw[2], hw[3]));
        xl[t] = __builtin_bit_cast(f16x8_t, make_uint4(lw[0], lw[1], lw[2], lw[3]));
      }
    } else {
    // ---- row norm, row scale (largest |x| 2^shift in [2^12, 2^13)), hi / lo halves
    float n2 = 0.f, amax = 0.f, rs_r = 0.f;
    if constexpr (DescTraits<KIND>::rootsift) {
      // The prefilter does not need the reference's bits of sqrt(raw / (sum + 1e-7)), only a value within a known distance of
      // them: y' = v_sqrt(raw * (1 / d)) is within 2^-21 relative of the exact element (1/d, the product and the square root
      // round once each: 2^-24 + 2^-24 halved by the root, + 1 ulp of v_sqrt_f32; a raw row sum formed in another order than the
      // exact kernel's moves every element by the same few ulps), which adds 2^-21 |x||c| to a product -- 1 % of the margin
      // (`eps` below carries it).  |x|^2 = sum / d and the largest element follow from the raw row: 3 instead of 16 vector
      // instructions per element, none of them in the exact kernels (assign_kernel, aggregate, fused), whose values are unchanged.
      float s = 0.f, rmax = 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          s += xf[t][q];
          rmax = fmaxf(rmax, xf[t][q]);
        }
      s += __shfl_xor(s, 32, 64);
      rmax = fmaxf(rmax, __shfl_xor(rmax, 32, 64));
      rs_r = 1.0f / (s + 1e-7f);
      n2 = s * rs_r * 1.0001f;                                   // >= |x|^2 of the exact row (every element within 2^-21)
      amax = __builtin_amdgcn_sqrtf(rmax * rs_r) * 1.0001f;      // >= the largest element
      if (!(s >= 0.f) || !(rmax * rs_r <= 3.0e38f)) n2 = NAN;    // negative / non-finite raw rows go to the exact kernel
      if constexpr (KIND == PVS_DESC_U8_ROOTSIFT) {
        if (a.rowstat != nullptr && h == 0 && rvalid) a.rowstat[row] = make_float2(s + 1e-7f, rs_r);   // = RootsiftRow(s), bit for bit
      }
    } else {
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          n2 = fmaf(xf[t][q], xf[t][q], n2);
          amax = fmaxf(amax, fabsf(xf[t][q]));
        }
      n2 += __shfl_xor(n2, 32, 64);
      amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
    }
    nx = sqrtf(n2) * 1.0001f;
    int ex = 13;
    if (amax > 0.f) (void)frexpf(amax, &ex);
    x_shift = 13 - ex;
    finite = nx <= 3.0e38f;   // false for NaN too
    if (x_shift > 40 || x_shift < -40) { finite = false; x_shift = 0; }
    if constexpr (STEPS > 0) {     // the row scale enters the -|c|^2/2 step as an fp16 factor: it must be a normal fp16 number
      if (x_shift + a.cn_e1 < -14 || x_shift + a.cn_e1 > 15) finite = false;
    }
    const float xs = ldexpf(1.f, x_shift);
    const float rs_rs = rs_r * xs * xs;     // rootsift kinds: sqrt(raw r) 2^shift = sqrt(raw r 4^shift), the scale is exact
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        float v;
        if constexpr (DescTraits<KIND>::rootsift) v = __builtin_amdgcn_sqrtf(xf[t][q] * rs_rs);
        else v = xf[t][q] * xs;
        const _Float16 hi = (_Float16)v;
        xh[t][q] = hi;
        xl[t][q] = (_Float16)(v - (float)hi);
      }
    }
    // ---- clusters in groups of G tiles (accumulators of one group live at a time): smallest and second smallest v16
    constexpr int G = NT < 4 ? NT : 4;
    const float m2s = -2.f * ldexpf(1.f, -(x_shift + a.c_shift));   // v = cn - 2 acc 2^-(shifts): one fma, the scale is exact (LUT: x_shift = 0, v in units of 1 / sqrt(d))
    float best = INFINITY, second = INFINITY;
    int bidx = 0;
    if constexpr (PREFETCH) {
      __builtin_amdgcn_sched_barrier(0);      // after the conversion: the old row registers are dead, the new ones not yet live
      if (blk + gridDim.x < nblocks) request_rows(blk + gridDim.x);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (DIAG) dt1 = __builtin_amdgcn_s_memtime();
    if constexpr (PP) __builtin_amdgcn_s_barrier();
    if constexpr (DIAG) dt2 = __builtin_amdgcn_s_memtime();
    if constexpr (STEPS > 0) {
      constexpr int G2 = NT < 2 ? NT : 2;     // tiles per group: two accumulator sets alternate between groups
      constexpr int NG = NT / G2;
      f32x16 acc[2][G2];
      f16x8_t fh[2][G2], fl[2][G2];
      auto fetch = [&](int buf, int step) {
        const int g = step / STEPS, t = step % STEPS;
#pragma unroll
        for (int tile = 0; tile < G2; ++tile) {
          fh[buf][tile] = *reinterpret_cast<const f16x8_t*>(lds_h + (32 * (g * G2 + tile) + j) * stride + 16 * t + 8 * h);
          fl[buf][tile] = *reinterpret_cast<const f16x8_t*>(lds_l + (32 * (g * G2 + tile) + j) * stride + 16 * t + 8 * h);
        }
      };
      // The scan works on the MFMA's own output s = 2^(shifts) (x.c - |c|^2/2): -|c|^2/2 enters as one more k-step (three exact
      // fp16 pieces of the table side times the row's power of two, added LAST so that the products accumulate as before), and
      // the LARGEST s is the nearest centre.  No fp32 add / mul / fma is left in the scan: compare, v_med3 and selects issue
      // next to the matrix pipe, fp32 arithmetic does not (profiles/r02_coissue.txt).
      // selection over the 4 clusters (tile, q) of group g this lane holds in acc[ab][tile][4 q ..]: 16 VALU instructions
      auto select4 = [&](int ab, int g, int part) {
        const int tile = part >> 2, q = part & 3;
        const int r0 = 32 * (g * G2 + tile) + 8 * q;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = acc[ab][tile][4 * q + e];
          const bool gt = v > best;          // ascending cluster order, strict '>': the first maximum stays
          second = __builtin_amdgcn_fmed3f(best, second, v);
          best = gt ? v : best;
          bidx = gt ? (r0 + e) : bidx;       // the lane's 4 h is added after the loop
        }
      };
      f16x8_t cnb;                            // B fragment of the -|c|^2/2 step: the row's 2^(x_shift + e1) in k-slots 0..2 of half-wave 0
      {
        const _Float16 pw = (h == 0 && finite) ? (_Float16)ldexpf(1.f, x_shift + a.cn_e1) : (_Float16)0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) cnb[q] = (_Float16)0.f;
        if constexpr (LUT) {   // six exact products: (three pieces of -|c|^2/2) x (two pieces of sqrt(d) 2^cn_e1)
          if (h == 0 && finite) { cnb[0] = lut_f1; cnb[1] = lut_f2; cnb[2] = lut_f1; cnb[3] = lut_f2; cnb[4] = lut_f1; cnb[5] = lut_f2; }
        } else {
          cnb[0] = pw; cnb[1] = pw; cnb[2] = pw;
        }
      }
      best = -INFINITY;
      second = -INFINITY;
      static_assert(G2 * 4 <= STEPS, "one selection part per k-step");
      fetch(0, 0);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int ab = g & 1;
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < STEPS; ++t) {
          const int step = g * STEPS + t, buf = step & 1;
          if (step + 1 < NG * STEPS) fetch(buf ^ 1, step + 1);
          __builtin_amdgcn_sched_barrier(0);     // the fetch stays AHEAD of this step's MFMAs (the scheduler sinks it otherwise)
          // the three products of a tile go to the same accumulator in a fixed order; the tiles alternate so that an MFMA
          // never waits for the result of the one issued just before it.  (The group's first MFMA takes the constant 0 as its
          // accumulator input: no sixteen v_mov per tile to clear it.)
          if constexpr (NP >= 3) {
#pragma unroll
            for (int tile = 0; tile < G2; ++tile)
              acc[ab][tile] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[buf][tile], xh[t], t == 0 ? zero16 : acc[ab][tile], 0, 0, 0);
          }
          if constexpr (NP >= 2) {
#pragma unroll
            for (int tile = 0; tile < G2; ++tile)
              acc[ab][tile] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[buf][tile], xl[t], (NP == 2 && t == 0) ? zero16 : acc[ab][tile], 0, 0, 0);
          }
#pragma unroll
          for (int tile = 0; tile < G2; ++tile)
            acc[ab][tile] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[buf][tile], xh[t], (NP == 1 && t == 0) ? zero16 : acc[ab][tile], 0, 0, 0);
          if (t == STEPS - 1) {                  // - |c|^2 / 2 . 2^(shifts), after every product of the group
#pragma unroll
            for (int tile = 0; tile < G2; ++tile) {
              const f16x4_t pk = *reinterpret_cast<const f16x4_t*>(lds_k + 4 * (32 * (g * G2 + tile) + j));
              f16x8_t ca;
#pragma unroll
              for (int q = 0; q < 8; ++q) ca[q] = (_Float16)0.f;
              if constexpr (LUT) {
                if (h == 0) { ca[0] = pk[0]; ca[1] = pk[0]; ca[2] = pk[1]; ca[3] = pk[1]; ca[4] = pk[2]; ca[5] = pk[2]; }
              } else {
                if (h == 0) { ca[0] = pk[0]; ca[1] = pk[1]; ca[2] = pk[2]; }
              }
              acc[ab][tile] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ca, cnb, acc[ab][tile], 0, 0, 0);
            }
          }
          if (g > 0 && t < G2 * 4) select4(ab ^ 1, g - 1, t);   // the previous group's selection, a part under each step's MFMAs
          __builtin_amdgcn_sched_barrier(0);     // keep the steps apart: hoisting every fetch to the top spills
        }
      }
#pragma unroll
      for (int part = 0; part < G2 * 4; ++part) select4((NG - 1) & 1, NG - 1, part);
      // back to v = |c|^2 - 2 x.c = m2s s (exact: a power of two), where the smallest is the nearest
      best *= m2s;
      second *= m2s;
    } else {
#pragma unroll
    for (int g0 = 0; g0 < NT; g0 += G) {
      f32x16 acc[G];
#pragma unroll
      for (int t = 0; t < G; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        if (t < nt) {
#pragma unroll
          for (int tile = 0; tile < G; ++tile) {
            const f16x8_t ch = *reinterpret_cast<const f16x8_t*>(lds_h + (32 * (g0 + tile) + j) * stride + 16 * t + 8 * h);
            const f16x8_t cl = *reinterpret_cast<const f16x8_t*>(lds_l + (32 * (g0 + tile) + j) * stride + 16 * t + 8 * h);
            acc[tile] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cl, xh[t], acc[tile], 0, 0, 0);
            acc[tile] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch, xl[t], acc[tile], 0, 0, 0);
            acc[tile] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch, xh[t], acc[tile], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int tile = 0; tile < G; ++tile)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int r0 = 32 * (g0 + tile) + 8 * g + 4 * h;
          const float4 cn = *reinterpret_cast<const float4*>(lds_n + r0);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = fmaf(m2s, acc[tile][4 * g + e], e == 0 ? cn.x : (e == 1 ? cn.y : (e == 2 ? cn.z : cn.w)));
            // ascending cluster order: strict '<' keeps the first minimum.  v is never NaN for a row that can settle (finite
            // x and tables give finite products; padded clusters are +inf), and rows with a non-finite x are listed whatever
            // comes out here.  Selects, not branches: as control flow this was ~2000 instructions per step.
            const bool lt = v < best;
            second = __builtin_amdgcn_fmed3f(best, second, v);       // best <= second: the median is the new runner-up
            best = lt ? v : best;
            bidx = lt ? (32 * (g0 + tile) + 8 * g + e) : bidx;       // the lane's 4 h is added after the loop
          }
        }
    }
    }
    if constexpr (DIAG) dt3 = __builtin_amdgcn_s_memtime();
    if constexpr (PP) __builtin_amdgcn_s_barrier();
    if constexpr (DIAG) dt4 = __builtin_amdgcn_s_memtime();
    bidx += 4 * h;
    {   // the two half-waves hold interleaved cluster subsets of the same descriptor
      const float ob = __shfl_xor(best, 32, 64), os = __shfl_xor(second, 32, 64);
      const int oi = __shfl_xor(bidx, 32, 64);
      const bool take = ob < best || (ob == best && oi < bidx);
      const float nb = take ? ob : best, loser = take ? best : ob;
      second = fminf(fminf(second, os), loser);
      best = nb;
      if (take) bidx = oi;
    }
    const float xc = nx * a.cmax;
    constexpr float conv_err = DescTraits<KIND>::rootsift ? 4.8e-7f : 0.f;   // 2^-21: the approximate RootSIFT elements above
    // LUT rows: everything in units of 1 / sqrt(d) (|T| |c| instead of |x| |c|; |c|^2 sqrt(d) instead of |c|^2); the -|c|^2/2 term
    // additionally carries the dropped third piece of sqrt(d) (2^-22) and six accumulation roundings (6 2^-24): 6e-7 instead of 2.4e-7
    const float cc = LUT ? a.cmax * a.cmax * lut_sd : a.cmax * a.cmax;
    constexpr float drop_err = (3 - NP) * 4.9e-4f;    // a dropped correction product: |cl| <= 2^-11 |c| or |xl| <= 2^-11 |x| element by element
    const float eps = 2.f * (4.8e-7f + 2.4e-7f + 4.8e-5f + 7.7e-6f + 1e-9f + conv_err + drop_err) * xc * (1.f + sqrt_d * 1e-9f) + (LUT ? 6.0e-7f : 2.4e-7f) * (cc + 2.f * xc);
    const int within = second <= best + 2.f * eps ? 2 : 1;
    // settled: exactly one cluster within the margin (the minimum itself) and everything finite
    const bool settled = within == 1 && finite && fabsf(best) <= 3.0e38f && bidx < a.K;   // (a padded cluster never settles a row)
    // the rest goes on this workgroup's list: one LDS atomic per wave (ballot + prefix count)
    const bool amb = h == 0 && rvalid && !settled;
    const unsigned long long amask = __ballot(amb);
    unsigned int base = 0;
    if (amask != 0ull) {
      if (lane == 0) base = atomicAdd(&s_count, (unsigned int)__popcll(amask));
      base = __shfl(base, 0, 64);
    }
    if (h == 0 && rvalid) {
      if (settled) a.labels[row] = bidx;
      else my_rows[base + __popcll(amask & ((1ull << lane) - 1ull))] = row;
    }
    if constexpr (DIAG) {
      const unsigned long long dt5 = __builtin_amdgcn_s_memtime();
      dg[0] += dt1 - dt0; dg[1] += dt2 - dt1; dg[2] += dt3 - dt2; dg[3] += dt4 - dt3; dg[4] += dt5 - dt4; dg[5] += 1;
    }
  }
  if constexpr (PP) {
    if (!pp_g1) __builtin_amdgcn_s_barrier();
  }
  if constexpr (DIAG) {   // waves 0 (group 0) and 4 (group 1): [0..4] / [5..9] conversion, barrier, MFMA loop, barrier, tail; [10] blocks
    if (a.stamps != nullptr && lane == 0 && (wave == 0 || wave == 4)) {
      const int o = wave == 0 ? 0 : 5;
      for (int q = 0; q < 5; ++q) atomicAdd(a.stamps + o + q, dg[q]);
      if (wave == 0) atomicAdd(a.stamps + 10, dg[5]);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) a.amb_count[blockIdx.x] = s_count;
}


// ---------------------------------------------------------------------------------------------------------------------------
// assign16x_kernel: the prefilter of the shape the benchmarks use (D = 128, 128 < K <= 256, 16-byte aligned rows) on
// v_mfma_f32_16x16x32_f16 -- a MEASUREMENT VARIANT (pvs_set_option(PVS_OPT_ASSIGN_PREFILTER, 4)).  Same products, same order per
// (row, cluster) element (cl.xh, ch.xl, ch.xh per 32-dim step, then the three pieces of -|c|^2/2), same margin, same lists for the
// exact kernel as assign16_kernel; what changes is the instruction shape (on the fp16 GEMM the 16 x 16 x 32 shape was worth 6-8 %
// through the clock the part holds).  Here it is 8-10 % SLOWER than the 32 x 32 x 16 kernel (profiles/r03_prefilter_16x16.txt:
// vlad512 assign 3.92 against 3.55 ms, headline 2.48 against 2.29 ms): twice the MFMA instructions for the same flop, four instead
// of two -|c|^2/2 instructions per group, and every per-descriptor step of the tail done for two descriptors per lane.
//   A operand = 16 clusters x 32 dims from the LDS tables: lane (i = lane & 15, g = lane >> 4) reads the 8 halfs at position
//     32 s + 8 g of cluster row i -- in the tables' stored order that is dims 16 t + {4h .. 4h+3, 8+4h .. 8+4h+3}, t = 2 s + (g >> 1),
//     h = g & 1;
//   B operand = 16 descriptors x 32 dims from registers: lane (j = lane & 15, g) holds the same dims of descriptors j (fragment 0)
//     and 16 + j (fragment 1) of the wave's 32;
//   C: lane (j, q = lane >> 4) holds clusters 4 q .. 4 q + 3 of the tile for descriptor j: the argmin is lane-local over tiles and
//     finishes with two exchanges among the four q-lanes of a descriptor.
// Clusters go through in 8 groups of 32 (two tiles x two fragments x 4 registers = 16 accumulator registers, two sets alternating);
// the selection over a group runs under the MFMAs of the next.
typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int KIND, int NP = 3>
__global__ __launch_bounds__(ASSIGN_THREADS, 2) void assign16x_kernel(Assign16Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ unsigned int s_count;
  constexpr int CB = 256, stride = 128 + 8;
  _Float16* lds_h = reinterpret_cast<_Float16*>(smem);
  _Float16* lds_l = lds_h + CB * stride;
  float* lds_n = reinterpret_cast<float*>(lds_l + CB * stride);    // (layout shared with assign16_kernel; not read here)
  _Float16* lds_k = reinterpret_cast<_Float16*>(lds_n + CB);       // [CB][4]: the -|c|^2/2 pieces
  constexpr bool LUT = KIND == PVS_DESC_U8_ROOTSIFT;
  uint32_t* lds_t = reinterpret_cast<uint32_t*>(lds_k + 4 * CB);   // [256] uint8 rows: sqrt(raw) as an fp16 pair
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int64_t nblocks = (a.total + ASSIGN_ROWS - 1) / ASSIGN_ROWS;

  for (int idx = threadIdx.x; idx < 2 * CB * 16; idx += ASSIGN_THREADS) {
    const int r = idx >> 4, c8 = idx & 15;      // r < 2 CB: hi rows then lo rows
    *reinterpret_cast<uint4*>(lds_h + r * stride + 8 * c8) = *reinterpret_cast<const uint4*>(a.C16 + (int64_t)r * 128 + 8 * c8);
  }
  for (int idx = threadIdx.x; idx < CB; idx += ASSIGN_THREADS)
    *reinterpret_cast<uint2*>(lds_k + 4 * idx) = *reinterpret_cast<const uint2*>(a.cnk + 4 * idx);
  if constexpr (LUT) {
    if (threadIdx.x < 256) {
      const float sv = sqrtf((float)threadIdx.x);
      const _Float16 hi = (_Float16)sv, lo = (_Float16)(sv - (float)hi);
      lds_t[threadIdx.x] = (uint32_t)__builtin_bit_cast(unsigned short, hi) | ((uint32_t)__builtin_bit_cast(unsigned short, lo) << 16);
    }
  }
  if (threadIdx.x == 0) s_count = 0u;
  __syncthreads();
  const float sqrt_d = sqrtf((float)a.D);
  int64_t* const my_rows = a.amb_rows + (int64_t)blockIdx.x * a.cap;
  // the dims this lane holds of every 32-dim step s: 16 t + 4 h + {0..3} and 16 t + 8 + 4 h + {0..3}, t = 2 s + (g >> 1), h = g & 1
  const int dlane = 16 * (g >> 1) + 4 * (g & 1);

  // rows of the NEXT block are requested before this block's MFMA phase and land under it (unconditional loads: a row past the end
  // reads the last row instead, its result is not stored)
  float xf[2][4][8];
  uint32_t xw[2][4][2];
  auto request_rows = [&](int64_t blk_) {
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      int64_t r = blk_ * ASSIGN_ROWS + wave * 32 + 16 * f + j;
      r = r < a.total ? r : a.total - 1;
      if constexpr (LUT) {
        const uint8_t* pr = static_cast<const uint8_t*>(a.X) + r * a.ld + dlane;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          xw[f][s][0] = *reinterpret_cast<const uint32_t*>(pr + 32 * s);
          xw[f][s][1] = *reinterpret_cast<const uint32_t*>(pr + 32 * s + 8);
        }
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float4 v0 = load4<KIND>(a.X, r, a.ld, 32 * s + dlane);
          const float4 v1 = load4<KIND>(a.X, r, a.ld, 32 * s + dlane + 8);
          xf[f][s][0] = v0.x; xf[f][s][1] = v0.y; xf[f][s][2] = v0.z; xf[f][s][3] = v0.w;
          xf[f][s][4] = v1.x; xf[f][s][5] = v1.y; xf[f][s][6] = v1.z; xf[f][s][7] = v1.w;
        }
      }
    }
  };
  if (blockIdx.x < nblocks) request_rows(blockIdx.x);

  // ping-pong: waves 0-3 / 4-7 (one of each per SIMD); group 1 runs one barrier behind group 0, so that on every SIMD one wave
  // feeds the matrix pipe while the other does its vector work.  All waves run the same number of blocks: the barrier counts match.
  const bool pp_g1 = wave >= ASSIGN_THREADS / 128;
  if (pp_g1) __builtin_amdgcn_s_barrier();
  for (int64_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
    f16x8_t xh[2][4], xl[2][4];
    float nx[2], lut_sd[2] = {1.f, 1.f};
    int x_shift[2] = {0, 0};
    bool finite[2], rvalid[2];
    _Float16 lut_f1[2] = {(_Float16)0.f, (_Float16)0.f}, lut_f2[2] = {(_Float16)0.f, (_Float16)0.f};
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const int64_t row = blk * ASSIGN_ROWS + wave * 32 + 16 * f + j;
      rvalid[f] = row < a.total;
      if constexpr (LUT) {
        unsigned ssum = 0;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          ssum = __builtin_amdgcn_sad_u8(xw[f][s][0], 0u, ssum);
          ssum = __builtin_amdgcn_sad_u8(xw[f][s][1], 0u, ssum);
        }
        ssum += __shfl_xor(ssum, 16, 64);
        ssum += __shfl_xor(ssum, 32, 64);
        const float sf = (float)ssum, dd = sf + 1e-7f;
        if (a.rowstat != nullptr && g == 0 && rvalid[f]) a.rowstat[row] = make_float2(dd, 1.0f / dd);   // = RootsiftRow(s), bit for bit
        lut_sd[f] = sqrtf(dd);
        const float ff = ldexpf(lut_sd[f], a.cn_e1);
        lut_f1[f] = (_Float16)ff;
        lut_f2[f] = (_Float16)(ff - (float)lut_f1[f]);
        finite[f] = ff >= 0.125f && ff <= 32768.f;      // f1, f2 normal fp16 numbers; all-zero rows go to the exact kernel
        nx[f] = sqrtf(sf) * 1.0001f;                    // |T| = sqrt(sum of the raw row), exactly
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          uint32_t hw[4], lw[4];
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pq = 0; pq < 2; ++pq) {
              const uint32_t w = xw[f][s][i];
              const uint32_t e0 = lds_t[(w >> (16 * pq)) & 0xffu], e1 = lds_t[(w >> (16 * pq + 8)) & 0xffu];
              hw[2 * i + pq] = __builtin_amdgcn_perm(e1, e0, 0x05040100u);   // hi(e0) | hi(e1) << 16
              lw[2 * i + pq] = __builtin_amdgcn_perm(e1, e0, 0x07060302u);   // lo(e0) | lo(e1) << 16
            }
          xh[f][s] = __builtin_bit_cast(f16x8_t, make_uint4(hw[0], hw[1], hw[2], hw[3]));
          xl[f][s] = __builtin_bit_cast(f16x8_t, make_uint4(lw[0], lw[1], lw[2], lw[3]));
        }
      } else {
        float n2 = 0.f, amax = 0.f, rs_r = 0.f;
        if constexpr (DescTraits<KIND>::rootsift) {   // see assign16_kernel: y' = v_sqrt(raw / d) within 2^-21 of the exact element
          float sm = 0.f, rmax = 0.f;
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              sm += xf[f][s][q];
              rmax = fmaxf(rmax, xf[f][s][q]);
            }
          sm += __shfl_xor(sm, 16, 64);
          sm += __shfl_xor(sm, 32, 64);
          rmax = fmaxf(rmax, __shfl_xor(rmax, 16, 64));
          rmax = fmaxf(rmax, __shfl_xor(rmax, 32, 64));
          rs_r = 1.0f / (sm + 1e-7f);
          n2 = sm * rs_r * 1.0001f;
          amax = __builtin_amdgcn_sqrtf(rmax * rs_r) * 1.0001f;
          if (!(sm >= 0.f) || !(rmax * rs_r <= 3.0e38f)) n2 = NAN;
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              n2 = fmaf(xf[f][s][q], xf[f][s][q], n2);
              amax = fmaxf(amax, fabsf(xf[f][s][q]));
            }
          n2 += __shfl_xor(n2, 16, 64);
          n2 += __shfl_xor(n2, 32, 64);
          amax = fmaxf(amax, __shfl_xor(amax, 16, 64));
          amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
        }
        nx[f] = sqrtf(n2) * 1.0001f;
        int ex = 13;
        if (amax > 0.f) (void)frexpf(amax, &ex);
        x_shift[f] = 13 - ex;
        finite[f] = nx[f] <= 3.0e38f;   // false for NaN too
        if (x_shift[f] > 40 || x_shift[f] < -40) { finite[f] = false; x_shift[f] = 0; }
        if (x_shift[f] + a.cn_e1 < -14 || x_shift[f] + a.cn_e1 > 15) finite[f] = false;   // the row scale enters the -|c|^2/2 step as a normal fp16 number
        const float xs = ldexpf(1.f, x_shift[f]);
        const float rs_rs = rs_r * xs * xs;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            float v;
            if constexpr (DescTraits<KIND>::rootsift) v = __builtin_amdgcn_sqrtf(xf[f][s][q] * rs_rs);
            else v = xf[f][s][q] * xs;
            const _Float16 hi = (_Float16)v;
            xh[f][s][q] = hi;
            xl[f][s][q] = (_Float16)(v - (float)hi);
          }
      }
    }
    __builtin_amdgcn_sched_barrier(0);      // after the conversion: the old row registers are dead, the new ones not yet live
    if (blk + gridDim.x < nblocks) request_rows(blk + gridDim.x);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();

    // ---- 8 groups of 32 clusters: s = 2^(shifts) (x.c - |c|^2/2) on the matrix pipe, the LARGEST s is the nearest centre
    f32x4v acc[2][2][2];                    // [set][tile][fragment]
    f16x8_t fh[2][2], fl[2][2];
    // per-lane table addresses; a group is 32 cluster rows further on (the group loop stays a loop: with all 32 units unrolled the
    // compiler kept one address register per far offset and spilled them)
    const _Float16* const ph = lds_h + j * stride + 8 * g;
    const _Float16* const pl = lds_l + j * stride + 8 * g;
    auto fetch = [&](int buf, int gr, int s) {
#pragma unroll
      for (int tile = 0; tile < 2; ++tile) {
        fh[buf][tile] = *reinterpret_cast<const f16x8_t*>(ph + (32 * gr + 16 * tile) * stride + 32 * s);
        fl[buf][tile] = *reinterpret_cast<const f16x8_t*>(pl + (32 * gr + 16 * tile) * stride + 32 * s);
      }
    };
    float best[2] = {-INFINITY, -INFINITY}, second[2] = {-INFINITY, -INFINITY};
    int bidx[2] = {0, 0};
    auto select4 = [&](int ab, int gr, int part) {   // part = 2 tile + fragment: the 4 clusters 32 gr + 16 tile + 4 q + e of this lane
      const int tile = part >> 1, f = part & 1;
      const int r0 = 32 * gr + 16 * tile;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = acc[ab][tile][f][e];
        const bool gt = v > best[f];           // ascending cluster order, strict '>': the first maximum stays
        second[f] = __builtin_amdgcn_fmed3f(best[f], second[f], v);
        best[f] = gt ? v : best[f];
        bidx[f] = gt ? (r0 + e) : bidx[f];     // the lane's 4 q is added after the loop
      }
    };
    f16x8_t cnb[2];                            // B fragment of the -|c|^2/2 step: the row's 2^(x_shift + e1) in k-slots 0..2 of the g = 0 lanes
#pragma unroll
    for (int f = 0; f < 2; ++f) {
#pragma unroll
      for (int q = 0; q < 8; ++q) cnb[f][q] = (_Float16)0.f;
      if constexpr (LUT) {   // six exact products: (three pieces of -|c|^2/2) x (two pieces of sqrt(d) 2^cn_e1)
        if (g == 0 && finite[f]) { cnb[f][0] = lut_f1[f]; cnb[f][1] = lut_f2[f]; cnb[f][2] = lut_f1[f]; cnb[f][3] = lut_f2[f]; cnb[f][4] = lut_f1[f]; cnb[f][5] = lut_f2[f]; }
      } else {
        const _Float16 pw = (g == 0 && finite[f]) ? (_Float16)ldexpf(1.f, x_shift[f] + a.cn_e1) : (_Float16)0.f;
        cnb[f][0] = pw; cnb[f][1] = pw; cnb[f][2] = pw;
      }
    }
    const f32x4v zero4 = {0.f, 0.f, 0.f, 0.f};
    fetch(0, 0, 0);
#pragma unroll 1
    for (int gp = 0; gp < 4; ++gp) {
#pragma unroll
      for (int ab = 0; ab < 2; ++ab) {         // group gr = 2 gp + ab accumulates in set ab
        const int gr = 2 * gp + ab;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int buf = s & 1;
          if (s < 3) fetch(buf ^ 1, gr, s + 1);
          else if (gr < 7) fetch(buf ^ 1, gr + 1, 0);
          __builtin_amdgcn_sched_barrier(0);     // the fetch stays AHEAD of this step's MFMAs
          // the three products of an element go to the same accumulator in a fixed order; consecutive MFMAs never share an accumulator
          if constexpr (NP >= 3) {
#pragma unroll
            for (int tile = 0; tile < 2; ++tile)
#pragma unroll
              for (int f = 0; f < 2; ++f)
                acc[ab][tile][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[buf][tile], xh[f][s], s == 0 ? zero4 : acc[ab][tile][f], 0, 0, 0);
          }
          if constexpr (NP >= 2) {
#pragma unroll
            for (int tile = 0; tile < 2; ++tile)
#pragma unroll
              for (int f = 0; f < 2; ++f)
                acc[ab][tile][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[buf][tile], xl[f][s], (NP == 2 && s == 0) ? zero4 : acc[ab][tile][f], 0, 0, 0);
          }
#pragma unroll
          for (int tile = 0; tile < 2; ++tile)
#pragma unroll
            for (int f = 0; f < 2; ++f)
              acc[ab][tile][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[buf][tile], xh[f][s], (NP == 1 && s == 0) ? zero4 : acc[ab][tile][f], 0, 0, 0);
          if (s == 3) {                          // - |c|^2 / 2 . 2^(shifts), after every product of the group
#pragma unroll
            for (int tile = 0; tile < 2; ++tile) {
              const f16x4_t pk = *reinterpret_cast<const f16x4_t*>(lds_k + 4 * (32 * gr + 16 * tile + j));
              f16x8_t ca;
#pragma unroll
              for (int q = 0; q < 8; ++q) ca[q] = (_Float16)0.f;
              if constexpr (LUT) {
                if (g == 0) { ca[0] = pk[0]; ca[1] = pk[0]; ca[2] = pk[1]; ca[3] = pk[1]; ca[4] = pk[2]; ca[5] = pk[2]; }
              } else {
                if (g == 0) { ca[0] = pk[0]; ca[1] = pk[1]; ca[2] = pk[2]; }
              }
#pragma unroll
              for (int f = 0; f < 2; ++f)
                acc[ab][tile][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ca, cnb[f], acc[ab][tile][f], 0, 0, 0);
            }
          }
          if (gr > 0) select4(ab ^ 1, gr - 1, s);   // the previous group's selection, one (tile, fragment) part under each step's MFMAs
          __builtin_amdgcn_sched_barrier(0);         // keep the steps apart: hoisting every fetch to the top spills
        }
      }
    }
#pragma unroll
    for (int part = 0; part < 4; ++part) select4(1, 7, part);
    __builtin_amdgcn_s_barrier();

    // ---- per descriptor: back to v = |c|^2 - 2 x.c (smallest = nearest), the four q-lanes combined, settled or listed
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const float m2s = -2.f * ldexpf(1.f, -(x_shift[f] + a.c_shift));   // exact: a power of two (uint8 rows: x_shift = 0, v in units of 1 / sqrt(d))
      float b = best[f] * m2s, sc = second[f] * m2s;
      int bi = bidx[f] + 4 * g;
#pragma unroll
      for (int m = 16; m <= 32; m <<= 1) {
        const float ob = __shfl_xor(b, m, 64), os = __shfl_xor(sc, m, 64);
        const int oi = __shfl_xor(bi, m, 64);
        const bool take = ob < b || (ob == b && oi < bi);
        const float nb = take ? ob : b, loser = take ? b : ob;
        sc = fminf(fminf(sc, os), loser);
        b = nb;
        if (take) bi = oi;
      }
      const float xc = nx[f] * a.cmax;
      constexpr float conv_err = DescTraits<KIND>::rootsift ? 4.8e-7f : 0.f;
      const float cc = LUT ? a.cmax * a.cmax * lut_sd[f] : a.cmax * a.cmax;
      constexpr float drop_err = (3 - NP) * 4.9e-4f;
      const float eps = 2.f * (4.8e-7f + 2.4e-7f + 4.8e-5f + 7.7e-6f + 1e-9f + conv_err + drop_err) * xc * (1.f + sqrt_d * 1e-9f) + (LUT ? 6.0e-7f : 2.4e-7f) * (cc + 2.f * xc);
      const bool settled = !(sc <= b + 2.f * eps) && finite[f] && fabsf(b) <= 3.0e38f && bi < a.K;
      const int64_t row = blk * ASSIGN_ROWS + wave * 32 + 16 * f + j;
      const bool amb = g == 0 && rvalid[f] && !settled;
      const unsigned long long amask = __ballot(amb);
      unsigned int base = 0;
      if (amask != 0ull) {
        if (lane == 0) base = atomicAdd(&s_count, (unsigned int)__popcll(amask));
        base = __shfl(base, 0, 64);
      }
      if (g == 0 && rvalid[f]) {
        if (settled) a.labels[row] = bi;
        else my_rows[base + __popcll(amask & ((1ull << lane) - 1ull))] = row;
      }
    }
  }
  if (!pp_g1) __builtin_amdgcn_s_barrier();
  __syncthreads();
  if (threadIdx.x == 0) a.amb_count[blockIdx.x] = s_count;
}

template <int NT, int KIND>
static int launch_assign_nt(pvs_ctx* ctx, const AssignArgs& a, bool vec, size_t lds, int grid) {
  auto kv = assign_kernel<NT, KIND, true>;
  auto ks = assign_kernel<NT, KIND, false>;
  auto k = vec ? kv : ks;
  PVS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds));
  hipLaunchKernelGGL(k, dim3(grid), dim3(ASSIGN_THREADS), lds, ctx->stream, a);
  PVS_HIP(hipGetLastError());
  return PVS_OK;
}

template <int KIND>
static int launch_assign_kind(pvs_ctx* ctx, const AssignArgs& a, int nt, bool vec, size_t lds, int grid) {
  switch (nt) {
    case 8: return launch_assign_nt<8, KIND>(ctx, a, vec, lds, grid);
    case 4: return launch_assign_nt<4, KIND>(ctx, a, vec, lds, grid);
    case 2: return launch_assign_nt<2, KIND>(ctx, a, vec, lds, grid);
    default: return launch_assign_nt<1, KIND>(ctx, a, vec, lds, grid);
  }
}

template <int NT, int KIND>
static int launch_assign16_nt(pvs_ctx* ctx, const Assign16Args& p, bool vec, size_t lds, int grid) {
  auto kv = assign16_kernel<NT, KIND, true, 0>;
  auto ks = assign16_kernel<NT, KIND, false, 0>;
  auto k8 = assign16_kernel<NT, KIND, true, 8>;
  auto k = vec ? (p.D == 128 && p.D_pad16 == 128 ? k8 : kv) : ks;
  if constexpr (NT == 8) {   // measurement variant of the shape the benchmarks use: the 16 x 16 x 32 kernel (same lists, 8-10 % slower)
    if (k == k8 && p.nprod == 3 && p.shape16 && p.stamps == nullptr) k = assign16x_kernel<KIND, 3>;
  }
  if constexpr (NT == 8) {   // measurement variants of the shape the benchmarks use: two products / one product per (row, cluster)
    if (k == k8 && p.nprod == 2) k = assign16_kernel<NT, KIND, true, 8, false, 2>;
    if (k == k8 && p.nprod == 1) k = assign16_kernel<NT, KIND, true, 8, false, 1>;
  }
  if constexpr (NT == 8) {   // pvs_fused_profile(ctx, 1, ...): the stamped build of the shape the benchmarks use
    if (p.stamps != nullptr && k == k8) k = assign16_kernel<NT, KIND, true, 8, true>;
  }
  PVS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k, dim3(grid), dim3(ASSIGN_THREADS), lds, ctx->stream, p);
  PVS_HIP(hipGetLastError());
  return PVS_OK;
}
template <int KIND>
static int launch_assign16_kind(pvs_ctx* ctx, const Assign16Args& p, int nt, bool vec, size_t lds, int grid) {
  switch (nt) {   // the whole (padded) table is one block of 32 nt clusters
    case 8: return launch_assign16_nt<8, KIND>(ctx, p, vec, lds, grid);
    case 4: return launch_assign16_nt<4, KIND>(ctx, p, vec, lds, grid);
    case 2: return launch_assign16_nt<2, KIND>(ctx, p, vec, lds, grid);
    case 1: return launch_assign16_nt<1, KIND>(ctx, p, vec, lds, grid);
    default: PVS_FAIL(PVS_ERR_INVALID, "assignment prefilter: unexpected table padding");
  }
}
static int launch_assign16(pvs_ctx* ctx, const Assign16Args& p, int kind, int nt, bool vec, size_t lds, int grid) {
  switch (kind) {
    case PVS_DESC_F32: return launch_assign16_kind<PVS_DESC_F32>(ctx, p, nt, vec, lds, grid);
    case PVS_DESC_F32_ROOTSIFT: return launch_assign16_kind<PVS_DESC_F32_ROOTSIFT>(ctx, p, nt, vec, lds, grid);
    case PVS_DESC_U8_ROOTSIFT: return launch_assign16_kind<PVS_DESC_U8_ROOTSIFT>(ctx, p, nt, vec, lds, grid);
    default: PVS_FAIL(PVS_ERR_INVALID, "unknown descriptor kind %d", kind);
  }
}

int assign_tiles_for(int K) {
  const int k32 = (K + 31) / 32 * 32;
  return k32 > 128 ? 8 : (k32 > 64 ? 4 : (k32 > 32 ? 2 : 1));
}

int launch_assign(pvs_ctx* ctx, const pvs_codebook* cb, const void* d_desc, int kind, int64_t total, int ld,
                  int32_t* d_labels, const float2** rowstat_out) {
  if (rowstat_out) *rowstat_out = nullptr;
  if (total <= 0) return PVS_OK;
  const bool rs = kind != PVS_DESC_F32;
  if (rs && cb->D > ASSIGN_DCHUNK)
    PVS_FAIL(PVS_ERR_UNSUPPORTED, "fused RootSIFT needs D <= %d (got %d)", ASSIGN_DCHUNK, cb->D);
  const int nt = assign_tiles_for(cb->K);
  if (cb->K_pad % (32 * nt) != 0) PVS_FAIL(PVS_ERR_INVALID, "codebook padding does not match the tile count");
  AssignArgs a{d_desc, total, cb->D, ld, cb->d_cpad, cb->d_cnorm, cb->K_pad, cb->D_pad, d_labels, nullptr, nullptr, 0};
  const int esz = kind == PVS_DESC_U8_ROOTSIFT ? 1 : 4;
  // vector path: rows and 4-element groups are 16-B (f32) / 4-B (u8) aligned
  const bool vec = (cb->D % 4 == 0) && (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_desc) % (4 * esz)) == 0);
  const int cw = cb->D_pad < ASSIGN_DCHUNK ? cb->D_pad : ASSIGN_DCHUNK;
  const size_t lds = (size_t)(32 * nt) * (cw + 4) * 4 + (size_t)(32 * nt) * 4;
  const int64_t nblocks = (total + ASSIGN_ROWS - 1) / ASSIGN_ROWS;
  const int grid = (int)(nblocks < ctx->num_cu ? nblocks : ctx->num_cu);
  ScopedTimer tm(ctx, T_ASSIGN);
  // ---- prefilter pass (fp16 MFMA) when the whole table has an fp16 copy; it leaves the near ties to the exact kernel
  const bool pre = cb->d_c16 != nullptr && total >= 4096 && ctx->opt[PVS_OPT_ASSIGN_PREFILTER] != 0;
  if (pre) {
    const int64_t cap = (nblocks + grid - 1) / grid * ASSIGN_ROWS;   // a workgroup can list at most what it processes
    char* ws = nullptr;
    const size_t cnt_b = ((size_t)grid * 8 + 255) / 256 * 256;
    // uint8 rows: the prefilter forms every row's sum anyway and leaves (sum + 1e-7, 1 / that) for the aggregate pass,
    // which then converts elements without a reduction and a division per member row
    const bool stat = rowstat_out != nullptr && kind == PVS_DESC_U8_ROOTSIFT;
    const size_t list_b = ((size_t)grid * cap * 8 + 255) / 256 * 256;
    PVS_TRY(ws_reserve(ctx, 6, cnt_b + list_b + (stat ? (size_t)total * sizeof(float2) : 0), reinterpret_cast<void**>(&ws)));
    unsigned long long* cnt = reinterpret_cast<unsigned long long*>(ws);
    int64_t* rows = reinterpret_cast<int64_t*>(ws + cnt_b);
    float2* rowstat = stat ? reinterpret_cast<float2*>(ws + cnt_b + list_b) : nullptr;
    if (stat) *rowstat_out = rowstat;
    Assign16Args p{d_desc, total, cb->D, ld, static_cast<const _Float16*>(cb->d_c16), cb->d_cnorm, cb->K_pad, cb->D_pad16,
                   cb->c16_shift, cb->cmax, static_cast<const _Float16*>(cb->d_cnk), cb->cn_e1, cb->K, d_labels, rows, cnt, cap, rowstat,
                   ctx->d_fused_stamps, ctx->opt[PVS_OPT_ASSIGN_PREFILTER] == 2 ? 2 : (ctx->opt[PVS_OPT_ASSIGN_PREFILTER] == 3 ? 1 : 3),
                   ctx->opt[PVS_OPT_ASSIGN_PREFILTER] == 4 ? 1 : 0};   // .shape16
    const size_t lds16 = (size_t)2 * cb->K_pad * (128 + 8) * 2 + (size_t)cb->K_pad * 4 + (size_t)cb->K_pad * 8 + 1024;   // + the sqrt table of uint8 rows
    PVS_TRY(launch_assign16(ctx, p, kind, cb->K_pad / 32, vec, lds16, grid));
    a.rows = rows;
    a.nrows = cnt;
    a.rows_cap = cap;
  }
  switch (kind) {
    case PVS_DESC_F32: return launch_assign_kind<PVS_DESC_F32>(ctx, a, nt, vec, lds, grid);
    case PVS_DESC_F32_ROOTSIFT: return launch_assign_kind<PVS_DESC_F32_ROOTSIFT>(ctx, a, nt, vec, lds, grid);
    case PVS_DESC_U8_ROOTSIFT: return launch_assign_kind<PVS_DESC_U8_ROOTSIFT>(ctx, a, nt, vec, lds, grid);
    default: PVS_FAIL(PVS_ERR_INVALID, "unknown descriptor kind %d", kind);
  }
}

// =============================================================================== K2+K3 aggregate
struct AggArgs {
  const void* X;
  int D, ld;
  const int64_t* offsets;  // [n_images+1]
  const int32_t* labels;   // [total]
  const float* cent;       // [K][D]
  int K;
  float power, eps;
  int norm_mode;           // 0: general p, 1: L1, 2: L2, 3: +inf, 4: none (raw residual sums, training)
  float norm_p;
  float* out;              // [n_images][K*D]
  float* inv_norm;         // [n_images] or null
  const float2* rowstat;   // uint8 rows: (row sum + 1e-7, reciprocal) per descriptor from the assignment pass, or null
};

constexpr int AGG_THREADS = 256;
constexpr int AGG_WAVES = AGG_THREADS / 64;
constexpr int AGG_CHUNK = 4096;  // descriptors sorted per pass (u16 indices in LDS)

// One workgroup per image.  GROUP lanes cooperate on one cluster row; each lane owns NREG vectors of
// VW consecutive dims: dims (r*GROUP + gl)*VW ... for r < NREG.
// HIOCC (short images: a few rows per cluster): the member loop is then a chain of dependent L2 / HBM round trips per cluster,
// bound by how many workgroups a CU holds -- 64 registers let eight waves share a SIMD instead of five (batches of 4 rows).
// Long images prefer batches of 8 rows at five waves per SIMD.  Same arithmetic, same order: the same bits either way.
template <int KIND, int GROUP, int VW, int NREG, bool HIOCC = false>
__global__ __launch_bounds__(AGG_THREADS, (HIOCC ? 8 : 1)) void vlad_aggregate_kernel(AggArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int K = a.K, D = a.D;
  int* hist = reinterpret_cast<int*>(smem);              // [AGG_WAVES][K]  counts, then cursors
  int* start = hist + AGG_WAVES * K;                     // [K+1]
  float* rowsq = reinterpret_cast<float*>(start + K + 1);  // [K]
  int* scan_tmp = reinterpret_cast<int*>(rowsq + K);       // [AGG_WAVES] (all LDS stays in the dynamic region)
  uint16_t* order = reinterpret_cast<uint16_t*>(scan_tmp + AGG_WAVES);  // [AGG_CHUNK]

  const int img = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row0 = a.offsets[img];
  const int64_t n = a.offsets[img + 1] - row0;
  float* out_img = a.out + (int64_t)img * K * D;

  constexpr int NGROUPS = AGG_THREADS / GROUP;
  const int grp = tid / GROUP, gl = tid % GROUP;
  int kbits = 0;
  while ((1 << kbits) < K) ++kbits;

  const int64_t nchunks = n > 0 ? (n + AGG_CHUNK - 1) / AGG_CHUNK : 1;
  for (int64_t ch = 0; ch < nchunks; ++ch) {
    const int64_t cbase = row0 + ch * AGG_CHUNK;
    const int cn = (int)min((int64_t)AGG_CHUNK, n - ch * AGG_CHUNK);  // may be 0 for an empty image
    const bool last = (ch == nchunks - 1);

    // ---- 1. per-wave histograms (integer LDS atomics: order independent)
    for (int i = tid; i < AGG_WAVES * K; i += AGG_THREADS) hist[i] = 0;
    __syncthreads();
    const int per_wave = (cn + AGG_WAVES - 1) / AGG_WAVES;
    const int wbeg = min(cn, wave * per_wave), wend = min(cn, wbeg + per_wave);
    for (int i = wbeg + lane; i < wend; i += 64) atomicAdd(&hist[wave * K + a.labels[cbase + i]], 1);
    __syncthreads();

    // ---- 2. exclusive scan in (cluster major, wave minor) order -> start[k], cursors
    {
      int carry = 0;  // running total of all clusters before this tile of 256
      for (int k0 = 0; k0 < K; k0 += AGG_THREADS) {
        const int k = k0 + tid;
        int c[AGG_WAVES], tot = 0;
#pragma unroll
        for (int w = 0; w < AGG_WAVES; ++w) {
          c[w] = k < K ? hist[w * K + k] : 0;
          tot += c[w];
        }
        // inclusive scan of tot across the 256 threads
        int incl = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int o = __shfl_up(incl, d, 64);
          if (lane >= d) incl += o;
        }
        if (lane == 63) scan_tmp[wave] = incl;
        __syncthreads();
        int wave_off = 0;
        for (int w = 0; w < wave; ++w) wave_off += scan_tmp[w];
        int tile_total = 0;
        for (int w = 0; w < AGG_WAVES; ++w) tile_total += scan_tmp[w];
        int excl = carry + wave_off + incl - tot;
        if (k < K) {
          start[k] = excl;
#pragma unroll
          for (int w = 0; w < AGG_WAVES; ++w) {
            hist[w * K + k] = excl;
            excl += c[w];
          }
        }
        carry += tile_total;
        __syncthreads();
      }
      if (tid == 0) start[K] = cn;
    }
    __syncthreads();

    // ---- 3. stable placement: each wave walks its contiguous quarter in order, 64 at a time;
    //         same-label lanes are found with kbits ballots (match-any), rank = #lower lanes
    for (int i0 = wbeg; i0 < wend; i0 += 64) {
      const int i = i0 + lane;
      const bool v = i < wend;
      const int lab = v ? a.labels[cbase + i] : -1;
      unsigned long long peers = __ballot(v);
      for (int b = 0; b < kbits; ++b) {
        const unsigned long long bal = __ballot((lab >> b) & 1);
        peers &= ((lab >> b) & 1) ? bal : ~bal;
      }
      if (v) {
        const unsigned long long lower = peers & ((1ull << lane) - 1ull);
        const int rank = __popcll(lower);
        const int basepos = hist[wave * K + lab];
        order[basepos + rank] = (uint16_t)i;
        // the highest peer lane advances the cursor after every peer has read it (one wave: lockstep)
        if ((peers >> lane) == 1ull) hist[wave * K + lab] = basepos + rank + 1;
      }
    }
    __syncthreads();

    // ---- 4. one lane-group per cluster: sequential fp32 sum of (x - c) in descriptor order
    for (int k = grp; k < K; k += NGROUPS) {
      const int s = start[k], e = start[k + 1];
      float acc[NREG][VW], c[NREG][VW];
#pragma unroll
      for (int r = 0; r < NREG; ++r) {
        const int d0 = (r * GROUP + gl) * VW;
#pragma unroll
        for (int q = 0; q < VW; ++q) {
          const bool in = d0 + q < D;
          c[r][q] = in ? a.cent[(int64_t)k * D + d0 + q] : 0.f;
          acc[r][q] = (ch > 0 && in) ? out_img[(int64_t)k * D + d0 + q] : 0.f;  // continue a long image
        }
      }
      // members in descriptor order; the loads of a batch of rows are issued before the first of them is added (one load in
      // flight per lane group leaves the kernel latency-bound), the additions themselves stay strictly in order.  The batch
      // is as long as the rows left (8 / 4 / 2): with 512 rows over 256 clusters most clusters have one to three members,
      // and this loop is instruction-bound -- slots of a batch that hold no row still cost their address arithmetic.
      constexpr bool PACKED = KIND == PVS_DESC_U8_ROOTSIFT && VW == 4 && NREG == 1;
      constexpr int PFMAX = HIOCC ? (DescTraits<KIND>::rootsift && !PACKED ? 2 : 4)
                                  : (PACKED ? 8 : (DescTraits<KIND>::rootsift ? 2 : (NREG == 1 ? 8 : (NREG == 2 ? 4 : 2))));
      const bool packed = PACKED && a.rowstat != nullptr;
      auto batch = [&](auto nc, int p0) {
        constexpr int N = decltype(nc)::value;
        if constexpr (PACKED) {
          if (packed) {
            // uint8 rows with the row statistics of the assignment pass: a member row is ONE register per lane until it is
            // converted, and the conversion needs neither the 32-lane reduction nor the division per row.  Same (d, r), same
            // arithmetic: the same bits as the general branch.
            const int d0 = gl * 4;
            uint32_t w[N];
            float2 st[N];
#pragma unroll
            for (int u = 0; u < N; ++u) {
              const int64_t row = cbase + order[p0 + u < e ? p0 + u : p0];
              w[u] = d0 < D ? *reinterpret_cast<const uint32_t*>(static_cast<const uint8_t*>(a.X) + row * a.ld + d0) : 0u;
              st[u] = a.rowstat[row];
            }
#pragma unroll
            for (int u = 0; u < N; ++u) {
              if (p0 + u < e) {   // uniform over the lane group
                const RootsiftRow<KIND> rr(st[u].x, st[u].y);
                acc[0][0] += (rr(float(w[u] & 0xffu)) - c[0][0]);
                acc[0][1] += (rr(float((w[u] >> 8) & 0xffu)) - c[0][1]);
                acc[0][2] += (rr(float((w[u] >> 16) & 0xffu)) - c[0][2]);
                acc[0][3] += (rr(float(w[u] >> 24)) - c[0][3]);
              }
            }
            return;
          }
        }
        float x[N][NREG][VW];
#pragma unroll
        for (int u = 0; u < N; ++u) {
          const bool live = p0 + u < e;
          const int64_t row = cbase + order[live ? p0 + u : p0];
#pragma unroll
          for (int r = 0; r < NREG; ++r) {
            const int d0 = (r * GROUP + gl) * VW;
            if constexpr (VW == 4) {
              float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
              if (live && d0 < D) t = load4<KIND>(a.X, row, a.ld, d0);
              x[u][r][0] = t.x; x[u][r][1] = t.y; x[u][r][2] = t.z; x[u][r][3] = t.w;
            } else {
              x[u][r][0] = (live && d0 < D) ? load1<KIND>(a.X, row, a.ld, d0) : 0.f;
            }
          }
        }
#pragma unroll
        for (int u = 0; u < N; ++u) {
          if (p0 + u < e) {   // uniform over the lane group
            if constexpr (DescTraits<KIND>::rootsift) {
              float sm = 0.f;
#pragma unroll
              for (int r = 0; r < NREG; ++r)
#pragma unroll
                for (int q = 0; q < VW; ++q) sm += x[u][r][q];
              sm = wave_sum_xor(sm, GROUP);
              const RootsiftRow<KIND> rr(sm);
#pragma unroll
              for (int r = 0; r < NREG; ++r)
#pragma unroll
                for (int q = 0; q < VW; ++q) x[u][r][q] = rr(x[u][r][q]);
            }
#pragma unroll
            for (int r = 0; r < NREG; ++r)
#pragma unroll
              for (int q = 0; q < VW; ++q) acc[r][q] += (x[u][r][q] - c[r][q]);
          }
        }
      };
      {
        int p0 = s;
        if constexpr (PFMAX >= 8) {
          for (; e - p0 > 4; p0 += 8) batch(std::integral_constant<int, 8>{}, p0);
        }
        if constexpr (PFMAX >= 4) {
          for (; e - p0 > 2; p0 += 4) batch(std::integral_constant<int, 4>{}, p0);
        }
        for (; p0 < e; p0 += 2) batch(std::integral_constant<int, 2>{}, p0);
      }

      if (!last) {
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
          const int d0 = (r * GROUP + gl) * VW;
#pragma unroll
          for (int q = 0; q < VW; ++q)
            if (d0 + q < D) out_img[(int64_t)k * D + d0 + q] = acc[r][q];
        }
        continue;
      }

      if (a.norm_mode == 4) {  // training pass: the raw residual sums leave as they are
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
          const int d0 = (r * GROUP + gl) * VW;
#pragma unroll
          for (int q = 0; q < VW; ++q)
            if (d0 + q < D) out_img[(int64_t)k * D + d0 + q] = acc[r][q];
        }
        if (gl == 0) rowsq[k] = 0.f;
        continue;
      }

      // ---- K3: power norm, per-cluster norm + eps, divide
      float part = 0.f;
#pragma unroll
      for (int r = 0; r < NREG; ++r)
#pragma unroll
        for (int q = 0; q < VW; ++q) {
          const int d = (r * GROUP + gl) * VW + q;
          acc[r][q] = d < D ? power_norm(acc[r][q], a.power) : 0.f;
          const float t = norm_accum(acc[r][q], a.norm_mode, a.norm_p);
          part = a.norm_mode == 3 ? fmaxf(part, t) : part + t;
        }
      float nrm;
      if constexpr (GROUP == 32) nrm = a.norm_mode == 3 ? half_max_xor(part) : half_sum_xor(part);   // same pairs as the shuffle butterfly: same bits
      else nrm = a.norm_mode == 3 ? wave_max_xor(part, GROUP) : wave_sum_xor(part, GROUP);
      if (a.norm_mode == 2) nrm = sqrt_rn(nrm);
      else if (a.norm_mode == 0) nrm = powf(nrm, 1.f / a.norm_p);
      const float den = nrm + a.eps;
      // |acc| <= nrm <= den; after the square-root power norm a non-zero |acc| is >= 2^-75 and a zero is +0: no per-element guard
      const DivByRow dv(den);
      const bool dguard = a.power != 0.5f;
      float sq = 0.f;
#pragma unroll
      for (int r = 0; r < NREG; ++r) {
        const int d0 = (r * GROUP + gl) * VW;
        float o[VW];
#pragma unroll
        for (int q = 0; q < VW; ++q) {
          o[q] = dv(acc[r][q], dguard);
          if (d0 + q < D) sq += o[q] * o[q];
        }
        if constexpr (VW == 4) {
          if (d0 < D) *reinterpret_cast<float4*>(out_img + (int64_t)k * D + d0) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
          if (d0 < D) out_img[(int64_t)k * D + d0] = o[0];
        }
      }
      if constexpr (GROUP == 32) sq = half_sum_xor(sq);
      else sq = wave_sum_xor(sq, GROUP);
      if (gl == 0) rowsq[k] = sq;
    }
    __syncthreads();
  }

  // ---- global 1/||row||_2 for the cosine step (sklearn normalize: zero norm -> 1), fixed order
  if (a.inv_norm != nullptr && wave == 0) {
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s += rowsq[k];
    s = wave_sum_xor(s, 64);
    if (lane == 0) a.inv_norm[img] = s > 0.f ? 1.f / sqrtf(s) : 1.f;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Streaming variant for tables whose K x D accumulators fit in LDS (K D <= 32768 floats, D <= 128, D % 4 == 0): the image's
// rows are read IN DESCRIPTOR ORDER (the block of an image is one contiguous HBM range) instead of cluster by cluster.
// 16 units (half-waves) split the clusters by `label & 15`; a stable counting sort on that 4-bit key gives every unit its
// rows in descriptor order, and the unit adds them one after the other into the cluster's accumulator row in LDS
// (read - add - write of one wave execute in order), so every cluster still sums its members sequentially in descriptor
// order: the same fp32 result as vlad_aggregate_kernel, bit for bit.  One workgroup (8 waves) per image, one per CU.
constexpr int ST_THREADS = 512;
constexpr int ST_WAVES = ST_THREADS / 64;
constexpr int ST_UNITS = 16;
constexpr int ST_PF = 8;

template <int KIND>
__global__ __launch_bounds__(ST_THREADS) void vlad_stream_kernel(AggArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int K = a.K, D = a.D;
  float* accs = reinterpret_cast<float*>(smem);                       // [K][D]
  int* hist = reinterpret_cast<int*>(accs + (size_t)K * D);           // [ST_WAVES][ST_UNITS] counts, then cursors
  int* start = hist + ST_WAVES * ST_UNITS;                            // [ST_UNITS + 1]
  float* rowsq = reinterpret_cast<float*>(start + ST_UNITS + 1);      // [K]
  uint16_t* order = reinterpret_cast<uint16_t*>(rowsq + K);           // [AGG_CHUNK] row of the chunk, unit by unit
  uint16_t* olab = order + AGG_CHUNK;                                 // [AGG_CHUNK] its label

  const int img = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int unit = tid >> 5, gl = tid & 31, d0 = 4 * gl;
  const bool dlive = d0 < D;
  const int64_t row0 = a.offsets[img];
  const int64_t n = a.offsets[img + 1] - row0;
  float* out_img = a.out + (int64_t)img * K * D;

  for (int i = tid * 4; i < K * D; i += ST_THREADS * 4) *reinterpret_cast<float4*>(accs + i) = make_float4(0.f, 0.f, 0.f, 0.f);

  const int64_t nchunks = (n + AGG_CHUNK - 1) / AGG_CHUNK;
  for (int64_t ch = 0; ch < nchunks; ++ch) {
    const int64_t cbase = row0 + ch * AGG_CHUNK;
    const int cn = (int)min((int64_t)AGG_CHUNK, n - ch * AGG_CHUNK);
    // ---- 1. per-wave histograms of the unit key
    if (tid < ST_WAVES * ST_UNITS) hist[tid] = 0;
    __syncthreads();
    const int per_wave = (cn + ST_WAVES - 1) / ST_WAVES;
    const int wbeg = min(cn, wave * per_wave), wend = min(cn, wbeg + per_wave);
    for (int i = wbeg + lane; i < wend; i += 64) atomicAdd(&hist[wave * ST_UNITS + (a.labels[cbase + i] & (ST_UNITS - 1))], 1);
    __syncthreads();
    // ---- 2. exclusive scan, key major / wave minor
    if (wave == 0) {      // lanes 0..15: one key each, prefix over the keys by shuffles
      const int u = lane & (ST_UNITS - 1);
      int c[ST_WAVES], tot = 0;
#pragma unroll
      for (int w = 0; w < ST_WAVES; ++w) {
        c[w] = hist[w * ST_UNITS + u];
        tot += c[w];
      }
      int incl = tot;
#pragma unroll
      for (int d = 1; d < ST_UNITS; d <<= 1) {
        const int o = __shfl_up(incl, d, ST_UNITS);
        if (u >= d) incl += o;
      }
      if (lane < ST_UNITS) {
        int run = incl - tot;
        start[u] = run;
#pragma unroll
        for (int w = 0; w < ST_WAVES; ++w) {
          hist[w * ST_UNITS + u] = run;
          run += c[w];
        }
        if (u == ST_UNITS - 1) start[ST_UNITS] = run;
      }
    }
    __syncthreads();
    // ---- 3. stable placement (as in vlad_aggregate_kernel, on the 4-bit key)
    for (int i0 = wbeg; i0 < wend; i0 += 64) {
      const int i = i0 + lane;
      const bool v = i < wend;
      const int label = v ? a.labels[cbase + i] : -1;
      const int key = v ? (label & (ST_UNITS - 1)) : -1;
      unsigned long long peers = __ballot(v);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const unsigned long long bal = __ballot((key >> b) & 1);
        peers &= ((key >> b) & 1) ? bal : ~bal;
      }
      if (v) {
        const int rank = __popcll(peers & ((1ull << lane) - 1ull));
        const int basepos = hist[wave * ST_UNITS + key];
        order[basepos + rank] = (uint16_t)i;
        olab[basepos + rank] = (uint16_t)label;
        if ((peers >> lane) == 1ull) hist[wave * ST_UNITS + key] = basepos + rank + 1;
      }
    }
    __syncthreads();
    // ---- 4. every unit walks its rows in descriptor order: two register sets of ST_PF rows, the loads of one set are in
    //         flight while the other is added
    const int s = start[unit], e = start[unit + 1];
    float4 xa[ST_PF], ca[ST_PF], xb[ST_PF], cb[ST_PF];
    int la[ST_PF], lb[ST_PF];
    // requests are unconditional (a position past the unit's end reads the unit's last row again, a lane past D the last
    // four dims): no branch between a request and the additions that wait for it, so the counters stay exact
    const int dl = dlive ? d0 : D - 4;
    auto request = [&](int p0, float4 (&x)[ST_PF], float4 (&c)[ST_PF], int (&lab)[ST_PF]) {
#pragma unroll
      for (int u = 0; u < ST_PF; ++u) {
        const int pos = min(p0 + u, e - 1);
        const int64_t row = cbase + order[pos];
        lab[u] = olab[pos];
        x[u] = load4<KIND>(a.X, row, a.ld, dl);
        c[u] = *reinterpret_cast<const float4*>(a.cent + (int64_t)lab[u] * D + dl);
      }
    };
    auto consume = [&](int p0, float4 (&x)[ST_PF], const float4 (&c)[ST_PF], const int (&lab)[ST_PF]) {
#pragma unroll
      for (int u = 0; u < ST_PF; ++u) {
        if (p0 + u < e) {   // uniform over the unit
          if constexpr (DescTraits<KIND>::rootsift) {
            float sm = dlive ? (x[u].x + x[u].y) + (x[u].z + x[u].w) : 0.f;   // integer-valued: exact in any order
            sm = wave_sum_xor(sm, 32);
            const RootsiftRow<KIND> rr(sm);
            x[u].x = rr(x[u].x); x[u].y = rr(x[u].y);
            x[u].z = rr(x[u].z); x[u].w = rr(x[u].w);
          }
          if (dlive) {
            float4* ap = reinterpret_cast<float4*>(accs + lab[u] * D + d0);
            float4 t = *ap;
            t.x += (x[u].x - c[u].x); t.y += (x[u].y - c[u].y); t.z += (x[u].z - c[u].z); t.w += (x[u].w - c[u].w);
            *ap = t;
          }
        }
      }
    };
    if (s < e) {
      request(s, xa, ca, la);
      for (int p0 = s; p0 < e; p0 += 2 * ST_PF) {
        request(p0 + ST_PF, xb, cb, lb);
        __builtin_amdgcn_sched_barrier(0);
        consume(p0, xa, ca, la);
        __builtin_amdgcn_sched_barrier(0);
        request(p0 + 2 * ST_PF, xa, ca, la);
        __builtin_amdgcn_sched_barrier(0);
        consume(p0 + ST_PF, xb, cb, lb);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  }
  if (nchunks == 0) __syncthreads();

  // ---- K3 per cluster row (one unit per row), then the whole-vector 1 / ||.||_2
  for (int k = unit; k < K; k += ST_UNITS) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (dlive) {
      const float4 t = *reinterpret_cast<const float4*>(accs + k * D + d0);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    if (a.norm_mode == 4) {
      if (dlive) *reinterpret_cast<float4*>(out_img + (int64_t)k * D + d0) = make_float4(v[0], v[1], v[2], v[3]);
      if (gl == 0) rowsq[k] = 0.f;
      continue;
    }
    float part = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v[q] = dlive ? power_norm(v[q], a.power) : 0.f;
      const float t = norm_accum(v[q], a.norm_mode, a.norm_p);
      part = a.norm_mode == 3 ? fmaxf(part, t) : part + t;
    }
    float nrm = a.norm_mode == 3 ? half_max_xor(part) : half_sum_xor(part);
    if (a.norm_mode == 2) nrm = sqrt_rn(nrm);
    else if (a.norm_mode == 0) nrm = powf(nrm, 1.f / a.norm_p);
    const float den = nrm + a.eps;
    const DivByRow dv(den);
    const bool dguard = a.power != 0.5f;
    float sq = 0.f, o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      o[q] = dv(v[q], dguard);
      if (dlive) sq += o[q] * o[q];
    }
    if (dlive) *reinterpret_cast<float4*>(out_img + (int64_t)k * D + d0) = make_float4(o[0], o[1], o[2], o[3]);
    sq = half_sum_xor(sq);
    if (gl == 0) rowsq[k] = sq;
  }
  __syncthreads();
  if (a.inv_norm != nullptr && wave == 0) {
    float s2 = 0.f;
    for (int k = lane; k < K; k += 64) s2 += rowsq[k];
    s2 = wave_sum_xor(s2, 64);
    if (lane == 0) a.inv_norm[img] = s2 > 0.f ? 1.f / sqrtf(s2) : 1.f;
  }
}

template <int KIND>
static int launch_stream_inst(pvs_ctx* ctx, const AggArgs& a, int64_t n_images, size_t lds) {
  auto k = vlad_stream_kernel<KIND>;
  PVS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k, dim3((unsigned)n_images), dim3(ST_THREADS), lds, ctx->stream, a);
  PVS_HIP(hipGetLastError());
  return PVS_OK;
}

template <int KIND, int GROUP, int VW, int NREG, bool HIOCC = false>
static int launch_agg_inst(pvs_ctx* ctx, const AggArgs& a, int64_t n_images, size_t lds) {
  auto k = vlad_aggregate_kernel<KIND, GROUP, VW, NREG, HIOCC>;
  PVS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds));
  hipLaunchKernelGGL(k, dim3((unsigned)n_images), dim3(AGG_THREADS), lds, ctx->stream, a);
  PVS_HIP(hipGetLastError());
  return PVS_OK;
}

template <int KIND>
static int launch_agg_kind(pvs_ctx* ctx, const AggArgs& a, int64_t n_images, size_t lds, bool vec, bool short_images) {
  const int D = a.D;
  if (vec) {  // 32 lanes x float4 = 128 dims per register step
    const int nreg = (D + 127) / 128;
    if (nreg <= 1 && short_images) return launch_agg_inst<KIND, 32, 4, 1, true>(ctx, a, n_images, lds);
    if (nreg <= 1) return launch_agg_inst<KIND, 32, 4, 1>(ctx, a, n_images, lds);
    if (nreg <= 2) return launch_agg_inst<KIND, 32, 4, 2>(ctx, a, n_images, lds);
    if (nreg <= 4) return launch_agg_inst<KIND, 32, 4, 4>(ctx, a, n_images, lds);
    if (nreg <= 8) return launch_agg_inst<KIND, 32, 4, 8>(ctx, a, n_images, lds);
  } else {    // 64 lanes x 1 float
    const int nreg = (D + 63) / 64;
    if (nreg <= 2) return launch_agg_inst<KIND, 64, 1, 2>(ctx, a, n_images, lds);
    if (nreg <= 4) return launch_agg_inst<KIND, 64, 1, 4>(ctx, a, n_images, lds);
    if (nreg <= 8) return launch_agg_inst<KIND, 64, 1, 8>(ctx, a, n_images, lds);
    if (nreg <= 16) return launch_agg_inst<KIND, 64, 1, 16>(ctx, a, n_images, lds);
  }
  PVS_FAIL(PVS_ERR_UNSUPPORTED, "descriptor dimension %d too large for the VLAD aggregate kernel (max 1024)", D);
}

int launch_vlad_aggregate(pvs_ctx* ctx, const pvs_codebook* cb, const void* d_desc, int kind, int ld,
                          const int64_t* d_offsets, int64_t n_images, const int32_t* d_labels,
                          const pvs_norm_params& prm, float* d_out, float* d_inv_norm, bool raw, const float2* rowstat,
                          int64_t total_hint) {
  if (n_images <= 0) return PVS_OK;
  if (cb->K > 2048) PVS_FAIL(PVS_ERR_UNSUPPORTED, "K = %d exceeds the VLAD aggregate kernel limit (2048)", cb->K);
  if (n_images > 0x7fffffffLL) PVS_FAIL(PVS_ERR_UNSUPPORTED, "too many images in one call");
  AggArgs a{};
  a.X = d_desc; a.D = cb->D; a.ld = ld; a.offsets = d_offsets; a.labels = d_labels; a.cent = cb->d_cent;
  a.K = cb->K; a.power = (float)prm.power_norm_weight; a.eps = (float)prm.epsilon;
  const double ord = prm.norm_order;
  if (std::isnan(ord) || ord <= 0.0) PVS_FAIL(PVS_ERR_UNSUPPORTED, "norm_order must be > 0 or +inf (got %g)", ord);
  a.norm_mode = raw ? 4 : (std::isinf(ord) ? 3 : (ord == 2.0 ? 2 : (ord == 1.0 ? 1 : 0)));
  a.norm_p = (float)ord;
  a.out = d_out; a.inv_norm = d_inv_norm;
  a.rowstat = kind == PVS_DESC_U8_ROOTSIFT ? rowstat : nullptr;
  const int esz = kind == PVS_DESC_U8_ROOTSIFT ? 1 : 4;
  const bool vec = (cb->D % 4 == 0) && (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_desc) % (4 * esz)) == 0) &&
                   ((reinterpret_cast<uintptr_t>(d_out) % 16) == 0);
  const size_t lds = (size_t)(AGG_WAVES * cb->K + cb->K + 1) * 4 + (size_t)cb->K * 4 + (size_t)AGG_WAVES * 4 + (size_t)AGG_CHUNK * 2 + 16;
  ScopedTimer tm(ctx, T_AGGREGATE);
  // accumulators in LDS, rows streamed in descriptor order (see vlad_stream_kernel): opt-in: pvs_set_option(PVS_OPT_VLAD_PATH, 2)
  const size_t lds_s = (size_t)cb->K * cb->D * 4 + (size_t)(ST_WAVES * ST_UNITS + ST_UNITS + 1) * 4 + (size_t)cb->K * 4 + (size_t)AGG_CHUNK * 4 + 16;
  const bool stream_ok = ctx->opt[PVS_OPT_VLAD_PATH] == 2;
  if (stream_ok && vec && cb->D <= 128 && (size_t)cb->K * cb->D <= 32768 && lds_s <= 160 * 1024 - 512 &&
      (reinterpret_cast<uintptr_t>(cb->d_cent) % 16) == 0) {
    switch (kind) {
      case PVS_DESC_F32: return launch_stream_inst<PVS_DESC_F32>(ctx, a, n_images, lds_s);
      case PVS_DESC_F32_ROOTSIFT: return launch_stream_inst<PVS_DESC_F32_ROOTSIFT>(ctx, a, n_images, lds_s);
      case PVS_DESC_U8_ROOTSIFT: return launch_stream_inst<PVS_DESC_U8_ROOTSIFT>(ctx, a, n_images, lds_s);
      default: PVS_FAIL(PVS_ERR_INVALID, "unknown descriptor kind %d", kind);
    }
  }
  // fewer than ~3.5 rows per cluster on average (when the caller knows the row count): the eight-waves-per-SIMD variant
  const int variant = ctx->opt[PVS_OPT_AGG_VARIANT];
  const bool short_images = variant == 1 || (variant != 2 && total_hint > 0 && (double)total_hint < 3.5 * (double)n_images * (double)cb->K);
  switch (kind) {
    case PVS_DESC_F32: return launch_agg_kind<PVS_DESC_F32>(ctx, a, n_images, lds, vec, short_images);
    case PVS_DESC_F32_ROOTSIFT: return launch_agg_kind<PVS_DESC_F32_ROOTSIFT>(ctx, a, n_images, lds, vec, short_images);
    case PVS_DESC_U8_ROOTSIFT: return launch_agg_kind<PVS_DESC_U8_ROOTSIFT>(ctx, a, n_images, lds, vec, short_images);
    default: PVS_FAIL(PVS_ERR_INVALID, "unknown descriptor kind %d", kind);
  }
}

}  // namespace pvs
