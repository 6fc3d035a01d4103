// Fused VLAD encode on gfx950: K1 (KMeans.predict) + K2 (residual sums) + K3 (normalisation) in ONE pass over the
// descriptors -- every descriptor row is read from HBM exactly once.  Headline table shape only: D = 128, 128 < K <= 256;
// everything else takes the two-kernel path of vlad.hip.
//
// Reference semantics (paths relative to the reference root) are those of vlad.hip:
//   K1  pyvisim/encoders/vlad.py:95 -> sklearn/cluster/_k_means_lloyd.pyx:168-218   argmin_j (|c_j|^2 - 2 x.c_j), first minimum
//   K2  vlad.py:98-104    V[label] += (x - c_label), sequentially in descriptor order, fp32
//   K3  vlad.py:106-111   sign(V)|V|^p, per-cluster norm + eps, divide, k-major flatten
//
// One persistent 512-thread workgroup per CU takes whole images from a queue and walks each image in stages of 64 rows:
//   P0  the 16 half-waves convert the staged rows (RootSIFT tail, row norm, power-of-two row scale, fp16 hi / lo split) and
//       store them in LDS: f32 rows (for K2 and the exact re-evaluation) + fp16 MFMA fragments; the next stage's rows are
//       requested from HBM into registers and land under the phases below
//   A   wave w owns clusters [32 w, 32 w + 32): its table fragments (fp16 hi / lo, 64 VGPRs) and its residual sums
//       (32 x 128 fp32, 64 VGPRs) live in registers for the whole launch.  Per 32-row tile 25 v_mfma_f32_32x32x16_f16:
//       16 for cl.xh + ch.xl, one that adds -|c|^2/2 (three exact fp16 pieces x the row scale), 8 for ch.xh, i.e.
//       s = 2^(cs+xs) (x.c - |c|^2/2): the LARGEST s is the nearest centre.  The scan over the 16 scores a lane holds is
//       v_and_or (cluster id into the 5 low mantissa bits), v_max, v_med3: instruction classes that issue next to the matrix
//       pipe (profiles/r02_coissue.txt: fp32 add / mul / fma do not)
//   B   every wave reduces the 8 per-wave candidates of all 64 rows (lane = row): label, runner-up, settled or not under the
//       proven margin (below); rows that are not settled are re-evaluated EXACTLY for every cluster inside the margin with
//       the fp32 recurrence of the exact kernel (assign_kernel: fma chain in the order 8t+e, 8t+4+e; vlad.hip) so the labels
//       are those of the exact kernel for every input; then each wave adds the rows of its clusters, in descriptor
//       order, into its register sums (wave-uniform switch on the cluster = static register indices)
//   E   after the image's last stage: power norm, per-cluster norm, divide, 512-B row stores, 1/||row||
//
// Margin.  With S = 2^(cs+xs), |x| <= nx, |c| <= cmax:
//   |s16/S - (x.c - |c|^2/2)| <= (3 2^-22 + 1e-9 + 256 2^-33) nx cmax            operand split, dropped cl.xl, flush, lo sums
//                               + 129 2^-23 (cmax^2/2 + 1.001 nx cmax)           fp32 accumulation of the 129 large terms
//   key truncation (5 bits)    <= 2 . 2^-18 (cmax^2/2 + nx cmax)                 both keys of a comparison
//   exact kernel               |v32 - v| <= 2 . 128 2^-24 nx cmax + 2^-24 (cmax^2 + 2 nx cmax)
//   => eps_v = 6.4e-5 nx cmax + 2.4e-5 cmax^2 bounds |v16 - v32| for every cluster; a row whose best key leads the runner-up
//      by more than eps_v S (i.e. 2 eps_v in v) is settled, every cluster whose key is within that distance is a candidate.
#include "common.hpp"
#include "desc_load.hpp"

namespace pvs {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));

constexpr int FU_THREADS = 512;
constexpr int FU_R = 64;            // rows per stage: two 32-row MFMA tiles
constexpr int FU_XS = 132;          // floats per staged fp32 row (528 B: 16 rows fall on 16 distinct 16-B bank slots)
constexpr int FU_TS = 1056;         // bytes between the k-steps of a tile's fragment image (1024 + 32: spreads the P0 stores)
constexpr int FU_TILE = 8 * FU_TS;  // one 32-row tile, hi or lo
constexpr int FU_UNUSABLE = 0x7fffffff;
constexpr int FU_KB = 4;            // rows of one wave added per batch in K2 (their loads are in flight together)

constexpr int FU_OFF_X32 = 0;                              // float [64][132]
constexpr int FU_OFF_XH = FU_OFF_X32 + FU_R * FU_XS * 4;   // fp16 fragments, hi: [2 tiles][8 k-steps][64 lanes][16 B]
constexpr int FU_OFF_XL = FU_OFF_XH + 2 * FU_TILE;         //                 lo
constexpr int FU_OFF_NX = FU_OFF_XL + 2 * FU_TILE;         // float [64]  |row| (upper bound)
constexpr int FU_OFF_XSH = FU_OFF_NX + 256;                // int   [64]  row scale exponent, FU_UNUSABLE for rows the prefilter cannot take
constexpr int FU_OFF_CAND = FU_OFF_XSH + 256;              // float2 [8 waves][64 rows]: best key, runner-up
constexpr int FU_OFF_ENT = FU_OFF_CAND + 8 * FU_R * 8;     // int   [512]   exact re-evaluation entries
constexpr int FU_OFF_ENTV = FU_OFF_ENT + 2048;             // float2 [512]  their results (value, cluster)
constexpr int FU_OFF_ROWSQ = FU_OFF_ENTV + 4096;           // float [256]
constexpr int FU_OFF_MISC = FU_OFF_ROWSQ + 1024;           // int [16]
constexpr int FU_LDS = FU_OFF_MISC + 64;

struct FusedArgs {
  const void* X;
  int ld;
  const int64_t* offsets;
  int64_t n_images;
  const _Float16* c16n;  // [2][256][128]  hi | lo of c 2^cs, natural dim order, zero rows for padded clusters
  const _Float16* cnk;   // [256][4]       three fp16 pieces of -|c|^2/2 2^(cs-e1) (padded clusters: -65504, 0, 0), 0
  const float* cnorm;    // [256]          |c|^2 (+inf padded)
  const float* cent;     // [K][128]
  const float* cpad;     // [256][128]     zero padded
  int K, c_shift, cn_e1;
  float cmax;
  float power, eps;
  int norm_mode;
  float norm_p;
  float* out;
  float* inv_norm;
  int32_t* labels;
  unsigned int* queue;   // next image to hand out (initialised to the grid size)
  unsigned long long* stamps;   // diagnostic build only: [16] cycle / event totals over all workgroups (pvs_fused_profile)
};

template <int KIND>
struct FuStage {   // the staged rows of one lane: 4 rows x 4 dims
  float4 v[4];
};
template <>
struct FuStage<PVS_DESC_U8_ROOTSIFT> {
  uint32_t v[4];
};

#define FU_CASES(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15) M(16) M(17) M(18) M(19) \
  M(20) M(21) M(22) M(23) M(24) M(25) M(26) M(27) M(28) M(29) M(30) M(31)

// DIAG: s_memtime stamps of wave 0 around the phases (P0, A, reduce, re-evaluation, K2, epilogue, image switch) summed over
// the workgroups into a.stamps[0..6], plus [8] stages, [9] stages with a re-evaluation, [10] entries, [11] rows not settled.
// The product instantiation (DIAG = false) contains no stamp.
template <int KIND, bool DIAG>
__global__ __launch_bounds__(FU_THREADS, 2) void vlad_fused_kernel(FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* const x32 = reinterpret_cast<float*>(smem + FU_OFF_X32);
  float* const s_nx = reinterpret_cast<float*>(smem + FU_OFF_NX);
  int* const s_xsh = reinterpret_cast<int*>(smem + FU_OFF_XSH);
  float2* const s_cand = reinterpret_cast<float2*>(smem + FU_OFF_CAND);
  int* const s_ent = reinterpret_cast<int*>(smem + FU_OFF_ENT);
  float2* const s_entv = reinterpret_cast<float2*>(smem + FU_OFF_ENTV);
  float* const s_rowsq = reinterpret_cast<float*>(smem + FU_OFF_ROWSQ);
  int* const s_misc = reinterpret_cast<int*>(smem + FU_OFF_MISC);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int hw = tid >> 5, g = tid & 31;   // P0: half-wave hw converts rows hw, 16 + hw, ...; lane g holds dims 4g .. 4g+3

  // ---- this wave's table fragments: cluster 32 wave + j, k-slots 8h .. 8h+7 of every 16-dim step
  f16x8_t tabh[8], tabl[8];
  {
    const _Float16* ph = a.c16n + (size_t)(32 * wave + j) * 128 + 8 * h;
    const _Float16* pl = ph + 256 * 128;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      tabh[t] = *reinterpret_cast<const f16x8_t*>(ph + 16 * t);
      tabl[t] = *reinterpret_cast<const f16x8_t*>(pl + 16 * t);
    }
  }
  f16x8_t cn_a;   // A fragment of the -|c|^2/2 step: pieces in k-slots 0..2 (half-wave 0), zeros elsewhere
  {
    const f16x4_t p = *reinterpret_cast<const f16x4_t*>(a.cnk + (size_t)(32 * wave + j) * 4);
#pragma unroll
    for (int q = 0; q < 8; ++q) cn_a[q] = (_Float16)0.f;
    if (h == 0) { cn_a[0] = p[0]; cn_a[1] = p[1]; cn_a[2] = p[2]; }
  }
  // the residual sums: every access below names its element with a literal index (macros, not loops), so the array is split
  // into 64 registers before any later pass can merge the 32 switch cases of K2 into one dynamically indexed access
  float acc[64];
#define FU_ZERO(c) acc[2 * c] = 0.f; acc[2 * c + 1] = 0.f;
  FU_CASES(FU_ZERO)

  const float eps_a = 6.4e-5f * a.cmax, eps_b = 2.4e-5f * a.cmax * a.cmax;
  unsigned long long dg_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dg_c[4] = {0, 0, 0, 0}, dg_last = 0;
  if constexpr (DIAG) dg_last = __builtin_amdgcn_s_memtime();
#define FU_STAMP(i)                                                  \
  if constexpr (DIAG) {                                              \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();    \
    dg_t[i] += now_ - dg_last;                                       \
    dg_last = now_;                                                  \
  }

  FuStage<KIND> stg;
  auto issue_loads = [&](int64_t rbase, int cnt) {   // rows rbase .. rbase + cnt - 1 of X -> registers (rows past cnt: zeros)
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int r = 16 * p + hw;
      if constexpr (KIND == PVS_DESC_U8_ROOTSIFT) {
        stg.v[p] = 0u;
        if (r < cnt) stg.v[p] = *reinterpret_cast<const uint32_t*>(static_cast<const uint8_t*>(a.X) + (rbase + r) * a.ld + 4 * g);
      } else {
        stg.v[p] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < cnt) stg.v[p] = *reinterpret_cast<const float4*>(static_cast<const float*>(a.X) + (rbase + r) * a.ld + 4 * g);
      }
    }
  };

  int64_t cur = blockIdx.x;
  if (cur < a.n_images) {
    const int64_t r0 = a.offsets[cur], n0 = a.offsets[cur + 1] - r0;
    if (n0 > 0) issue_loads(r0, (int)(n0 < FU_R ? n0 : FU_R));
  }

  while (cur < a.n_images) {
    const int64_t row0 = a.offsets[cur];
    const int64_t n = a.offsets[cur + 1] - row0;
    const int64_t nst = (n + FU_R - 1) / FU_R;
    float* const out_img = a.out + cur * (int64_t)a.K * 128;
    if (tid == 0) s_misc[0] = (int)atomicAdd(a.queue, 1u);   // the image after this one
    __syncthreads();
    const int64_t nxt = (int64_t)(unsigned int)s_misc[0];
    auto prefetch_next_image = [&]() {
      if (nxt < a.n_images) {
        const int64_t r0 = a.offsets[nxt], n0 = a.offsets[nxt + 1] - r0;
        if (n0 > 0) issue_loads(r0, (int)(n0 < FU_R ? n0 : FU_R));
      }
    };
    if (nst == 0) prefetch_next_image();
    FU_STAMP(6)

    for (int64_t s = 0; s < nst; ++s) {
      const int64_t sbase = row0 + s * FU_R;
      const int cnt = (int)((n - s * FU_R) < FU_R ? (n - s * FU_R) : FU_R);

      // ============================================================ P0: convert the staged rows into LDS
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int r = 16 * p + hw;
        float x[4];
        if constexpr (KIND == PVS_DESC_U8_ROOTSIFT) {
          const uint32_t w = stg.v[p];
          x[0] = float(w & 0xffu); x[1] = float((w >> 8) & 0xffu); x[2] = float((w >> 16) & 0xffu); x[3] = float(w >> 24);
        } else {
          x[0] = stg.v[p].x; x[1] = stg.v[p].y; x[2] = stg.v[p].z; x[3] = stg.v[p].w;
        }
        if constexpr (DescTraits<KIND>::rootsift) {
          float sm = (x[0] + x[1]) + (x[2] + x[3]);    // integer-valued rows: the sum is exact in any order
          sm = half_sum_xor(sm);
          const RootsiftRow<KIND> rr(sm);
#pragma unroll
          for (int q = 0; q < 4; ++q) x[q] = rr(x[q]);
        }
        float n2 = 0.f, amax = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          n2 = fmaf(x[q], x[q], n2);
          amax = fmaxf(amax, fabsf(x[q]));
        }
        n2 = half_sum_xor(n2);
        amax = half_max_xor(amax);
        const float nx = sqrtf(n2) * 1.0001f;
        int ex = 13;
        if (amax > 0.f) (void)frexpf(amax, &ex);
        int xsh = 13 - ex;                              // largest |x| 2^xsh in [2^12, 2^13)
        // usable: finite row, and 2^(xsh + e1) is a normal fp16 number (the row scale enters the -|c|^2/2 step as an fp16 factor)
        const bool usable = nx <= 3.0e38f && amax <= 3.0e38f && (xsh + a.cn_e1) >= -14 && (xsh + a.cn_e1) <= 15 && xsh >= -100 && xsh <= 100;
        if (!usable) xsh = 0;
        const float xs = ldexpf(1.f, xsh);
        f16x4_t hi, lo;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float v = x[q] * xs;
          const _Float16 hq = (_Float16)v;
          hi[q] = hq;
          lo[q] = (_Float16)(v - (float)hq);
        }
        *reinterpret_cast<float4*>(x32 + r * FU_XS + 4 * g) = make_float4(x[0], x[1], x[2], x[3]);
        // fragment image: k-step t = g >> 2 holds dims 16 t .. 16 t + 15, half-wave hh = (g >> 1) & 1 its k-slots 8 hh .. 8 hh + 7
        const int fo = (r >> 5) * FU_TILE + (g >> 2) * FU_TS + ((((g >> 1) & 1) * 32 + (r & 31)) * 16) + (g & 1) * 8;
        *reinterpret_cast<f16x4_t*>(smem + FU_OFF_XH + fo) = hi;
        *reinterpret_cast<f16x4_t*>(smem + FU_OFF_XL + fo) = lo;
        if (g == 0) {
          s_nx[r] = nx;
          s_xsh[r] = usable ? xsh : FU_UNUSABLE;
        }
      }
      // ---- request the next stage's rows (they land under phases A and B).  Fenced: hoisted to the last use of each staging
      // register, the requests would sit in front of this phase's remaining waits and expose a full HBM round trip per pass.
      __builtin_amdgcn_sched_barrier(0);
      if (s + 1 < nst) {
        const int64_t left = n - (s + 1) * FU_R;
        issue_loads(sbase + FU_R, (int)(left < FU_R ? left : FU_R));
      } else {
        prefetch_next_image();
      }
      __syncthreads();
      FU_STAMP(0)

      // ============================================================ A: scores of this wave's 32 clusters for all 64 rows
      {
        // One 32-row tile after the other: 16 MFMAs for the small products (their partial sums stay small, so do their rounding
        // errors), one for -|c|^2/2 2^(cs+xs) (three exact fp16 pieces of the table side times the row's power of two), 8 for ch.xh.
        // Fragment reads run one k-step ahead of the MFMAs that use them; the fences let vector / scalar ALU work cross (the scan
        // of tile 0 interleaves with the MFMAs of tile 1) but keep LDS reads and MFMAs in order -- hoisting ALL reads to the top
        // would push the table and sum registers into scratch.
        auto tile_scores = [&](int tile) -> f32x16 {
          f32x16 sc;
#pragma unroll
          for (int q = 0; q < 16; ++q) sc[q] = 0.f;
          const char* fh = smem + FU_OFF_XH + tile * FU_TILE + lane * 16;
          const char* fl = smem + FU_OFF_XL + tile * FU_TILE + lane * 16;
          f16x8_t bh = *reinterpret_cast<const f16x8_t*>(fh), bl = *reinterpret_cast<const f16x8_t*>(fl);
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            f16x8_t nh, nl = bl;
            if (t + 1 < 8) {
              nh = *reinterpret_cast<const f16x8_t*>(fh + (t + 1) * FU_TS);
              nl = *reinterpret_cast<const f16x8_t*>(fl + (t + 1) * FU_TS);
            } else {
              nh = *reinterpret_cast<const f16x8_t*>(fh);              // first k-step of the ch.xh pass
            }
            __builtin_amdgcn_sched_barrier(0x6);
            sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(tabl[t], bh, sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(tabh[t], bl, sc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0x6);
            bh = nh; bl = nl;
          }
          {
            const int xr = s_xsh[32 * tile + j];
            const _Float16 pw = (h == 0 && xr != FU_UNUSABLE) ? (_Float16)ldexpf(1.f, xr + a.cn_e1) : (_Float16)0.f;
            f16x8_t bb;
#pragma unroll
            for (int q = 0; q < 8; ++q) bb[q] = (_Float16)0.f;
            bb[0] = pw; bb[1] = pw; bb[2] = pw;
            sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(cn_a, bb, sc, 0, 0, 0);
          }
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            f16x8_t nh = bh;
            if (t + 1 < 8) nh = *reinterpret_cast<const f16x8_t*>(fh + (t + 1) * FU_TS);
            __builtin_amdgcn_sched_barrier(0x6);
            sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(tabh[t], bh, sc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0x6);
            bh = nh;
          }
          return sc;
        };
        // ---- scan: the lane holds, for row j of the tile, the scores of clusters (q & 3) + 8 (q >> 2) + 4 h of this wave
        auto scan = [&](const f32x16& sc, int tile) {
          float best = -INFINITY, second = -INFINITY;
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int low = 31 - ((q & 3) + 8 * (q >> 2));                     // 31 - cluster (half-wave 0); -4 for h = 1 below
            const float key = __int_as_float((__float_as_int(sc[q]) & ~31) | low);
            second = __builtin_amdgcn_fmed3f(best, second, key);               // best >= second: the median is the new runner-up
            best = fmaxf(best, key);
          }
          best = __int_as_float(__float_as_int(best) - 4 * h);                 // low bits = 31 - cluster for both half-waves
          const float ob = __shfl_xor(best, 32, 64), os = __shfl_xor(second, 32, 64);
          const float nb = fmaxf(best, ob);
          const float ns = fmaxf(fminf(best, ob), fmaxf(second, os));
          if (h == 0) s_cand[wave * FU_R + 32 * tile + j] = make_float2(nb, ns);
        };
        const f32x16 sc0 = tile_scores(0);
        scan(sc0, 0);
        __builtin_amdgcn_sched_barrier(0);
        const f32x16 sc1 = tile_scores(1);
        scan(sc1, 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
      FU_STAMP(1)

      // ============================================================ B: labels (lane = row, every wave the same), exact re-evaluation, sums
      int label;
      {
        const bool rvalid = lane < cnt;
        float t1 = -INFINITY, t2 = -INFINITY;
        int wb = 0;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
          const float2 c = s_cand[w * FU_R + lane];
          const bool gt = c.x > t1;                       // strict: equal keys keep the lower wave
          t2 = fmaxf(fmaxf(t2, c.y), gt ? t1 : c.x);
          wb = gt ? w : wb;
          t1 = gt ? c.x : t1;
        }
        label = 32 * wb + 31 - (__float_as_int(t1) & 31);
        const int xsh = s_xsh[lane];
        const bool usable = xsh != FU_UNUSABLE;
        const float margin = usable ? (eps_a * s_nx[lane] + eps_b) * ldexpf(1.f, a.c_shift + xsh) : 0.f;
        const bool settled = usable && label < a.K && (t1 - t2 > margin);
        const float thr = t1 - margin;
        const unsigned long long umask = __ballot(rvalid && !settled);
        FU_STAMP(2)
        if constexpr (DIAG) { dg_c[0] += 1; dg_c[3] += __popcll(umask); }
        if (umask != 0ull) {   // uniform over the workgroup: every wave reduced the same candidates
          // ---- entries: (row, wave) pairs whose best key is inside the margin; `all` when the wave's runner-up is too
          int cm = 0, am = 0;
          if (rvalid && !settled) {
            if (!usable || label >= a.K) {
              cm = 0xff; am = 0xff;                       // no usable prefilter scores: every cluster, exactly
            } else {
#pragma unroll
              for (int w = 0; w < 8; ++w) {
                const float2 c = s_cand[w * FU_R + lane];
                if (c.x >= thr) cm |= 1 << w;
                if (c.y >= thr) am |= 1 << w;
              }
            }
          }
          const int ne = __popc(cm);
          int incl = ne;
#pragma unroll
          for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
          }
          const int ebase = incl - ne;
          const int E = __shfl(incl, 63, 64);
          if constexpr (DIAG) { dg_c[1] += 1; dg_c[2] += E; }
          if (wave == 0 && ne > 0) {
            int e = ebase;
            for (int w = 0; w < 8; ++w)
              if (cm & (1 << w)) {
                const int sidx = 31 - (__float_as_int(s_cand[w * FU_R + lane].x) & 31);
                s_ent[e++] = lane | (w << 8) | (((am >> w) & 1) << 12) | (sidx << 16);
              }
          }
          __syncthreads();
          // ---- exact values: one entry per half-wave and pass; lane i of the half-wave takes cluster 32 w + i (or the single one)
          for (int e0 = 0; e0 < E; e0 += 16) {
            const int e = e0 + 2 * wave + h;
            const bool elive = e < E;
            const int ent = elive ? s_ent[e] : 0;
            const int er = ent & 0xff, ew = (ent >> 8) & 7, eall = (ent >> 12) & 1, esi = (ent >> 16) & 31;
            const int k = 32 * ew + (eall ? j : esi);
            const bool active = elive && (eall || j == 0);
            const float* xr = x32 + er * FU_XS;
            const float* cr = a.cpad + (size_t)k * 128;
            float dot = 0.f;
#pragma unroll 2
            for (int b8 = 0; b8 < 128; b8 += 8) {
              const float4 xa = *reinterpret_cast<const float4*>(xr + b8), xb = *reinterpret_cast<const float4*>(xr + b8 + 4);
              const float4 ca = *reinterpret_cast<const float4*>(cr + b8), cb = *reinterpret_cast<const float4*>(cr + b8 + 4);
              dot = fmaf(ca.x, xa.x, dot); dot = fmaf(cb.x, xb.x, dot);
              dot = fmaf(ca.y, xa.y, dot); dot = fmaf(cb.y, xb.y, dot);
              dot = fmaf(ca.z, xa.z, dot); dot = fmaf(cb.z, xb.z, dot);
              dot = fmaf(ca.w, xa.w, dot); dot = fmaf(cb.w, xb.w, dot);
            }
            const float v = fmaf(-2.f, dot, a.cnorm[k]);
            const bool ok = active && (v < INFINITY);          // the exact kernel's `v < best` never takes NaN or +inf
            float vv = ok ? v : INFINITY;
            int kk = ok ? k : 0x7fffffff;
#pragma unroll
            for (int m = 16; m >= 1; m >>= 1) {
              const float ov = __shfl_xor(vv, m, 64);
              const int okk = __shfl_xor(kk, m, 64);
              const bool take = ov < vv || (ov == vv && okk < kk);
              vv = take ? ov : vv;
              kk = take ? okk : kk;
            }
            if (elive && j == 0) s_entv[e] = make_float2(vv, __int_as_float(kk));
          }
          __syncthreads();
          if (rvalid && !settled) {
            float bv = INFINITY;
            int bk = 0x7fffffff;
            for (int i = 0; i < ne; ++i) {
              const float2 r = s_entv[ebase + i];
              const int rk = __float_as_int(r.y);
              if (r.x < bv || (r.x == bv && rk < bk)) { bv = r.x; bk = rk; }
            }
            label = bv < INFINITY ? bk : 0;     // nothing below +inf: the exact kernel's initial label
          }
        }
        FU_STAMP(3)
        if (a.labels != nullptr && wave == 0 && rvalid) a.labels[sbase + lane] = label;

        // ---- K2: this wave adds the rows of its clusters, in descriptor order; lane L owns dims 2L, 2L+1 of each cluster
        unsigned long long mine = __ballot(rvalid && (label >> 5) == wave);
        while (mine != 0ull) {
          int rr[FU_KB], ll[FU_KB];
          int cntb = 0;
#pragma unroll
          for (int u = 0; u < FU_KB; ++u) {
            const bool live = mine != 0ull;
            const int r = live ? (int)__builtin_ctzll(mine) : 0;
            if (live) { mine &= mine - 1ull; ++cntb; }
            rr[u] = r;
            ll[u] = __builtin_amdgcn_readlane(label, r);
          }
          float2 xv[FU_KB], cv[FU_KB];
#pragma unroll
          for (int u = 0; u < FU_KB; ++u) {
            xv[u] = *reinterpret_cast<const float2*>(x32 + rr[u] * FU_XS + 2 * lane);
            cv[u] = *reinterpret_cast<const float2*>(a.cent + (size_t)ll[u] * 128 + 2 * lane);
          }
#pragma unroll
          for (int u = 0; u < FU_KB; ++u) {
            if (u < cntb) {
              const float dx = xv[u].x - cv[u].x, dy = xv[u].y - cv[u].y;
#define FU_ADD(c) case c: acc[2 * c] += dx; acc[2 * c + 1] += dy; break;
              switch (ll[u] & 31) { FU_CASES(FU_ADD) }
#undef FU_ADD
            }
          }
        }
      }
      __syncthreads();   // LDS is rewritten by the next stage's P0
      FU_STAMP(4)
    }

    // ================================================================ E: K3 for this wave's 32 clusters, two clusters per pass
    {
      // 8 KB of scratch rows per wave (the stage buffers X32 | XH | XL are idle here and contiguous): two rounds of 16 clusters.
      // The sums are written with literal register indices; the 8 passes of a round then run as a REAL loop over the LDS copy
      // (unrolled 16 times, the normalisation -- with its inlined pow() for the general exponents -- was 37,000 instructions:
      // instruction-cache misses and spilled store addresses made the epilogue the slowest phase).
      float* const scr = x32 + wave * 2048;
      // pass p of a round finishes clusters cbase + 2p (lanes 0..31) and cbase + 2p + 1 (lanes 32..63)
      auto finish_round = [&](int cbase) {
#pragma unroll 2
        for (int p = 0; p < 8; ++p) {
          const int k = 32 * wave + cbase + 2 * p + h;
          const float4 t = *reinterpret_cast<const float4*>(scr + (2 * p + h) * 128 + 4 * j);
          float v[4] = {t.x, t.y, t.z, t.w};
          if (a.norm_mode == 4) {   // training pass: raw residual sums
            if (k < a.K) {
              *reinterpret_cast<float4*>(out_img + (int64_t)k * 128 + 4 * j) = t;
              if (j == 0) s_rowsq[k] = 0.f;
            }
            continue;
          }
          float part = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            v[q] = power_norm<false>(v[q], a.power);
            const float tt = norm_accum(v[q], a.norm_mode, a.norm_p);
            part = a.norm_mode == 3 ? fmaxf(part, tt) : part + tt;
          }
          float nrm = a.norm_mode == 3 ? half_max_xor(part) : half_sum_xor(part);   // the gather kernel's butterfly, bit for bit
          if (a.norm_mode == 2) nrm = sqrtf(nrm);
          else if (a.norm_mode == 0) nrm = powf(nrm, 1.f / a.norm_p);
          const float den = nrm + a.eps;
          float sq = 0.f, o[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            o[q] = v[q] / den;      // (IEEE forms here, the short forms of desc_load.hpp in the gather kernel: the same bits)
            sq += o[q] * o[q];
          }
          if (k < a.K) *reinterpret_cast<float4*>(out_img + (int64_t)k * 128 + 4 * j) = make_float4(o[0], o[1], o[2], o[3]);
          sq = half_sum_xor(sq);
          if (j == 0 && k < a.K) s_rowsq[k] = sq;
        }
      };
#define FU_EW(c) *reinterpret_cast<float2*>(scr + ((c) & 15) * 128 + 2 * lane) = make_float2(acc[2 * (c)], acc[2 * (c) + 1]);
      FU_EW(0) FU_EW(1) FU_EW(2) FU_EW(3) FU_EW(4) FU_EW(5) FU_EW(6) FU_EW(7)
      FU_EW(8) FU_EW(9) FU_EW(10) FU_EW(11) FU_EW(12) FU_EW(13) FU_EW(14) FU_EW(15)
      __builtin_amdgcn_wave_barrier();
      finish_round(0);
      __builtin_amdgcn_wave_barrier();
      FU_EW(16) FU_EW(17) FU_EW(18) FU_EW(19) FU_EW(20) FU_EW(21) FU_EW(22) FU_EW(23)
      FU_EW(24) FU_EW(25) FU_EW(26) FU_EW(27) FU_EW(28) FU_EW(29) FU_EW(30) FU_EW(31)
      __builtin_amdgcn_wave_barrier();
      finish_round(16);
#undef FU_EW
      FU_CASES(FU_ZERO)
    }
    __syncthreads();
    if (a.inv_norm != nullptr && wave == 0) {   // 1 / ||row||_2 for the cosine step (zero norm -> 1), same order as vlad_aggregate_kernel
      float s2 = 0.f;
      for (int k = lane; k < a.K; k += 64) s2 += s_rowsq[k];
      s2 = wave_sum_xor(s2, 64);
      if (lane == 0) a.inv_norm[cur] = s2 > 0.f ? 1.f / sqrtf(s2) : 1.f;
    }
    FU_STAMP(5)
    cur = nxt;
    // (the next image's first barrier separates this image's s_rowsq / scratch reads from the next writes)
  }
  if constexpr (DIAG) {
    if (tid == 0 && a.stamps != nullptr) {
      for (int i = 0; i < 8; ++i) atomicAdd(a.stamps + i, dg_t[i]);
      for (int i = 0; i < 4; ++i) atomicAdd(a.stamps + 8 + i, dg_c[i]);
    }
  }
#undef FU_STAMP
}

__global__ void fused_queue_init_kernel(unsigned int* q, unsigned int v) { *q = v; }

template <int KIND>
static int launch_fused_kind(pvs_ctx* ctx, const FusedArgs& a, int grid) {
  auto kp = vlad_fused_kernel<KIND, false>;
  auto kd = vlad_fused_kernel<KIND, true>;
  auto k = a.stamps != nullptr ? kd : kp;
  PVS_TRY(ensure_lds(ctx, reinterpret_cast<const void*>(k), FU_LDS));
  hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(FU_THREADS), FU_LDS, ctx->stream, a);
  PVS_HIP(hipGetLastError());
  return PVS_OK;
}

bool vlad_fused_eligible(const pvs_codebook* cb, const void* d_desc, int kind, int ld, const float* d_out) {
  if (cb->d_c16n == nullptr || cb->K_pad != 256 || cb->D != 128) return false;
  const int esz = kind == PVS_DESC_U8_ROOTSIFT ? 1 : 4;
  if (ld % 4 != 0 || ld < 128) return false;
  if (reinterpret_cast<uintptr_t>(d_desc) % (4 * esz) != 0) return false;
  if (reinterpret_cast<uintptr_t>(d_out) % 16 != 0 || reinterpret_cast<uintptr_t>(cb->d_cent) % 16 != 0) return false;
  return true;
}

int launch_vlad_fused(pvs_ctx* ctx, const pvs_codebook* cb, const void* d_desc, int kind, int ld, const int64_t* d_offsets,
                      int64_t n_images, const pvs_norm_params& prm, float* d_out, int32_t* d_labels, float* d_inv_norm, bool raw) {
  if (n_images <= 0) return PVS_OK;
  if (n_images > 0x7fffffffLL) PVS_FAIL(PVS_ERR_UNSUPPORTED, "too many images in one call");
  FusedArgs a{};
  a.X = d_desc; a.ld = ld; a.offsets = d_offsets; a.n_images = n_images;
  a.c16n = static_cast<const _Float16*>(cb->d_c16n); a.cnk = static_cast<const _Float16*>(cb->d_cnk);
  a.cnorm = cb->d_cnorm; a.cent = cb->d_cent; a.cpad = cb->d_cpad;
  a.K = cb->K; a.c_shift = cb->c16_shift; a.cn_e1 = cb->cn_e1; a.cmax = cb->cmax;
  a.power = (float)prm.power_norm_weight; a.eps = (float)prm.epsilon;
  const double ord = prm.norm_order;
  if (std::isnan(ord) || ord <= 0.0) PVS_FAIL(PVS_ERR_UNSUPPORTED, "norm_order must be > 0 or +inf (got %g)", ord);
  a.norm_mode = raw ? 4 : (std::isinf(ord) ? 3 : (ord == 2.0 ? 2 : (ord == 1.0 ? 1 : 0)));
  a.norm_p = (float)ord;
  a.out = d_out; a.inv_norm = d_inv_norm; a.labels = d_labels;
  const int grid = (int)(n_images < ctx->num_cu ? n_images : ctx->num_cu);
  if (ctx->d_queue == nullptr) PVS_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->d_queue), 1024));
  a.queue = ctx->d_queue;
  a.stamps = ctx->d_fused_stamps;
  ScopedTimer tm(ctx, T_ASSIGN);
  hipLaunchKernelGGL(fused_queue_init_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->d_queue, (unsigned int)grid);
  switch (kind) {
    case PVS_DESC_F32: return launch_fused_kind<PVS_DESC_F32>(ctx, a, grid);
    case PVS_DESC_F32_ROOTSIFT: return launch_fused_kind<PVS_DESC_F32_ROOTSIFT>(ctx, a, grid);
    case PVS_DESC_U8_ROOTSIFT: return launch_fused_kind<PVS_DESC_U8_ROOTSIFT>(ctx, a, grid);
    default: PVS_FAIL(PVS_ERR_INVALID, "unknown descriptor kind %d", kind);
  }
}

}  // namespace pvs
