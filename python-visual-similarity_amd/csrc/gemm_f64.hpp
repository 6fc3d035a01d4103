// float64 "NT" GEMM for the cosine step on the f64 matrix pipe:  out[m][n] = (A_m . B_n) * (inva[m] * invb[n]).
//
// Reference semantics: pyvisim/_utils.py:312-330 -> sklearn cosine_similarity keeps float64 unless BOTH operands are
// float32, so Fisher encodings (float64, fisher_vector.py:99-135) are scored -- and then ranked, eval.py:37-43 -- in float64.
//
//   * v_mfma_f64_16x16x4_f64 (78.6 TFLOP/s dense on MI355X = 64 cycles per instruction per SIMD): A[i = lane & 15][k = lane >> 4],
//     B[k = lane >> 4][j = lane & 15], one f64 per lane each; C/D 4 f64 per lane, col = lane & 15, row = (lane >> 4) + 4 reg.
//   * 128 x 128 block tile, 4 waves (2 x 2), wave tile 64 x 64 = 4 x 4 MFMA tiles (128 accumulator registers), two workgroups
//     per CU: the two waves of a SIMD belong to different workgroups, one computes while the other waits at its barrier.
//   * k-tile = 128 B per row = 16 doubles, streamed HBM -> LDS by the loader of the f32 kernel (16-B-per-lane LDS-DMA, image
//     XOR-swizzled through the source address, no VALU in the load phase), double buffered.  A fragment read is one
//     ds_read_b128 = the lane's doubles (2c, 2c + 1) of chunk c = 4 s + (lane >> 4), s = 0, 1: conflict-free with the same
//     swizzle (rows 0-3 | 12-15 | 4-11 of a 16-lane read group fall on 16 distinct 16-B slots).
//   * FIXED k order: per k-tile the MFMAs (s, e), s = 0, 1, e = 0, 1 sum the four k = 16 t + 8 s + 2 q + e, q = 0..3; one
//     accumulation chain over the whole row (fp64: the drift over 262,400 terms is ~1e-14 relative).  A score therefore
//     depends only on its two rows and on how its tile is split along k (the plan is a function of the shape alone).
//   * SYMM (A == B): only tiles tn >= tm are listed; the mirrored tile goes through an LDS transpose and is bit-identical.
//   * split-K tail as in the f32 kernel (MODE_PARTIAL / MODE_REDUCE): the tiles of a partly filled last round are cut into
//     k-slices whose raw accumulators are added in slice order.
#pragma once
#include "gemm_mfma.hpp"

namespace pvs {

typedef double f64x4_t __attribute__((ext_vector_type(4)));
typedef double f64x2_t __attribute__((ext_vector_type(2)));

struct GemmArgsF64 {
  const void* A;  // float64 rows
  const void* B;
  int64_t M, N, L;   // L in elements (even: rows are 16-B aligned)
  int64_t lda, ldb;  // row strides in elements
  const double* inva;
  const double* invb;
  double* out;
  int64_t ldo;
  const GemmTile* tiles;
  int tile_base;
  int splitk;       // MODE_PARTIAL: k-slices per tile
  int nparts;       // partial images per tile (= splitk)
  double* partial;  // [tiles in this launch][nparts][128*128] raw accumulators
  const float* zero16;
};

struct GemmCfgF64 {
  static constexpr int BM = 128, BN = 128, WM = 2, WN = 2, STAGES = 2;
  static constexpr int ESZ = 8;
  static constexpr int BK = GEMM_ROW_BYTES / ESZ;  // 16 doubles per k-tile
  static constexpr int THREADS = 64 * WM * WN, WAVES = WM * WN;
  static constexpr int MI = BM / WM / 16, NI = BN / WN / 16;  // 4 x 4 MFMA tiles of 16 x 16 per wave
  static constexpr int A_BYTES = BM * GEMM_ROW_BYTES, B_BYTES = BN * GEMM_ROW_BYTES;
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int A_INSTR = BM / 8, B_INSTR = BN / 8;
  static constexpr int LOADER_WAVES = WAVES;
  static constexpr int LOADS_PER_WAVE = (A_INSTR + B_INSTR) / WAVES;
  static constexpr int MIRROR_BYTES = WAVES * 16 * 17 * 8;
  static constexpr int LDS_BYTES = STAGES * STAGE_BYTES;
  static constexpr int PART_ELEMS = BM * BN;
};

template <bool SYMM, int MODE = GEMM_MODE_FULL>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(GemmArgsF64 g) {
  using Cfg = GemmCfgF64;
  constexpr int MI = Cfg::MI, NI = Cfg::NI, BK = Cfg::BK;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, q = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;

  int lin, slice = 0;
  {
    int bid = blockIdx.x;
    if constexpr (MODE == GEMM_MODE_PARTIAL) {
      slice = bid % g.splitk;
      bid /= g.splitk;
    }
    if constexpr (MODE == GEMM_MODE_FULL) {  // XCD-aware, bijective: blocks b and b + 8 share an XCD
      const int nwg = gridDim.x;
      const int xcd = bid & 7, pos = bid >> 3;
      const int q8 = nwg >> 3, r8 = nwg & 7;
      lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + pos;
    } else {
      lin = bid;
    }
  }
  const GemmTile tile = g.tiles[g.tile_base + lin];
  const int tm = tile.tm, tn = tile.tn;
  const int64_t m0 = (int64_t)tm * Cfg::BM, n0 = (int64_t)tn * Cfg::BN;

  f64x4_t acc[MI][NI];
#pragma unroll
  for (int a = 0; a < MI; ++a)
#pragma unroll
    for (int b = 0; b < NI; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.0;

  if constexpr (MODE != GEMM_MODE_REDUCE) {
    const int nk_all = (int)((g.L + BK - 1) / BK);
    int kt0 = 0, kt1 = nk_all;
    if constexpr (MODE == GEMM_MODE_PARTIAL) {
      const int per = (nk_all + g.splitk - 1) / g.splitk;
      kt0 = min(nk_all, slice * per);
      kt1 = min(nk_all, kt0 + per);
    }
    const bool ktail = (g.L % BK) != 0;
    GemmLoader<Cfg> ld;
    ld.init(g, m0, n0, wave, lane);
    auto stage_tile = [&](int t) {
      char* st = smem + ((t - kt0) & 1) * Cfg::STAGE_BYTES;
      if (ktail && t == nk_all - 1) ld.issue_checked(g, (int64_t)t * BK, st);
      else ld.template issue<0, Cfg::LOADS_PER_WAVE>((int64_t)t * BK, st);
    };
    if (kt0 < kt1) stage_tile(kt0);

    unsigned offa[MI][2], offb[NI][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int a = 0; a < MI; ++a) offa[a][s] = gemm_frag_off(wm * 64 + 16 * a + i, 4 * s + q);
#pragma unroll
      for (int b = 0; b < NI; ++b) offb[b][s] = Cfg::A_BYTES + gemm_frag_off(wn * 64 + 16 * b + i, 4 * s + q);
    }
    const unsigned lds0 = lds_addr(smem);

    for (int kt = kt0; kt < kt1; ++kt) {
      wait_vm<0>();  // tile kt has landed (this wave's part; the barrier covers the others')
      __builtin_amdgcn_s_barrier();
      if (kt + 1 < kt1) stage_tile(kt + 1);
      const unsigned base = lds0 + ((kt - kt0) & 1) * Cfg::STAGE_BYTES;
      f32x4_t av[2][MI], bv[2][NI];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int a = 0; a < MI; ++a) ds_read_frag(av[s][a], base + offa[a][s]);
#pragma unroll
        for (int b = 0; b < NI; ++b) ds_read_frag(bv[s][b], base + offb[b][s]);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (s == 0) wait_lgkm<8>();
        else wait_lgkm<0>();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int a = 0; a < MI; ++a)
#pragma unroll
            for (int b = 0; b < NI; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(__builtin_bit_cast(f64x2_t, av[s][a])[e],
                                                               __builtin_bit_cast(f64x2_t, bv[s][b])[e], acc[a][b], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  // raw accumulator image of a block: [wave][a*NI+b][reg][lane]
  if constexpr (MODE == GEMM_MODE_PARTIAL) {
    double* dst = g.partial + ((int64_t)lin * g.nparts + slice) * Cfg::PART_ELEMS + wave * (MI * NI * 256);
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
      for (int b = 0; b < NI; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(a * NI + b) * 256 + r * 64 + lane] = acc[a][b][r];
    return;
  }
  if constexpr (MODE == GEMM_MODE_REDUCE) {
    for (int s = 0; s < g.nparts; ++s) {  // slices in k order
      const double* src = g.partial + ((int64_t)lin * g.nparts + s) * Cfg::PART_ELEMS + wave * (MI * NI * 256);
#pragma unroll
      for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[a][b][r] += src[(a * NI + b) * 256 + r * 64 + lane];
    }
  }

  // ---- epilogue.  C/D layout: col = lane & 15, row = (lane >> 4) + 4 reg
#pragma unroll
  for (int a = 0; a < MI; ++a) {
    double sa[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t m = m0 + wm * 64 + 16 * a + q + 4 * r;
      sa[r] = (m < g.M && g.inva) ? g.inva[m] : 1.0;
    }
#pragma unroll
    for (int b = 0; b < NI; ++b) {
      const int64_t n = n0 + wn * 64 + 16 * b + i;
      const double sb = (n < g.N && g.invb) ? g.invb[n] : 1.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t m = m0 + wm * 64 + 16 * a + q + 4 * r;
        acc[a][b][r] = acc[a][b][r] * (sa[r] * sb);  // sa*sb commutes: out[m][n] == out[n][m] bitwise
        if (m < g.M && n < g.N) g.out[m * g.ldo + n] = acc[a][b][r];
      }
    }
  }
  if constexpr (SYMM) {
    if (tm != tn) {
      __syncthreads();  // every wave is done reading the operand stages
      double* patch = reinterpret_cast<double*>(smem) + wave * (16 * 17);
#pragma unroll
      for (int a = 0; a < MI; ++a) {
#pragma unroll
        for (int b = 0; b < NI; ++b) {
#pragma unroll
          for (int r = 0; r < 4; ++r) patch[i * 17 + q + 4 * r] = acc[a][b][r];  // [n][m]
          const int64_t mb = m0 + wm * 64 + 16 * a, nb = n0 + wn * 64 + 16 * b;
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int nn = 4 * rr + q;
            const double v = patch[nn * 17 + i];  // lanes i -> consecutive m (LDS operations of one wave execute in order)
            if (nb + nn < g.N && mb + i < g.M) g.out[(nb + nn) * g.ldo + mb + i] = v;
          }
        }
      }
    }
  }
}

}  // namespace pvs
