// MFMA "NT" GEMM for the cosine step:  out[m][n] = (A_m . B_n) * (inva[m] * invb[n])  (fp32 out),
// A [M][lda], B [N][ldb] both K-major (row = one encoding).  Shared by cosine.hip and the variant harness
// (bench/gemm_variants.hip).  Two element types, one kernel:
//     F16 = false  exact fp32: v_mfma_f32_32x32x2_f32, BK = 32 floats.  Shipped 128x128 tile, 4 waves, 2 workgroups
//                  per CU (two waves per SIMD from DIFFERENT workgroups: one computes while the other syncs).
//                  Measured 8189 x 8189 x 32768: 146.7 TFLOP/s = 93 % of the f32 MFMA peak.
//     F16 = true   fp16 operands, fp32 accumulate: v_mfma_f32_32x32x16_f16, BK = 64 halfs.  Same bytes per k-tile
//                  row (128 B), so the LDS image, the swizzle and the loader are shared.
//
//   * wave tile (32*MI) x (32*NI), block tile BM x BN, k-tile = 128 B per row.
//   * operands stream HBM -> LDS with 16-B-per-lane LDS-DMA; one wave-instruction fills 8 rows x 128 B.  The LDS image
//     is XOR-swizzled through the per-lane SOURCE offset (chunk ^= (row >> 1) & 7) with the same XOR on the
//     ds_read_b128: conflict-free 16-lane read groups.
//   * What the load phase must NOT do (each was measured on this kernel):
//       - use the vector ALU.  The f32 MFMA executes at the vector-FMA rate and a wave that needs VALU slots while
//         its SIMD partner streams MFMAs starves: with 8 x `v_lshl_add_u64` per k-tile the load phase took ~4040
//         cycles (the whole of the partner's MFMA phase), with none 842.  Addresses are therefore
//         `scalar 64-bit base (SALU add per k-tile) + constant 32-bit lane offset` in hand-written
//         `global_load_lds_dwordx4 voff, s[base:base+1] offset:imm` groups (hipcc always materialises a 64-bit VGPR
//         address for the builtin);
//       - let hipcc see the fragment ds_reads.  It cannot prove that the in-flight LDS-DMA of the NEXT tile does not
//         alias them and puts `s_waitcnt vmcnt(0)` in front of the first read of every k-tile (load and compute
//         serialise: 125 TFLOP/s).  Reads are inline asm with hand-counted lgkmcnt + sched_barrier.
//   * per k-tile: counted vmcnt (tile t landed; t+1 .. t+STAGES-2 in flight), raw s_barrier, issue tile
//     t+STAGES-1, compute tile t.
//   * two-level accumulation: MFMA chains are cut every KBLOCK k-values and summed into a second register tile
//     (a single 32768-long fp32 fma chain drifts ~4e-6 relative; blocked it is ~1.6e-7, BLAS level).
//   * tiles come from a host-built list (any order / subset): XCD-aware 8x8 super-tile order for L2 panel sharing;
//     SYMM (A == B): only tiles tn >= tm are listed, the mirrored tile is written through an LDS transpose
//     (coalesced) and is bit-identical to the direct one.
//   * MODE_PARTIAL / MODE_REDUCE: split-K for the tiles of a partly filled round (and for small problems): S blocks
//     per tile each run a slice of whole 1024-k chains and store every chain's accumulator; one block per tile adds
//     the chains in chain order -- the very additions the unsplit kernel makes -- and runs the normal epilogue.
//     A score therefore does not depend on which tile, launch or split produced it (bitwise).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace pvs {

typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

struct GemmTile {
  int tm, tn;
};

struct GemmArgs {
  const void* A;     // fp32 or fp16 rows
  const void* B;
  int64_t M, N, L;   // L in elements
  int64_t lda, ldb;  // row strides in elements (>= L)
  const float* inva;
  const float* invb;
  float* out;
  int64_t ldo;
  float* out_t;      // DUAL: transposed panel out_t[n][m] (row stride ldt); SYMM mirrors into `out` itself
  int64_t ldt;
  const GemmTile* tiles;  // tile list (device)
  int tile_base;          // first list entry handled by this launch
  int splitk;             // MODE_PARTIAL: blocks (k-slices) per tile
  int nparts;             // partial images per tile: TWO ? number of 1024-k chains : splitk
  float* partial;         // [tiles in this launch][nparts][BM*BN] raw accumulators
  const float* zero16;    // 16 B of zeros in device memory (source for k-chunks past L)
  unsigned long long* stamps;  // diagnostic builds only (STAMP)
  int accumulate;         // != 0: out += result (a long row dimension processed in segments, one launch each)
  int dbg;                // STAMP builds only (ablation): 1 no LDS-DMA in the k-loop, 2 no fragment reads after the first tile, 4 no MFMA, 8 all tiles load panel 0
};

#ifndef PVS_GEMM_DBG
#define PVS_GEMM_DBG 0   // 1 (variant harness only): GemmArgs::dbg ablation bits are honoured by unstamped builds too
#endif
constexpr bool F16_PRIO = true;      // s_setprio 1 around the fp16 MFMA block (measured: profiles/r02_fp16sim_*)
constexpr int GEMM_ROW_BYTES = 128;  // bytes of one operand row in a k-tile
constexpr int GEMM_KBLOCK = 1024;    // k-values per MFMA accumulation chain (two-level summation)
enum { GEMM_MODE_FULL = 0, GEMM_MODE_PARTIAL = 1, GEMM_MODE_REDUCE = 2 };

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

// LW = number of waves that issue the LDS-DMA loads (the first LW waves of the workgroup); the others only compute
template <int BM, int BN, int WM, int WN, int STAGES, bool F16 = false, int LW = WM * WN>
struct GemmCfg {
  static constexpr int ESZ = F16 ? 2 : 4;
  static constexpr int BK = GEMM_ROW_BYTES / ESZ;  // k-values per k-tile: 32 floats or 64 halfs
  static constexpr int THREADS = 64 * WM * WN;
  static constexpr int WAVES = WM * WN;
  static constexpr int MI = BM / WM / 32, NI = BN / WN / 32;
  static constexpr int A_BYTES = BM * GEMM_ROW_BYTES, B_BYTES = BN * GEMM_ROW_BYTES;
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int A_INSTR = BM / 8, B_INSTR = BN / 8;            // wave-instructions per tile
  static constexpr int LOADER_WAVES = LW;
  static constexpr int LOADS_PER_WAVE = (A_INSTR + B_INSTR) / LW;  // per k-tile, per loader wave
  static constexpr int MIRROR_BYTES = WAVES * 32 * 33 * 4;
  static constexpr int LDS_BYTES = STAGES * STAGE_BYTES > MIRROR_BYTES ? STAGES * STAGE_BYTES : MIRROR_BYTES;
  static_assert((A_INSTR + B_INSTR) % LW == 0, "tile loads must divide evenly over the loader waves");
  static_assert(A_INSTR % LOADS_PER_WAVE == 0, "a wave's loads must not straddle the A / B boundary");
  static_assert(BM % (32 * WM) == 0 && BN % (32 * WN) == 0, "wave tile must be a multiple of 32x32");
};

__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ unsigned gemm_frag_off(int row, int cc) { return (row * 8 + (cc ^ ((row >> 1) & 7))) * 16; }
__device__ __forceinline__ void ds_read_frag(f32x4_t& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr));
}
template <int N>
__device__ __forceinline__ void wait_lgkm() {
  if constexpr (N == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
  else static_assert(N < 0, "add this lgkmcnt value");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else static_assert(N < 0, "add this vmcnt value");
}

// This wave's LOADS_PER_WAVE LDS-DMA instructions per k-tile.  Wave w's q-th load fills LDS bytes
// [(w*LPW + q) * 1024, +1024) of the stage: A rows first, then B rows (A_BYTES = A_INSTR * 1024).
// Loads go in groups of <= 4 sharing one M0 (LDS base) and one scalar base; the q-th of a group uses the immediate
// offset q KiB, which the hardware adds to BOTH the LDS and the global address, so the lane offset is pre-biased.
template <class Cfg>
struct GemmLoader {
  static constexpr int LPW = Cfg::LOADS_PER_WAVE;
  const char* base;    // this wave's operand (A or B) at the tile's first row, minus 3 KiB (keeps voff >= 0)
  unsigned voff[LPW];  // per-lane byte offset from base (row * ld * 4 + swizzled chunk * 16 + 3 KiB - (q&3) KiB)
  unsigned gce[LPW];   // first element of the swizzled 16-B chunk inside the k-tile: k-tail check only
  int wave_base;       // byte offset of this wave's first load inside a stage buffer
  bool is_a;

  template <class Args>
  __device__ __forceinline__ void init(const Args& g, int64_t m0, int64_t n0, int wave, int lane) {
    wave_base = wave * LPW * 1024;
    is_a = wave * LPW < Cfg::A_INSTR;
    const int64_t row0 = is_a ? m0 : n0, nrows = is_a ? g.M : g.N, ld = is_a ? g.lda : g.ldb;
    base = static_cast<const char*>(is_a ? g.A : g.B) + row0 * ld * Cfg::ESZ - 3072;
#pragma unroll
    for (int q = 0; q < LPW; ++q) {
      const int inst = wave * LPW + q;
      const int r = 8 * (is_a ? inst : inst - Cfg::A_INSTR) + (lane >> 3);
      const int gc = (lane & 7) ^ ((r >> 1) & 7);
      int64_t grow = row0 + r;
      grow = grow < nrows ? grow : nrows - 1;  // rows past the edge are computed and discarded
      voff[q] = (unsigned)((grow - row0) * ld * Cfg::ESZ + 16 * gc + 3072 - (q & 3) * 1024);
      gce[q] = (16 / Cfg::ESZ) * gc;
    }
  }

  // full tile, loads [Q0, Q1): no VALU, no compiler-visible load (the caller counts vmcnt by hand).  Groups of <= 4
  // loads aligned to multiples of 4 share one M0 / one scalar base.
  template <int Q0, int Q1>
  __device__ __forceinline__ void issue(int64_t k0, char* stage) const {
    if constexpr (Q0 < Q1) {
      constexpr int GE = ((Q0 & ~3) + 4) < Q1 ? ((Q0 & ~3) + 4) : Q1;
      constexpr int I0 = (Q0 & 3) * 1024;  // immediate of the first load of this (possibly partial) group
      const char* sb = base + k0 * Cfg::ESZ;
      const unsigned m0v = lds_addr(stage + wave_base + (Q0 & ~3) * 1024);
      if constexpr (GE - Q0 == 4)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %2, %1\n\t"
                     "global_load_lds_dwordx4 %3, %1 offset:1024\n\t"
                     "global_load_lds_dwordx4 %4, %1 offset:2048\n\t"
                     "global_load_lds_dwordx4 %5, %1 offset:3072"
                     ::"s"(m0v), "s"(sb), "v"(voff[Q0]), "v"(voff[Q0 + 1]), "v"(voff[Q0 + 2]), "v"(voff[Q0 + 3])
                     : "memory");
      else if constexpr (GE - Q0 == 2 && I0 == 0)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %2, %1\n\t"
                     "global_load_lds_dwordx4 %3, %1 offset:1024"
                     ::"s"(m0v), "s"(sb), "v"(voff[Q0]), "v"(voff[Q0 + 1])
                     : "memory");
      else if constexpr (GE - Q0 == 2 && I0 == 2048)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %2, %1 offset:2048\n\t"
                     "global_load_lds_dwordx4 %3, %1 offset:3072"
                     ::"s"(m0v), "s"(sb), "v"(voff[Q0]), "v"(voff[Q0 + 1])
                     : "memory");
      else
        static_assert(GE - Q0 == 4, "unsupported LDS-DMA group shape");
      issue<GE, Q1>(k0, stage);
    }
  }

  // last k-tile when L % BK != 0: 16-B chunks at or past L come from a zero buffer (builtin path, rare)
  template <class Args>
  __device__ __forceinline__ void issue_checked(const Args& g, int64_t k0, char* stage) const {
#pragma unroll
    for (int q = 0; q < LPW; ++q) {
      const char* p = (k0 + gce[q] < g.L) ? base + k0 * Cfg::ESZ + (size_t)voff[q] + (q & 3) * 1024
                                          : reinterpret_cast<const char*>(g.zero16);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                       (__attribute__((address_space(3))) void*)(stage + wave_base + q * 1024), 16, 0, 0);
    }
  }
};

// ---- ping-pong kernel: operand ring.  A k-tile is four UNITS of 128 rows x 128 B (A top, A bottom, B left, B right; same
// swizzled image as above, row-local); the whole LDS is a ring of 10 unit slots (160 KB).  Unit j = 4 * tile + type sits in
// slot j % 10.  Every wave loads rows [16 w, 16 w + 16) of every unit (two LDS-DMA instructions, 1 KB each) and reads
// fragments from exactly one A unit (its wave row) and one B unit (its wave-column pair).
constexpr int RING_UNITS = 10, RING_UNIT_BYTES = 128 * GEMM_ROW_BYTES, RING_LDS_BYTES = RING_UNITS * RING_UNIT_BYTES;

struct GemmRingLoader {
  const char* base_a;   // operand at the tile's first row, minus 1 KiB (the second load of a pair carries offset:1024)
  const char* base_b;
  unsigned voff[4][2];  // [unit type][q]: per-lane byte offset from base (row * ld * 2 + swizzled chunk * 16 + 1 KiB - q KiB)
  unsigned gce[2];      // [q]: first element of this lane's 16-B chunk inside the k-tile (k-tail check)
  int wave_off;         // 2 KiB * wave: this wave's rows inside a unit

  __device__ __forceinline__ void init(const GemmArgs& g, int64_t m0, int64_t n0, int wave, int lane) {
    wave_off = wave * 2048;
    base_a = static_cast<const char*>(g.A) + m0 * g.lda * 2 - 1024;
    base_b = static_cast<const char*>(g.B) + n0 * g.ldb * 2 - 1024;
#pragma unroll
    for (int ut = 0; ut < 4; ++ut)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int r = 16 * wave + 8 * q + (lane >> 3);            // row inside the unit
        const int gc = (lane & 7) ^ ((r >> 1) & 7);
        const bool is_a = ut < 2;
        const int64_t row0 = is_a ? m0 : n0, nrows = is_a ? g.M : g.N, ld = is_a ? g.lda : g.ldb;
        int64_t grow = row0 + 128 * (ut & 1) + r;
        grow = grow < nrows ? grow : nrows - 1;                   // rows past the edge are computed and discarded
        voff[ut][q] = (unsigned)((grow - row0) * ld * 2 + 16 * gc + 1024 - q * 1024);
        gce[q] = 8 * gc;                                          // (the chunk column depends on q only)
      }
  }
  // the two units of one operand (A: types 0, 1; B: types 2, 3) of k-tile kt into slots s0, s1: 4 instructions
  template <bool IS_A>
  __device__ __forceinline__ void issue(int64_t k0, unsigned lds0, int s0, int s1) const {
    const char* sb = (IS_A ? base_a : base_b) + k0 * 2;
    const unsigned m0a = lds0 + s0 * RING_UNIT_BYTES + wave_off, m0b = lds0 + s1 * RING_UNIT_BYTES + wave_off;
    constexpr int U = IS_A ? 0 : 2;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %3, %2\n\t"
                 "global_load_lds_dwordx4 %4, %2 offset:1024\n\t"
                 "s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %5, %2\n\t"
                 "global_load_lds_dwordx4 %6, %2 offset:1024"
                 ::"s"(m0a), "s"(m0b), "s"(sb), "v"(voff[U][0]), "v"(voff[U][1]), "v"(voff[U + 1][0]), "v"(voff[U + 1][1])
                 : "memory");
  }
  // last k-tile when L % 64 != 0: chunks at or past L come from a zero buffer (builtin path, rare)
  template <bool IS_A>
  __device__ __forceinline__ void issue_checked(const GemmArgs& g, int64_t k0, char* smem, int s0, int s1) const {
    constexpr int U = IS_A ? 0 : 2;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const char* p = (k0 + gce[q] < g.L) ? (IS_A ? base_a : base_b) + k0 * 2 + (size_t)voff[U + u][q] + q * 1024
                                         : reinterpret_cast<const char*>(g.zero16);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                         (__attribute__((address_space(3))) void*)(smem + (u ? s1 : s0) * RING_UNIT_BYTES + wave_off + q * 1024),
                                         16, 0, 0);
      }
  }
};

template <int IMM>
__device__ __forceinline__ void ds_read_frag_imm(f32x4_t& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(IMM));
}

// OCC = minimum waves per SIMD the register allocator must leave room for (blocks per CU * threads / 256)
// TWO = two-level accumulation (needed for fp32-grade sums; off for the fp16 path, whose input rounding dominates)
// ILV = spread the next tile's LDS-DMA issue over the four k-steps of the current tile instead of one burst
// DUAL = besides out[m][n] also write the transposed panel out_t[n][m] (every tile), so that one GEMM serves the
//        row queries AND the column queries of a block pair (multi-GPU symmetric scheme).
// PP = ping-pong schedule (fp16, two wave groups = the two rows of waves, one wave of each group per SIMD): a k-tile is two
//      phases of [12 fragment reads | barrier | 16 MFMAs | barrier]; the second group runs one barrier behind the first, so
//      on every SIMD one wave feeds the matrix pipe while the other reads its next fragments and issues the LDS-DMA.
template <int BM, int BN, int WM, int WN, int STAGES, bool SYMM, int OCC, int MODE = GEMM_MODE_FULL, bool STAMP = false,
          bool F16 = false, bool TWO = true, bool ILV = false, bool DUAL = false, int LW = WM * WN, bool PP = false>
__global__ __launch_bounds__(64 * WM * WN, OCC) void gemm_mfma_kernel(GemmArgs g) {
  using Cfg = GemmCfg<BM, BN, WM, WN, STAGES, F16, LW>;
  constexpr int GEMM_BK = Cfg::BK;
  constexpr int MI = Cfg::MI, NI = Cfg::NI;
  static_assert(!(SYMM || DUAL) || BM == BN, "the mirrored store needs square tiles");
  static_assert(!(SYMM && DUAL), "SYMM already mirrors");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: LDS-DMA bases stay in SGPRs
  const int i = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;

  // ---- block -> list entry (XCD-aware, bijective): blocks b and b+8 share an XCD, give each XCD a contiguous run
  int lin, slice = 0;
  {
    int bid = blockIdx.x;
    if constexpr (MODE == GEMM_MODE_PARTIAL) {
      slice = bid % g.splitk;
      bid /= g.splitk;
    }
    if constexpr (MODE == GEMM_MODE_FULL) {
      const int nwg = gridDim.x;
      const int xcd = bid & 7, pos = bid >> 3;
      const int q8 = nwg >> 3, r8 = nwg & 7;
      lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + pos;
    } else {
      lin = bid;  // tail launches are small: plain order
    }
  }
  const GemmTile tile = g.tiles[g.tile_base + lin];
  const int tm = tile.tm, tn = tile.tn;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

  f32x16_t acc[MI][NI];
#pragma unroll
  for (int a = 0; a < MI; ++a)
#pragma unroll
    for (int b = 0; b < NI; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  if constexpr (MODE != GEMM_MODE_REDUCE) {
    f32x16_t tot[TWO ? MI : 1][TWO ? NI : 1];
    if constexpr (TWO) {
#pragma unroll
      for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) tot[a][b][r] = 0.f;
    }

    const int nk_all = (int)((g.L + GEMM_BK - 1) / GEMM_BK);
    constexpr int CHAIN_KT = GEMM_KBLOCK / GEMM_BK;  // k-tiles per accumulation chain
    int kt0 = 0, kt1 = nk_all;
    if constexpr (MODE == GEMM_MODE_PARTIAL) {
      // slices hold whole chains (TWO) so that the reduce can replay the unsplit kernel's additions exactly
      const int unit = TWO ? CHAIN_KT : 1;
      const int units = (nk_all + unit - 1) / unit;
      const int per = ((units + g.splitk - 1) / g.splitk) * unit;
      kt0 = min(nk_all, slice * per);
      kt1 = min(nk_all, kt0 + per);
    }
    const bool ktail = (g.L % GEMM_BK) != 0;  // only the last k-tile can reach past L
    GemmLoader<Cfg> ld;
    const bool loader = wave < Cfg::LOADER_WAVES;  // wave-uniform
    if (loader && !PP) ld.init(g, m0, n0, wave, lane);
    auto stage_tile = [&](int t) {
      if (!loader) return;
      char* st = smem + ((t - kt0) % STAGES) * Cfg::STAGE_BYTES;
      if (ktail && t == nk_all - 1) ld.issue_checked(g, (int64_t)t * GEMM_BK, st);
      else ld.template issue<0, Cfg::LOADS_PER_WAVE>((int64_t)t * GEMM_BK, st);
    };
    if constexpr (!PP) {
#pragma unroll
      for (int s = 0; s < STAGES - 1; ++s)
        if (kt0 + s < kt1) stage_tile(kt0 + s);
    }

    // per-lane fragment offsets inside a stage for the four k-steps (swizzled chunk cc = 2t + h)
    unsigned offa[MI][4], offb[NI][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int a = 0; a < MI; ++a) offa[a][t] = gemm_frag_off(wm * (32 * MI) + 32 * a + i, 2 * t + h);
#pragma unroll
      for (int b = 0; b < NI; ++b) offb[b][t] = Cfg::A_BYTES + gemm_frag_off(wn * (32 * NI) + 32 * b + i, 2 * t + h);
    }
    const unsigned lds0 = lds_addr(smem);
    unsigned long long st_t0 = 0, st_r0 = 0, seg[4] = {0, 0, 0, 0};
    if constexpr (STAMP) {  // in-kernel clock = d(memtime) / d(memrealtime) * 100 MHz
      st_t0 = __builtin_amdgcn_s_memtime();
      st_r0 = __builtin_amdgcn_s_memrealtime();
    }

    if constexpr (PP) {
      static_assert(F16 && !TWO && !ILV && STAGES == 2 && WM == 2 && WN == 4 && BM == 256 && BN == 256 && LW == 8,
                    "ping-pong: fp16, 256 x 256, eight waves");
      // Two wave groups (the two rows of waves; one wave of each group per SIMD).  A k-tile is two phases of
      // [12 fragment reads | barrier | 16 MFMAs | barrier]; group 1 runs one barrier behind group 0, so on every SIMD one wave
      // feeds the matrix pipe while the other reads its next fragments.  Barrier intervals of tile u (b = 4 u):
      //                group 0                                          group 1
      //   [b,   b+1)   LDS-DMA B(u+1); MFMA phase 0                     LDS-DMA B(u+1); reads phase 0
      //   [b+1, b+2)   reads phase 1                                    MFMA phase 0
      //   [b+2, b+3)   LDS-DMA A(u+2); MFMA phase 1, then vmcnt         LDS-DMA A(u+2); reads phase 1, then vmcnt
      //   [b+3, b+4)   reads phase 0 of tile u+1                        MFMA phase 1
      // The ring keeps 96 KB in flight: B(u+1) has one tile period to land, A(u+2) two (the L2 -> LDS path of a CU moves
      // ~70 GB/s: 64 KB per k-tile take 0.9 us of the ~1.1 us a k-tile's MFMAs take -- one burst per tile, waited for at the
      // end of the tile, was the bound of the two-stage kernel).  vmcnt(4) leaves exactly the four A(u+2) loads outstanding.
      // RAW: every issuer's vmcnt precedes barrier b+3, the first read of tile u+1 follows it.  WAR: slots of tile u-1 are
      // refilled after barrier b; its last reads (group 1, [b-2, b-1)) were retired (lgkmcnt(0)) before barrier b.
      const bool g1 = wm != 0;
      constexpr bool DBG = STAMP || PVS_GEMM_DBG;
      const int dbg = DBG ? g.dbg : 0;
      GemmRingLoader rl;
      rl.init(g, (dbg & 8) ? 0 : m0, (dbg & 8) ? 0 : n0, wave, lane);     // dbg 8: every tile streams the same two panels
      const int T = kt1 - kt0;
      auto load = [&](int u, bool is_a) {     // units of local tile u: A -> slots (4u, 4u+1) % 10, B -> (4u+2, 4u+3) % 10
        const int kt = kt0 + u;
        const int j = 4 * u + (is_a ? 0 : 2);
        const int s0 = j % RING_UNITS, s1 = (j + 1) % RING_UNITS;
        const int64_t k0 = (int64_t)kt * GEMM_BK;
        if (ktail && kt == nk_all - 1) {
          if (is_a) rl.issue_checked<true>(g, k0, smem, s0, s1);
          else rl.issue_checked<false>(g, k0, smem, s0, s1);
        } else {
          if (is_a) rl.issue<true>(k0, lds0, s0, s1);
          else rl.issue<false>(k0, lds0, s0, s1);
        }
      };
      // per-lane fragment offset inside a unit for the four k-steps; fragment m of a k-step is + 4096 m (immediate)
      unsigned fo[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) fo[t] = 128 * i + 16 * ((2 * t + h) ^ ((i >> 1) & 7));
      const unsigned b_half = (wn & 1) * 8192;     // this wave's 64 rows inside its B unit
      if (T > 0) { load(0, true); load(0, false); }
      if (T > 1) { load(1, true); wait_vm<4>(); } else wait_vm<0>();
      __builtin_amdgcn_s_barrier();
      if (g1) __builtin_amdgcn_s_barrier();
      f32x4_t av[2][MI], bv[2][NI];
      for (int u = 0; u < T; ++u) {
        const unsigned base_a = lds0 + ((4 * u + wm) % RING_UNITS) * RING_UNIT_BYTES;
        const unsigned base_b = lds0 + ((4 * u + 2 + (wn >> 1)) % RING_UNITS) * RING_UNIT_BYTES + b_half;
        const bool more1 = u + 1 < T && !(dbg & 1), more2 = u + 2 < T && !(dbg & 1);
        auto issue_b = [&]() { if (more1) load(u + 1, false); };
        auto issue_a = [&]() { if (more2) load(u + 2, true); };
        auto landed = [&]() {
          if (more2) wait_vm<4>();
          else wait_vm<0>();
        };
        auto reads = [&](int p) {
          if (DBG && (dbg & 2) && u != 0) return;
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const unsigned aa = base_a + fo[2 * p + s2], ab = base_b + fo[2 * p + s2];
            ds_read_frag_imm<0>(av[s2][0], aa);
            ds_read_frag_imm<4096>(av[s2][1], aa);
            ds_read_frag_imm<8192>(av[s2][2], aa);
            ds_read_frag_imm<12288>(av[s2][3], aa);
            ds_read_frag_imm<0>(bv[s2][0], ab);
            ds_read_frag_imm<4096>(bv[s2][1], ab);
          }
        };
        auto mfmas = [&]() {
          wait_lgkm<0>();
          __builtin_amdgcn_sched_barrier(0);
          if (DBG && (dbg & 4)) return;
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int a = 0; a < MI; ++a)
#pragma unroll
              for (int b = 0; b < NI; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, av[s2][a]),
                                                                   __builtin_bit_cast(f16x8_t, bv[s2][b]), acc[a][b], 0, 0, 0);
          __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);
        };
        unsigned long long p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0, p5 = 0, p6 = 0, p7 = 0, p8 = 0;
        if constexpr (STAMP) p0 = __builtin_amdgcn_s_memtime();
        if (g1) issue_b();
        reads(0);
        if constexpr (STAMP) p1 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_barrier();
        if constexpr (STAMP) p2 = __builtin_amdgcn_s_memtime();
        if (!g1) issue_b();
        mfmas();
        if constexpr (STAMP) p3 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_barrier();
        if constexpr (STAMP) p4 = __builtin_amdgcn_s_memtime();
        if (g1) issue_a();
        reads(1);
        if (g1) landed();
        if constexpr (STAMP) p5 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_barrier();
        if constexpr (STAMP) p6 = __builtin_amdgcn_s_memtime();
        if (!g1) issue_a();
        mfmas();
        if (!g1) landed();
        if constexpr (STAMP) p7 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_barrier();
        if constexpr (STAMP) {   // [0] read issue, [1] barrier after the reads, [2] lgkmcnt + 16 MFMAs (+ vmcnt), [3] barrier after the MFMAs
          p8 = __builtin_amdgcn_s_memtime();
          seg[0] += (p1 - p0) + (p5 - p4); seg[1] += (p2 - p1) + (p6 - p5); seg[2] += (p3 - p2) + (p7 - p6); seg[3] += (p4 - p3) + (p8 - p7);
        }
      }
      if (!g1) __builtin_amdgcn_s_barrier();
    } else
    for (int kt = kt0; kt < kt1; ++kt) {
      unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0;
      if constexpr (STAMP) s0 = __builtin_amdgcn_s_memtime();
      // tile kt must have landed; tiles kt+1 .. kt+STAGES-2 stay in flight across the barrier
      if (kt + STAGES - 2 < kt1) wait_vm<(STAGES - 2) * Cfg::LOADS_PER_WAVE>();
      else wait_vm<0>();
      if constexpr (STAMP) s1 = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_barrier();
      if constexpr (STAMP) s2 = __builtin_amdgcn_s_memtime();
      const int tnext = kt + STAGES - 1;
      const bool inl = ILV && tnext < kt1 && !(ktail && tnext == nk_all - 1);  // interleave only full tiles
      if (tnext < kt1 && !inl) stage_tile(tnext);
      if constexpr (STAMP) s3 = __builtin_amdgcn_s_memtime();
      const unsigned base = lds0 + ((kt - kt0) % STAGES) * Cfg::STAGE_BYTES;
      f32x4_t av[2][MI], bv[2][NI];
#pragma unroll
      for (int a = 0; a < MI; ++a) ds_read_frag(av[0][a], base + offa[a][0]);
#pragma unroll
      for (int b = 0; b < NI; ++b) ds_read_frag(bv[0][b], base + offb[b][0]);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int c = t & 1, n = c ^ 1;
        if (t < 3) {
#pragma unroll
          for (int a = 0; a < MI; ++a) ds_read_frag(av[n][a], base + offa[a][t + 1]);
#pragma unroll
          for (int b = 0; b < NI; ++b) ds_read_frag(bv[n][b], base + offb[b][t + 1]);
        }
        if constexpr (ILV) {
          if (inl && loader) {
            constexpr int QN = Cfg::LOADS_PER_WAVE;
            char* st = smem + ((tnext - kt0) % STAGES) * Cfg::STAGE_BYTES;
            const int64_t k0n = (int64_t)tnext * GEMM_BK;
            if (t == 0) ld.template issue<0, QN / 4>(k0n, st);
            if (t == 1) ld.template issue<QN / 4, QN / 2>(k0n, st);
            if (t == 2) ld.template issue<QN / 2, 3 * QN / 4>(k0n, st);
            if (t == 3) ld.template issue<3 * QN / 4, QN>(k0n, st);
          }
        }
        if (t < 3) wait_lgkm<MI + NI>();  // the step-t fragments are back; the step-t+1 reads stay in flight
        else wait_lgkm<0>();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (F16) {
          // one 16-B fragment = 8 halfs = this lane's k-slice of a 32x32x16 MFMA (k = 16*t + 8*h + j)
          if constexpr (F16_PRIO) __builtin_amdgcn_s_setprio(1);   // the MFMA block outranks the SIMD partner's read / DMA issue
#pragma unroll
          for (int a = 0; a < MI; ++a)
#pragma unroll
            for (int b = 0; b < NI; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, av[c][a]),
                                                                 __builtin_bit_cast(f16x8_t, bv[c][b]), acc[a][b], 0, 0, 0);
          if constexpr (F16_PRIO) __builtin_amdgcn_s_setprio(0);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int a = 0; a < MI; ++a)
#pragma unroll
              for (int b = 0; b < NI; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c][a][e], bv[c][b][e], acc[a][b], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (STAMP) {
        const unsigned long long s4 = __builtin_amdgcn_s_memtime();
        seg[0] += s1 - s0; seg[1] += s2 - s1; seg[2] += s3 - s2; seg[3] += s4 - s3;
      }
      if constexpr (TWO) {
        if ((kt & (CHAIN_KT - 1)) == CHAIN_KT - 1 || (MODE == GEMM_MODE_PARTIAL && kt == kt1 - 1)) {  // chain complete
          if constexpr (MODE == GEMM_MODE_PARTIAL) {
            float* dst = g.partial + ((int64_t)lin * g.nparts + kt / CHAIN_KT) * (BM * BN) + wave * (MI * NI * 1024);
#pragma unroll
            for (int a = 0; a < MI; ++a)
#pragma unroll
              for (int b = 0; b < NI; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                  dst[(a * NI + b) * 1024 + r * 64 + lane] = acc[a][b][r];
                  acc[a][b][r] = 0.f;
                }
          } else {
#pragma unroll
            for (int a = 0; a < MI; ++a)
#pragma unroll
              for (int b = 0; b < NI; ++b) {
                tot[a][b] += acc[a][b];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
              }
          }
        }
      }
    }
    if constexpr (STAMP) {
      const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
      if (threadIdx.x == 0 && g.stamps) {
        unsigned long long* o = g.stamps + 8 * blockIdx.x;
        o[0] = st_t0; o[1] = t1; o[2] = st_r0; o[3] = r1;
        o[4] = seg[0]; o[5] = seg[1]; o[6] = seg[2]; o[7] = seg[3];
      }
    }
    if constexpr (TWO) {
#pragma unroll
      for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b) acc[a][b] += tot[a][b];
    }
  }

  // raw accumulator image of a block: [wave][a*NI+b][reg][lane]  (256-B coalesced rows)
  constexpr int PART_ELEMS = BM * BN;
  if constexpr (MODE == GEMM_MODE_PARTIAL) {
    if constexpr (!TWO) {  // single-level accumulation: one image per slice
      float* dst = g.partial + ((int64_t)lin * g.nparts + slice) * PART_ELEMS + wave * (MI * NI * 1024);
#pragma unroll
      for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[(a * NI + b) * 1024 + r * 64 + lane] = acc[a][b][r];
    }
    return;
  }
  if constexpr (MODE == GEMM_MODE_REDUCE) {
    // images are added in index order: with TWO these are the chain sums, added exactly as the unsplit kernel does
    for (int s = 0; s < g.nparts; ++s) {
      const float* src = g.partial + ((int64_t)lin * g.nparts + s) * PART_ELEMS + wave * (MI * NI * 1024);
#pragma unroll
      for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][b][r] += src[(a * NI + b) * 1024 + r * 64 + lane];
    }
  }

  // ---- epilogue: scale, store.  C/D layout: col = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
  for (int a = 0; a < MI; ++a) {
#pragma unroll
    for (int b = 0; b < NI; ++b) {
      const int64_t n = n0 + wn * (32 * NI) + 32 * b + i;
      const float sb = (n < g.N && g.invb) ? g.invb[n] : 1.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + wm * (32 * MI) + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float sa = (m < g.M && g.inva) ? g.inva[m] : 1.f;
        acc[a][b][r] = acc[a][b][r] * (sa * sb);  // sa*sb commutes: out[m][n] == out[n][m] bitwise
        if (m < g.M && n < g.N) {
          if (g.accumulate) acc[a][b][r] += g.out[m * g.ldo + n];   // (the mirrored store below then carries the sum)
          g.out[m * g.ldo + n] = acc[a][b][r];
        }
      }
    }
  }
  if constexpr (SYMM || DUAL) {
    if (DUAL || tm != tn) {
      // mirrored tile: transpose each 32x32 sub-tile through this wave's private LDS patch, coalesced rows out
      float* const dst = DUAL ? g.out_t : g.out;
      const int64_t ldd = DUAL ? g.ldt : g.ldo;
      __syncthreads();  // every wave is done reading the operand stages
      float* patch = reinterpret_cast<float*>(smem) + wave * (32 * 33);
#pragma unroll
      for (int a = 0; a < MI; ++a) {
#pragma unroll
        for (int b = 0; b < NI; ++b) {
#pragma unroll
          for (int r = 0; r < 16; ++r) patch[i * 33 + (r & 3) + 8 * (r >> 2) + 4 * h] = acc[a][b][r];  // [n][m]
          const int64_t mb = m0 + wm * (32 * MI) + 32 * a, nb = n0 + wn * (32 * NI) + 32 * b;
#pragma unroll
          for (int rr = 0; rr < 16; ++rr) {
            const int nn = 2 * rr + h;           // row of the mirrored tile handled by this half-wave
            const float v = patch[nn * 33 + i];  // lanes i -> consecutive m (LDS ops of one wave execute in order)
            if (nb + nn < g.N && mb + i < g.M) dst[(nb + nn) * ldd + mb + i] = v;
          }
        }
      }
    }
  }
}

}  // namespace pvs
