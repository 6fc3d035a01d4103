// fp16-operand "NT" GEMM, 256 x 256 tile, on the 8-phase schedule of the CDNA4 guide (cdna_hip_programming.md, "The 256^2
// 8-phase template"):  out[m][n] = (A_m . B_n) * (inva[m] * invb[n]), fp32 accumulate -- BASELINE configs[4], the similarity
// of fp16 encodings (reference semantics: pyvisim/_utils.py:312-330; fp16 storage is this engine's option).
//
//   * 8 waves = 2 (wr) x 4 (wc); a wave owns rows {64 wr + [0,64)} U {128 + 64 wr + [0,64)} and columns {32 wc + [0,32)} U
//     {128 + 32 wc + [0,32)} of the tile, i.e. one 64-row piece of EACH A half-tile and one 32-row piece of EACH B half-tile:
//     a half-tile (128 rows x 128 B = 16 KB of a k-tile of 64 halfs) is then read completely within ONE phase and can be
//     restaged early.  LDS = 2 k-tile buffers x 4 half-tiles (A0, A1, B0, B1) = 128 KB.
//   * a k-tile is four phases, one 64 x 32 quadrant of the wave's tile each (16 MFMAs 16x16x32 or 8 MFMAs 32x32x16):
//         phase 1: read B0 (4 x ds_read_b128), A0 (8)   quadrant (a0, b0)
//         phase 2: read B1 (4)                           quadrant (a0, b1)
//         phase 3: read A1 (8)                           quadrant (a1, b1)
//         phase 4: --                                    quadrant (a1, b0)      (B0 stays in registers)
//     every phase = [ds_reads | stage ONE half-tile (2 LDS-DMA per wave) | s_barrier | lgkmcnt(0) | MFMAs | s_barrier].  The
//     wave row wr = 1 runs one barrier behind wr = 0, so on every SIMD one wave feeds the matrix pipe while its partner reads
//     and stages.  An iteration is eight phases = two k-tiles (even / odd buffer).
//   * the LDS-DMA stream is B0, A0, B1, A1 of tile 0, 1, 2, ...; phase p of iteration i stages stream entry 8 i + 6 + p: three
//     half-tiles stay in flight, `s_waitcnt vmcnt(6)` sits at phases 4 and 8 only (never 0 in the loop):
//         phase 4's wait retires the odd buffer  -> read in phases 5-7;   phase 8's retires the even buffer -> phases 1-3.
//     WAR: B0 is restaged one phase after its reads (the 12-read phase retires its 4 B reads with lgkmcnt(8) BEFORE its first
//     barrier), the other half-tiles two phases after theirs.
//   * M16 selects v_mfma_f32_16x16x32_f16 (the shape on which the chip holds the higher clock under load, MI355X guide "DVFS
//     give-back" item 7) or v_mfma_f32_32x32x16_f16; same LDS image, same reads, same schedule.
//   * same LDS image as gemm_mfma.hpp: 128-B rows, 16-B chunk c of row r at chunk c ^ ((r >> 1) & 7) (swizzle applied on the
//     LDS-DMA's per-lane SOURCE address and on the read): 16-lane read groups are conflict-free for both MFMA shapes.
//   * k-tiles at or past the end of the row (the second tile of an odd count, the look-ahead of the last iterations, 16-B
//     chunks past L in a partial last tile) are staged from a zero buffer: the instruction count per phase never changes, so the
//     counted waits hold to the end.
#pragma once
#include <type_traits>

#include "gemm_mfma.hpp"

namespace pvs {

constexpr int G8_HALF_BYTES = 128 * GEMM_ROW_BYTES;   // 16 KB
constexpr int G8_LDS_BYTES = 8 * G8_HALF_BYTES;       // 128 KB
constexpr int G8_MIRROR_BYTES = 8 * 32 * 33 * 4;

template <int IMM>
__device__ __forceinline__ void g8_read(f32x4_t& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(IMM));
}

// SYMM: A == B, only tiles tn >= tm are listed, the mirrored tile is written through an LDS transpose.
// STAMP: diagnostic build (variant harness): in-kernel clock of every workgroup.
// TILED (variant harness: an experiment on the operand layout): A and B point at a TILED copy of the rows --
//   [row block of 128][k-tile][128 rows x 128 B, 16-B chunk g of row r stored at chunk g ^ ((r >> 1) & 7)], zero padded --
// so that a half-tile is 16 KB of CONTIGUOUS memory in exactly the LDS image's order and an LDS-DMA instruction copies 1 KB
// linearly (8 consecutive cache lines instead of one line from each of 8 rows 2 L bytes apart).
template <bool SYMM, bool M16, bool STAMP = false, bool TILED = false>
__global__ __launch_bounds__(512, 2) void gemm_f16_8ph_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const bool g1 = wr != 0;

  // ---- block -> tile (XCD-aware, bijective: blocks b and b + 8 share an XCD, each XCD gets a contiguous run of the list)
  int lin;
  {
    const int bid = blockIdx.x, nwg = gridDim.x;
    const int xcd = bid & 7, pos = bid >> 3;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + pos;
  }
  const GemmTile tile = g.tiles[g.tile_base + lin];
  const int tm = tile.tm, tn = tile.tn;
  const int64_t m0 = (int64_t)tm * 256, n0 = (int64_t)tn * 256;

  // ---- loader: this wave stages rows [16 w, 16 w + 16) of every half-tile (two 1-KB LDS-DMA instructions)
  const char* base_a = static_cast<const char*>(g.A) + m0 * g.lda * 2 - 1024;
  const char* base_b = static_cast<const char*>(g.B) + n0 * g.ldb * 2 - 1024;
  unsigned voff[4][2];   // [half-tile A0 A1 B0 B1][q]: per-lane byte offset from the base (second load carries offset:1024)
  {
    const int c = lane & 7;
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int r = 16 * wave + 8 * q + (lane >> 3);   // row inside the half-tile
        const int gc = c ^ ((r >> 1) & 7);
        const bool is_a = h < 2;
        const int64_t row0 = is_a ? m0 : n0, nrows = is_a ? g.M : g.N, ld = is_a ? g.lda : g.ldb;
        int64_t grow = row0 + 128 * (h & 1) + r;
        grow = grow < nrows ? grow : nrows - 1;          // rows past the edge are computed and discarded
        voff[h][q] = (unsigned)((grow - row0) * ld * 2 + 16 * gc + 1024 - q * 1024);
      }
  }
  const unsigned lds0 = lds_addr(smem);
  const int nk = (int)((g.L + 63) / 64);
  const int nk_full = TILED ? nk : (int)(g.L / 64);      // tiles [0, nk_full) are complete; tile nk_full (if < nk) is partial
  const unsigned wave_lds = (unsigned)wave * 2048;
  const char* tbase[4] = {nullptr, nullptr, nullptr, nullptr};   // TILED: the four half-tiles' row blocks at k-tile 0 (minus 1 KiB)
  if constexpr (TILED) {
    const int64_t nba = (g.M + 127) / 128, nbb = (g.N + 127) / 128;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      int64_t blk = (h < 2 ? m0 : n0) / 128 + (h & 1);
      const int64_t nb = h < 2 ? nba : nbb;
      blk = blk < nb ? blk : nb - 1;
      tbase[h] = static_cast<const char*>(h < 2 ? g.A : g.B) + blk * (int64_t)nk * G8_HALF_BYTES - 1024;
#pragma unroll
      for (int q = 0; q < 2; ++q) voff[h][q] = (unsigned)(wave * 2048 + lane * 16 + 1024);   // + q KiB comes from the immediate
    }
  }

  // stage half-tile H (0 A0, 1 A1, 2 B0, 3 B1) of k-tile t into buffer BUF
  auto stage = [&](auto H_, auto BUF_, int t) {
    constexpr int H = decltype(H_)::value, BUF = decltype(BUF_)::value;
    constexpr bool IS_A = H < 2;
    const unsigned dst = lds0 + (BUF * 4 + H) * G8_HALF_BYTES + wave_lds;
    if (t < nk_full) {
      const char* sb = TILED ? tbase[H] + (int64_t)t * G8_HALF_BYTES : (IS_A ? base_a : base_b) + (int64_t)t * 128;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %2, %1\n\t"
                   "global_load_lds_dwordx4 %3, %1 offset:1024"
                   ::"s"(dst), "s"(sb), "v"(voff[H][0]), "v"(voff[H][1])
                   : "memory");
    } else if constexpr (TILED) {
      // a tile past the end: zeros (same instruction count)
#pragma unroll
      for (int q = 0; q < 2; ++q)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g.zero16,
                                         (__attribute__((address_space(3))) void*)(smem + (BUF * 4 + H) * G8_HALF_BYTES + wave * 2048 + q * 1024),
                                         16, 0, 0);
    } else {
      // partial last tile or a tile past the end: chunks at or past L come from 16 B of zeros (same instruction count)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int r = 16 * wave + 8 * q + (lane >> 3);
        const int gc = (lane & 7) ^ ((r >> 1) & 7);
        const int64_t k = (int64_t)t * 64 + 8 * gc;
        const char* p = (k < g.L) ? (IS_A ? base_a : base_b) + (int64_t)t * 128 + (size_t)voff[H][q] + q * 1024
                                  : reinterpret_cast<const char*>(g.zero16);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                         (__attribute__((address_space(3))) void*)(smem + (BUF * 4 + H) * G8_HALF_BYTES + wave * 2048 + q * 1024),
                                         16, 0, 0);
      }
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;

  // ---- accumulators
  constexpr int NACC = M16 ? 32 : 8;
  using acc_t = typename std::conditional<M16, f32x4_t, f32x16_t>::type;
  acc_t acc[NACC];
#pragma unroll
  for (int x = 0; x < NACC; ++x)
#pragma unroll
    for (int r = 0; r < (M16 ? 4 : 16); ++r) acc[x][r] = 0.f;

  // ---- per-lane fragment read bases (byte offsets from the start of a half-tile), one per k-step (the swizzle XOR is not an add)
  //   M16: lane (i = l & 15, q = l >> 4) reads row i of a 16-row block, chunk 4 s + q, s = 0, 1; blocks are 2048 B apart
  //   M32: lane (i = l & 31, h = l >> 5) reads row i of a 32-row block, chunk 2 t + h, t = 0..3; blocks are 4096 B apart
  constexpr int KS = M16 ? 2 : 4;
  unsigned ra[KS], rb[KS];   // A piece at rows 64 wr, B piece at rows 32 wc of their half-tiles, buffer 0
  {
    const int i = M16 ? (lane & 15) : (lane & 31), hq = M16 ? (lane >> 4) : (lane >> 5);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int cc = M16 ? 4 * s + hq : 2 * s + hq;
      ra[s] = lds0 + gemm_frag_off(64 * wr + i, cc);
      rb[s] = lds0 + 2 * G8_HALF_BYTES + gemm_frag_off(32 * wc + i, cc);
    }
  }
  f32x4_t fa[8], fb0[4], fb1[4];   // A piece (64 rows x 64 k), B0 piece, B1 piece (32 rows x 64 k each)

  // reads of one A piece (half-tile AH = 0 / 1) or B piece (half-tile 2 + BH) of buffer BUF
  auto read_a = [&](auto BUF_, auto AH_) {
    constexpr int OFF = decltype(BUF_)::value * 4 * G8_HALF_BYTES + decltype(AH_)::value * G8_HALF_BYTES;
    if constexpr (OFF + 3 * 2048 < 65536) {
      if constexpr (M16) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          g8_read<OFF>(fa[4 * s + 0], ra[s]);
          g8_read<OFF + 2048>(fa[4 * s + 1], ra[s]);
          g8_read<OFF + 4096>(fa[4 * s + 2], ra[s]);
          g8_read<OFF + 6144>(fa[4 * s + 3], ra[s]);
        }
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          g8_read<OFF>(fa[2 * t + 0], ra[t]);
          g8_read<OFF + 4096>(fa[2 * t + 1], ra[t]);
        }
      }
    } else {   // buffer 1: the immediate field is 16 bits, move the buffer offset into the address
      constexpr int HI = 4 * G8_HALF_BYTES, LO = OFF - HI;
      if constexpr (M16) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const unsigned b = ra[s] + HI;
          g8_read<LO>(fa[4 * s + 0], b);
          g8_read<LO + 2048>(fa[4 * s + 1], b);
          g8_read<LO + 4096>(fa[4 * s + 2], b);
          g8_read<LO + 6144>(fa[4 * s + 3], b);
        }
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const unsigned b = ra[t] + HI;
          g8_read<LO>(fa[2 * t + 0], b);
          g8_read<LO + 4096>(fa[2 * t + 1], b);
        }
      }
    }
  };
  auto read_b = [&](auto BUF_, auto BH_, f32x4_t (&fb)[4]) {
    constexpr int HI = decltype(BUF_)::value * 4 * G8_HALF_BYTES, LO = decltype(BH_)::value * G8_HALF_BYTES;
    if constexpr (M16) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const unsigned b = rb[s] + HI;
        g8_read<LO>(fb[2 * s + 0], b);
        g8_read<LO + 2048>(fb[2 * s + 1], b);
      }
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) g8_read<LO>(fb[t], rb[t] + HI);
    }
  };
  // the MFMAs of quadrant (QA, QB)
  auto mfmas = [&](auto QA_, auto QB_, const f32x4_t (&fb)[4]) {
    constexpr int QA = decltype(QA_)::value, QB = decltype(QB_)::value;
    __builtin_amdgcn_s_setprio(1);
    if constexpr (M16) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            acc_t& c = acc[(QA * 4 + m) * 4 + QB * 2 + n];
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, fa[4 * s + m]),
                                                       __builtin_bit_cast(f16x8_t, fb[2 * s + n]), c, 0, 0, 0);
          }
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          acc_t& c = acc[(QA * 2 + m) * 2 + QB];
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, fa[2 * t + m]), __builtin_bit_cast(f16x8_t, fb[t]), c,
                                                     0, 0, 0);
        }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  auto sync_mfma = [&]() {   // first barrier of a phase, then the reads of this phase must be back
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  auto end_phase = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };

  unsigned long long st_t0 = 0, st_r0 = 0;
  if constexpr (STAMP) {
    st_t0 = __builtin_amdgcn_s_memtime();
    st_r0 = __builtin_amdgcn_s_memrealtime();
  }

  // ---- prologue: tile 0 complete, three half-tiles of tile 1 in flight
  stage(I2{}, I0{}, 0); stage(I0{}, I0{}, 0); stage(I3{}, I0{}, 0); stage(I1{}, I0{}, 0);
  stage(I2{}, I1{}, 1); stage(I0{}, I1{}, 1); stage(I3{}, I1{}, 1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (g1) __builtin_amdgcn_s_barrier();   // wave row 1 runs one barrier behind

  const int niter = (nk + 1) / 2;
  for (int it = 0; it < niter; ++it) {
    const int t1 = 2 * it + 1, t2 = 2 * it + 2, t3 = 2 * it + 3;
    // ------------------------------------------------ k-tile 2 it (even buffer)
    // phase 1: B0, A0 | stage tile t1 . A1 (odd)
    read_b(I0{}, I0{}, fb0);
    __builtin_amdgcn_sched_barrier(0);
    read_a(I0{}, I0{});
    stage(I1{}, I1{}, t1);
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");   // the four B0 reads are back: B0 may be restaged next phase
    sync_mfma();
    mfmas(I0{}, I0{}, fb0);
    end_phase();
    // phase 2: B1 | stage tile t2 . B0 (even)
    read_b(I0{}, I1{}, fb1);
    stage(I2{}, I0{}, t2);
    sync_mfma();
    mfmas(I0{}, I1{}, fb1);
    end_phase();
    // phase 3: A1 | stage tile t2 . A0 (even)
    read_a(I0{}, I1{});
    stage(I0{}, I0{}, t2);
    sync_mfma();
    mfmas(I1{}, I1{}, fb1);
    end_phase();
    // phase 4: -- | stage tile t2 . B1 (even); the odd buffer (tile t1) must have landed
    stage(I3{}, I0{}, t2);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    sync_mfma();
    mfmas(I1{}, I0{}, fb0);
    end_phase();
    // ------------------------------------------------ k-tile 2 it + 1 (odd buffer)
    // phase 5: B0, A0 | stage tile t2 . A1 (even)
    read_b(I1{}, I0{}, fb0);
    __builtin_amdgcn_sched_barrier(0);
    read_a(I1{}, I0{});
    stage(I1{}, I0{}, t2);
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    sync_mfma();
    mfmas(I0{}, I0{}, fb0);
    end_phase();
    // phase 6: B1 | stage tile t3 . B0 (odd)
    read_b(I1{}, I1{}, fb1);
    stage(I2{}, I1{}, t3);
    sync_mfma();
    mfmas(I0{}, I1{}, fb1);
    end_phase();
    // phase 7: A1 | stage tile t3 . A0 (odd)
    read_a(I1{}, I1{});
    stage(I0{}, I1{}, t3);
    sync_mfma();
    mfmas(I1{}, I1{}, fb1);
    end_phase();
    // phase 8: -- | stage tile t3 . B1 (odd); the even buffer (tile t2) must have landed
    stage(I3{}, I1{}, t3);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    sync_mfma();
    mfmas(I1{}, I0{}, fb0);
    end_phase();
  }
  if (!g1) __builtin_amdgcn_s_barrier();                  // wave row 0 catches up with the row that ran a barrier behind
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the look-ahead of the last iteration (zero tiles) has landed
  __builtin_amdgcn_sched_barrier(0);

  if constexpr (STAMP) {
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && g.stamps) {
      unsigned long long* o = g.stamps + 8 * blockIdx.x;
      o[0] = st_t0; o[1] = t1; o[2] = st_r0; o[3] = r1;
      o[4] = o[5] = o[6] = o[7] = 0;
    }
  }

  // ---- epilogue: scale, store (and mirror).  Accumulator x covers rows mrow(x) + .., columns ncol(x) + ..
  const bool mirror = SYMM && tm != tn;
  if (mirror) __syncthreads();   // every wave is done with the operand buffers (the patches below reuse them)
  if constexpr (M16) {
    const int i = lane & 15, q = lane >> 4;
    float* patch = reinterpret_cast<float*>(smem) + wave * (16 * 17);
#pragma unroll
    for (int x = 0; x < 32; ++x) {
      const int am = x >> 2, bn = x & 3;   // am = QA * 4 + m, bn = QB * 2 + n
      const int64_t mb = m0 + (am >> 2) * 128 + 64 * wr + 16 * (am & 3), nb = n0 + (bn >> 1) * 128 + 32 * wc + 16 * (bn & 1);
      const int64_t n = nb + i;
      const float sb = (n < g.N && g.invb) ? g.invb[n] : 1.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t m = mb + 4 * q + r;
        const float sa = (m < g.M && g.inva) ? g.inva[m] : 1.f;
        acc[x][r] = acc[x][r] * (sa * sb);   // sa*sb commutes: out[m][n] == out[n][m] bitwise
        if (m < g.M && n < g.N) {
          if (g.accumulate) acc[x][r] += g.out[m * g.ldo + n];
          g.out[m * g.ldo + n] = acc[x][r];
        }
      }
      if (mirror) {
#pragma unroll
        for (int r = 0; r < 4; ++r) patch[i * 17 + 4 * q + r] = acc[x][r];   // [n][m]
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int nn = 4 * rr + q;
          const float v = patch[nn * 17 + i];   // lanes i -> consecutive m (LDS operations of one wave execute in order)
          if (nb + nn < g.N && mb + i < g.M) g.out[(nb + nn) * g.ldo + mb + i] = v;
        }
      }
    }
  } else {
    const int i = lane & 31, h = lane >> 5;
    float* patch = reinterpret_cast<float*>(smem) + wave * (32 * 33);
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      const int am = x >> 1, bn = x & 1;   // am = QA * 2 + m, bn = QB
      const int64_t mb = m0 + (am >> 1) * 128 + 64 * wr + 32 * (am & 1), nb = n0 + bn * 128 + 32 * wc;
      const int64_t n = nb + i;
      const float sb = (n < g.N && g.invb) ? g.invb[n] : 1.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = mb + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float sa = (m < g.M && g.inva) ? g.inva[m] : 1.f;
        acc[x][r] = acc[x][r] * (sa * sb);
        if (m < g.M && n < g.N) {
          if (g.accumulate) acc[x][r] += g.out[m * g.ldo + n];
          g.out[m * g.ldo + n] = acc[x][r];
        }
      }
      if (mirror) {
#pragma unroll
        for (int r = 0; r < 16; ++r) patch[i * 33 + (r & 3) + 8 * (r >> 2) + 4 * h] = acc[x][r];
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          const int nn = 2 * rr + h;
          const float v = patch[nn * 33 + i];
          if (nb + nn < g.N && mb + i < g.M) g.out[(nb + nn) * g.ldo + mb + i] = v;
        }
      }
    }
  }
}

}  // namespace pvs
