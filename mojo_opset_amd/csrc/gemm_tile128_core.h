// 128-row-tile MFMA GEMM core for UNDER-FILLED launches of the 256 x 256 kernel (round 5); instantiated by gemm_tile128.hip
// (bf16 / fp16, plain epilogue) and quant_gemm.hip (int8 / fp8 with [N,K] weights, dequantising epilogue).
//
// Role: the dense products behind `MojoGemm.forward` = `F.linear(input, weight, bias)` (core/operators/gemm.py:45-46) and the
// GEMM halves of the GEMM + collective operators (core/operators/compute_with_comm.py:12-24: `F.linear` for [N,K] weights,
// `input @ weight (+ bias)` for [K,N]) at MID-SIZE M — a chunked prefill of 256..2048 tokens, config 4 at M 1024, the row chunks
// of the GEMM + collective pipelines.  There a product has fewer 256 x 256 tiles than the chip has CUs, and gemm256_core.h either
// runs on part of the chip or cuts K into slices whose fp32 slabs cost more than the matrix work (M 1024 x 4096 x 4096: 4 slices,
// 27 of 48 us are slab traffic).  hipBLASLt switches to 128-row macro tiles for these shapes and was 15-35 % faster
// (profiles/r5_gemm_mid_m_before.txt).  This kernel is that tile shape (measured against both: profiles/r5_gemm_tile128_ab.txt,
// r5_gemm_tile128_ab_kn.txt — M 1024 x 4096 x 4096 [N,K]: 47.7 -> 32.2 us, hipBLASLt 36.3):
//
//   * 128 x 128 output tile, K-tiles of 128 bytes, FOUR waves (2 x 2), each 64 x 64 of C on v_mfma_f32_16x16x32 (64 accumulator
//     registers).  A (and [N,K] weights) K-major in LDS as row-blocks of 16 rows x 128 bytes, filled by LDS-DMA in pieces of 8
//     WHOLE rows (a lane's 16-byte chunk goes to slot chunk ^ (row / 2 % 8) of its row: the swizzle is applied on the SOURCE
//     side, and every ds_read_b128 of a fragment is conflict-free).  Whole 128-byte lines per request fill 15 % faster than
//     gemm256_core.h's 16-row x 64-byte pieces (0.37 vs 0.44 us per K-tile with nothing else in the loop).  [K,N] weights in
//     gemm256_core.h's n-major image, read back by ds_read_b64_tr_b16;
//   * a ring of 4 stages of 32 KiB (three K-tiles in flight), ONE barrier per K-tile, fragments double-buffered in registers:
//     wait for stage t + 1 and for the own reads of K-tile t, barrier, then EIGHT SLICES of 4 MFMAs on K-tile t, each carrying
//     one of the eight fragments of K-tile t + 1 and one LDS-DMA request of stage t + 4 (into slot t % 4: everybody has its
//     fragments of t in registers).  The issue order is the point: with one wave per SIMD an instruction in front of the MFMAs
//     costs its whole issue time with the matrix unit idle — reads and requests ahead of the MFMAs, the compiler's own order,
//     took 0.57 us per K-tile; reads over the first half of the MFMAs and requests over the second 0.41; in slices 0.38 (K 4096,
//     one tile per CU on an eighth of the chip; scripts/probes/tile128_anatomy.hip, profiles/r5_tile128_anatomy.txt).  What
//     bounds it now is the per-CU fill rate: 32 KiB per K-tile at ~90 GB/s per CU = 0.36 us, against 0.21-0.29 us of MFMA time;
//   * beyond one tile per CU the SAME wave code on a 128 x 256 tile: EIGHT waves (2 x 4, two per SIMD), ring of 3 stages of
//     48 KiB — half again the fill for twice the MFMAs.  (A 64 x 128 wave tile on four waves needs the accumulators in AGPRs,
//     and hipcc then shuttles fragments through them; two workgroups of the 128 x 128 shape per CU on a two-stage ring were
//     the first form of this range: 63 us at M 2048 x 4096 x 4096 against 57 now, 71 on 256 x 256 tiles, hipBLASLt 58);
//   * tiles in the 256 kernel's order (bijective XCD remap, panels of 1024 columns, m-tile by m-tile inside a panel); row maps
//     on both sides; bias with the golden's rounding (F.linear: in the accumulator, one rounding; x @ w + b: after the rounding).
//
// Same fp32 accumulation per output element over K in index order as the 256 kernel's unsplit form: the two give the same bits
// (tests/test_hip_gemm_tile128.py).
#pragma once
#include "gemm256_core.h"

#ifndef T128_SCHED             // issue order of a step: 0 = the compiler's (reads and requests ahead of the MFMAs), 1 = in slices (see step)
#define T128_SCHED 1
#endif
#ifndef T128_TR_BUILTIN        // 1 = the [K,N] transposed reads through the builtin: hipcc then waits vmcnt(0) in front of each (measurement)
#define T128_TR_BUILTIN 0
#endif
#ifndef T128_ABLATE            // scripts/probes/tile128_anatomy.hip compiles this file with 1 (no C stores) / 2 / 3 / 4 (timing only, wrong results)
#define T128_ABLATE 0
#endif

namespace mojo {
namespace g128 {

using g256::frag16;
using g256::frag32;
using g256::glds16;
using g256::join;
using g256::lds_char;

constexpr int BM = 128;
constexpr int KT_BYTES = g256::KT_BYTES;            // 128 bytes of K per row and K-tile
constexpr int TILE_A_BYTES = BM * KT_BYTES;         // A's K-tile: 16 KiB = 8 row-blocks of 16 rows x 128 bytes
constexpr int PANEL_COLS = 1024;                    // output columns per panel

// s_waitcnt immediate on gfx9: vmcnt [3:0] + [15:14], expcnt [6:4] (7 = no wait), lgkmcnt [11:8] (15 = no wait)
constexpr int waitcnt_imm(int vm, int lgkm) { return (vm & 15) | ((vm >> 4) << 14) | 0x70 | ((lgkm & 15) << 8); }

// NWN = waves along N (each wave owns 64 x 64 of C): 2 -> 128 x 128 tile, four waves; 4 -> 128 x 256 tile, eight waves (two per
// SIMD: a 64 x 128 wave tile would need the accumulators in AGPRs, and hipcc then shuttles fragments through them); S = ring stages
template <typename P, typename Epi, int NWN, int S, bool W_NMAJOR /* true: W is [K,N] (n contiguous); false: [N,K] */>
__global__ __launch_bounds__(128 * NWN, 2) void gemm128_kernel(GemmArgs a, Epi epi) {
  typedef typename P::acc_t acc_t;
  constexpr int EB = P::EB;
  constexpr int BK = KT_BYTES / EB;
  constexpr int WN = 4;                                          // 16-column tiles per wave
  constexpr int BN = NWN * 64;
  constexpr int STAGE_BYTES = TILE_A_BYTES + BN * KT_BYTES;      // A | W
  constexpr int AB = 4 / NWN, WB = 2;                            // row-blocks of A / of W a wave stages
  constexpr int PIECES = 2 * AB + 2 * WB;                        // LDS-DMA requests per wave and stage
  constexpr int PANEL = PANEL_COLS / BN;                         // n-tiles per panel
  extern __shared__ __attribute__((aligned(1024))) char smem_generic[];
  lds_char* smem = (lds_char*)smem_generic;

  const int n_tiles = (a.N + BN - 1) / BN;
  const int m_tiles = gemm_m_tiles(a, BM);           // (ragged groups: the prefix arrays were built for 128-row tiles)
  const int tiles_mn = m_tiles * n_tiles;
  const int total = tiles_mn * a.splitk;             // split-K (dense products only): slice-major, so an XCD's run shares A and W
  const int bid = blockIdx.x;
  if (bid >= total) return;
  int tile;
  {  // bijective XCD remap: blocks b, b + 8, ... share an XCD; each XCD gets one contiguous run of tiles
    const int q = total >> 3, r = total & 7, x = bid & 7, i = bid >> 3;
    tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  const int kslice = tile / tiles_mn;                // 0 when splitk == 1
  tile -= kslice * tiles_mn;
  int mi, ni;
  {  // panel-major: panels of PANEL n-tiles, inside a panel m-tile by m-tile
    const int full_panels = n_tiles / PANEL, rem = n_tiles - full_panels * PANEL;
    const int in_full = full_panels * m_tiles * PANEL;
    if (tile < in_full) {
      const int p = tile / (m_tiles * PANEL), t = tile - p * (m_tiles * PANEL);
      mi = t / PANEL;
      ni = p * PANEL + (t - mi * PANEL);
    } else {
      const int t = tile - in_full;
      mi = t / rem;
      ni = full_panels * PANEL + (t - mi * rem);
    }
  }
  int grp_i, m0, m_end;                              // group, first row and end row (exclusive) of this m-tile
  gemm_locate_tile(a, mi, BM, grp_i, m0, m_end);
  const int n0 = ni * BN;
  const char* const W_g = static_cast<const char*>(a.W) + static_cast<int64_t>(grp_i) * a.w_group * EB;
  const int nkt_all = a.K / BK;
  const int kt0 = static_cast<int>(static_cast<int64_t>(nkt_all) * kslice / a.splitk);
  const int nkt = static_cast<int>(static_cast<int64_t>(nkt_all) * (kslice + 1) / a.splitk) - kt0;   // K-tiles of this slice (>= 1: the launcher's check)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / NWN, wn = wave % NWN;

  // ---- staging: wave w fills A's row-blocks AB * w ... and W's row-blocks 2w, 2w + 1 (16 rows x 128 bytes each), in pieces of
  // 8 WHOLE rows: lane l -> row l / 8 of the piece, LDS slot l % 8 of that row, which holds the row's 16-byte chunk (l % 8) ^ (row / 2 % 8)
  const char* srcA[2 * AB];
  const char* srcW[2 * WB];
  int w2_off[2] = {0, 0};                            // [K,N]: byte distance of a k-block's second request (n-blocks 4-7)
  {
    const int rr = lane >> 3, p = lane & 7;
#pragma unroll
    for (int h = 0; h < 2 * WB; ++h) {               // h = row-block (h / 2), piece (h % 2)
      const int r = (h & 1) * 8 + rr;
      const int chunk = p ^ ((r >> 1) & 7);
      if (h < 2 * AB) {
        int m = m0 + (AB * wave + (h >> 1)) * 16 + r;
        if (m >= m_end) m = m_end - 1;               // rows past the end: re-read a valid row, never stored
        srcA[h] = static_cast<const char*>(a.A) + static_cast<int64_t>(map_row(m, a.a_rc, a.a_ml, a.a_off, a.a_mul)) * a.lda * EB + chunk * 16;
      }
      if constexpr (!W_NMAJOR) {
        int n = n0 + (WB * wave + (h >> 1)) * 16 + r;
        if (n >= a.N) n = a.N - 1;
        srcW[h] = W_g + static_cast<int64_t>(n) * a.w_n * EB + chunk * 16;
      }
    }
  }
  // [K,N] weights: gemm256_core.h's image of a 64 k x 128 n half-tile — [k / 8][n / 16][8 k][16 n], 256-byte blocks, odd k-blocks
  // hold their rows 4-7 first — read back transposed by ds_read_b64_tr_b16.  The waves share the 8 k-blocks of each half-tile
  // (BN / 128 of them); a k-block takes two requests (n-blocks 0-3, 4-7): lane l -> n-block l / 16, stored row (l % 16) / 2,
  // columns (l % 2) * 8 .. + 8.  u = 0, 1 enumerates this wave's (half-tile, k-block) pairs.
  constexpr int NW = 2 * NWN, NH = BN / 128, KBW = 8 / NW;       // waves, half-tiles, k-blocks per wave and half-tile
  static_assert(NH * KBW == 2, "two (half-tile, k-block) pairs per wave");
  if constexpr (W_NMAJOR && EB == 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int hh = NH == 2 ? u : 0, kb = NH == 2 ? wave : wave * 2 + u;
      const int rr = ((lane & 15) >> 1) ^ ((kb & 1) ? 4 : 0);
      const int n = n0 + hh * 128 + (lane >> 4) * 16 + (lane & 1) * 8;
      const int n_a = n > a.N - 8 ? a.N - 8 : n;                   // partial n-tile: stay inside the row (those columns are never stored)
      const int n_b = n + 64 > a.N - 8 ? a.N - 8 : n + 64;
      srcW[u] = W_g + (static_cast<int64_t>(kb * 8 + rr) * a.w_k + n_a) * 2;
      w2_off[u] = (n_b - n_a) * 2;
    }
  }
  // 1-byte [K,N] weights: gemm256_core.h's image of a 128 k x 128 n half-tile — [k / 8][n / 16][8 k][16 n], 128-byte blocks, block
  // (kb, nb) in slot nb ^ (kb / 2 % 2) of its k-block — read back by ds_read_b64_tr_b8.  16 k-blocks per half-tile; a wave
  // fills four (half-tile, k-block) pairs u = 0 .. 3 with one request each (8 k-rows x 128 columns): lane l -> row l % 8,
  // slot l / 8.
  if constexpr (W_NMAJOR && EB == 1) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int hh = NH == 2 ? (u >> 1) : 0, kb = NH == 2 ? wave * 2 + (u & 1) : wave * 4 + u;
      const int nb = (lane >> 3) ^ ((kb >> 1) & 1);
      int n = n0 + hh * 128 + nb * 16;
      if (n > a.N - 16) n = a.N - 16;                              // partial n-tile: stay inside the row (those columns are never stored)
      srcW[u] = W_g + static_cast<int64_t>(kb * 8 + (lane & 7)) * a.w_k + n;
    }
  }
  const int64_t w_step = W_NMAJOR ? static_cast<int64_t>(BK) * a.w_k * EB : KT_BYTES;
  auto stage_piece = [&](int kt, int slot, int p) {  // LDS-DMA request p of the wave's PIECES for K-tile kt of the slice (p is a constant after unrolling)
    lds_char* dst = smem + slot * STAGE_BYTES;
    kt += kt0;
    if (p < 2 * AB) {
      glds16(srcA[p] + static_cast<int64_t>(kt) * KT_BYTES, dst + (AB * wave) * 2048 + p * 1024);
    } else if constexpr (!W_NMAJOR) {
      const int h = p - 2 * AB;
      glds16(srcW[h] + static_cast<int64_t>(kt) * KT_BYTES, dst + TILE_A_BYTES + (WB * wave) * 2048 + h * 1024);
    } else if constexpr (EB == 2) {
      const int u = (p - 2 * AB) >> 1, second = (p - 2 * AB) & 1;
      const int hh = NH == 2 ? u : 0, kb = NH == 2 ? wave : wave * 2 + u;
      glds16(srcW[u] + kt * w_step + (second ? w2_off[u] : 0), dst + TILE_A_BYTES + hh * 16384 + kb * 2048 + second * 1024);
    } else {
      const int u = p - 2 * AB;
      const int hh = NH == 2 ? (u >> 1) : 0, kb = NH == 2 ? wave * 2 + (u & 1) : wave * 4 + u;
      glds16(srcW[u] + kt * w_step, dst + TILE_A_BYTES + hh * 16384 + kb * 1024);
    }
  };
  auto stage = [&](int kt, int slot) {               // all PIECES of a stage (prologue)
    if (kt >= nkt) kt = nkt - 1;
#pragma unroll
    for (int p = 0; p < PIECES; ++p) stage_piece(kt, slot, p);
  };
  // ---- fragment reads: lane reads row l & 15, chunk ks * 4 + (l >> 4), stored in slot chunk ^ ((row / 2) % 8): conflict-free in
  // each of ds_read_b128's four 16-lane groups (rows of one parity share a 128-byte half of the banks and get 8 distinct slots)
  int frag_off[2];
  {
    const int r = lane & 15, q = lane >> 4, sw = (r >> 1) & 7;
    frag_off[0] = r * 128 + ((q ^ sw) * 16);
    frag_off[1] = r * 128 + (((4 + q) ^ sw) * 16);
  }
  typedef const __attribute__((address_space(3))) frag16* lds_frag_ptr;
  auto read_a = [&](frag32& f, int slot, int i) {    // 16 rows of A: two ds_read_b128
    const lds_char* base = smem + slot * STAGE_BYTES + (wm * 4 + i) * 2048;
    f = join(*reinterpret_cast<lds_frag_ptr>(base + frag_off[0]), *reinterpret_cast<lds_frag_ptr>(base + frag_off[1]));
  };
  auto read_w = [&](frag32& f, int slot, int j) {    // 16 columns of [N,K] weights
    const lds_char* base = smem + slot * STAGE_BYTES + TILE_A_BYTES + (wn * WN + j) * 2048;
    f = join(*reinterpret_cast<lds_frag_ptr>(base + frag_off[0]), *reinterpret_cast<lds_frag_ptr>(base + frag_off[1]));
  };
  // [K,N]: transposed reads, issued from inline asm (hipcc drains vmcnt(0) in front of the ds_read_tr builtins: it cannot prove
  // them independent of the LDS-DMA writes in flight) and retired by the step's own lgkmcnt(0) + an empty asm that names every
  // destination register.  Block (kb, nb) at (kb * 8 + nb) * 256, kb = ks * 4 + lane / 16; lane 4q + p -> stored row q (+ 4), cols 4p.
  // Outputs are EARLY-CLOBBER (a destination sharing a register with an address is overwritten while later reads still need it).
  struct WBuf { frag32 f[WN]; i32x2 r[16]; };        // [N,K]: f;  [K,N]: r[j * 4 + ks * 2 + (k rows 0-3 | 4-7)]
  const unsigned smem_u32 = static_cast<unsigned>(reinterpret_cast<size_t>(smem));
  unsigned tr_lane[2];
  if constexpr (EB == 2) {
    const int grp = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
    const unsigned base = TILE_A_BYTES + (wn >> 1) * 16384 + ((wn & 1) * 4) * 256 + grp * 2048 + pp * 8;
    tr_lane[0] = base + (qq + ((grp & 1) ? 4 : 0)) * 32;          // k rows 0-3 of the block
    tr_lane[1] = base + (qq + ((grp & 1) ? 0 : 4)) * 32;          // k rows 4-7
  } else {   // 1-byte: block (kb, nb) at kb * 1024 + (nb ^ (kb / 2 % 2)) * 128, kb = ks * 8 + 2 * (lane / 16) + half; lane 2q + p -> row q, cols 8p
    const int grp = lane >> 4, qq = (lane & 15) >> 1, pp = lane & 1;
    const unsigned base = TILE_A_BYTES + (wn >> 1) * 16384 + ((wn & 1) * 4) * 128 + grp * 2048 + qq * 16 + pp * 8;
    tr_lane[0] = base + ((0 ^ (grp & 1)) * 128);                  // column tiles 0 and 2 (+ 256)
    tr_lane[1] = base + ((1 ^ (grp & 1)) * 128);                  // column tiles 1 and 3 (+ 256)
  }
#define T128_TR4(R, A0, A1, O0, O1)                                                                                   \
  asm volatile("ds_read_b64_tr_b16 %0, %4 offset:" #O0 "\n\tds_read_b64_tr_b16 %1, %5 offset:" #O0                     \
               "\n\tds_read_b64_tr_b16 %2, %4 offset:" #O1 "\n\tds_read_b64_tr_b16 %3, %5 offset:" #O1                 \
               : "=&v"((R)[0]), "=&v"((R)[1]), "=&v"((R)[2]), "=&v"((R)[3]) : "v"(A0), "v"(A1) : "memory")
#define T128_TR4B(R, A, O0, O1, O2, O3)                                                                               \
  asm volatile("ds_read_b64_tr_b8 %0, %4 offset:" #O0 "\n\tds_read_b64_tr_b8 %1, %4 offset:" #O1                       \
               "\n\tds_read_b64_tr_b8 %2, %4 offset:" #O2 "\n\tds_read_b64_tr_b8 %3, %4 offset:" #O3                   \
               : "=&v"((R)[0]), "=&v"((R)[1]), "=&v"((R)[2]), "=&v"((R)[3]) : "v"(A) : "memory")
  auto issue_w_tr = [&](WBuf& b, int slot, int j) {  // 16 columns of [K,N] weights: four transposed reads (j is a constant after unrolling)
#if T128_TR_BUILTIN
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
    const lds_char* base = smem + slot * STAGE_BYTES + j * 256;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      b.r[j * 4 + e] = __builtin_bit_cast(i32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + tr_lane[e & 1] + (e >> 1) * 8192)));
#else
    const unsigned a0 = smem_u32 + slot * STAGE_BYTES + tr_lane[0], a1 = smem_u32 + slot * STAGE_BYTES + tr_lane[1];
    i32x2* r = b.r + j * 4;
    if constexpr (EB == 2) {
      if (j == 0) T128_TR4(r, a0, a1, 0, 8192);
      else if (j == 1) T128_TR4(r, a0, a1, 256, 8448);
      else if (j == 2) T128_TR4(r, a0, a1, 512, 8704);
      else T128_TR4(r, a0, a1, 768, 8960);
    } else {                                         // r[j * 4 + ks * 2 + (k-block 2 grp | 2 grp + 1)]
      if (j == 0) T128_TR4B(r, a0, 0, 1024, 8192, 9216);
      else if (j == 1) T128_TR4B(r, a1, 0, 1024, 8192, 9216);
      else if (j == 2) T128_TR4B(r, a0, 256, 1280, 8448, 9472);
      else T128_TR4B(r, a1, 256, 1280, 8448, 9472);
    }
#endif
  };
#undef T128_TR4
#undef T128_TR4B
  auto retire_w_tr = [&](WBuf& b) {                  // behind an lgkmcnt(0): the registers are defined from here on
#if !T128_TR_BUILTIN
    asm volatile("" : "+v"(b.r[0]), "+v"(b.r[1]), "+v"(b.r[2]), "+v"(b.r[3]), "+v"(b.r[4]), "+v"(b.r[5]), "+v"(b.r[6]), "+v"(b.r[7]));
    asm volatile("" : "+v"(b.r[8]), "+v"(b.r[9]), "+v"(b.r[10]), "+v"(b.r[11]), "+v"(b.r[12]), "+v"(b.r[13]), "+v"(b.r[14]), "+v"(b.r[15]));
#endif
#pragma unroll
    for (int j = 0; j < WN; ++j)
      b.f[j] = frag32{b.r[j * 4][0], b.r[j * 4][1], b.r[j * 4 + 1][0], b.r[j * 4 + 1][1],
                      b.r[j * 4 + 2][0], b.r[j * 4 + 2][1], b.r[j * 4 + 3][0], b.r[j * 4 + 3][1]};
  };

  typename Epi::Pre pre;                             // the epilogue's scales, requested now, used after the K loop
  epi.preload(pre, m0 + wm * 64 + (lane & 15), m_end, n0 + wn * 64 + (lane >> 4) * 4, a.N);

  acc_t acc[4][WN];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[i][j] = acc_t{0, 0, 0, 0};       // (f32x4 or i32x4)

  // ---- K loop -----------------------------------------------------------------------------------------------------------
  // The fragments are double-buffered in registers: while the MFMAs of K-tile t run, the fragment reads of K-tile t + 1 are in
  // flight.  Once a wave's reads of K-tile t have retired and the barrier is passed, slot t % S is free for stage t + S.
  frag32 f0a[4], f1a[4];
  WBuf w0, w1;
  auto step = [&](int t, const frag32 (&ca)[4], WBuf& cw, frag32 (&na)[4], WBuf& nw) {
    // stages t + 2 .. t + S - 1 may still be in flight; this wave's reads of K-tile t have retired
    // (the builtin, not asm: the compiler's own wait-count pass has to see that the older reads have retired, or it puts an
    // lgkmcnt(0) in front of the MFMAs — behind the reads just issued)
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(PIECES * (S - 2), 0));
    if constexpr (W_NMAJOR) retire_w_tr(cw);
    __builtin_amdgcn_s_barrier();                    // stage t + 1 has landed for every wave; nobody reads slot t % S any more
    // EIGHT SLICES of 4 MFMAs (slice g = k-block g / 4, A row tile g % 4, all four column tiles).  Each slice carries one of the
    // eight fragments of K-tile t + 1 (A's row tiles, then W's column tiles: two ds_read_b128, or four transposed reads) and one
    // of the LDS-DMA requests of stage t + S.  The issue order is the point: with one wave per SIMD an instruction in front of
    // the MFMAs costs its whole issue time with the matrix unit idle.  The source order read, request, read, request ... is what
    // lets the scheduler spread both: it orders every LDS read and every LDS-DMA request as written (it cannot tell the slots
    // apart), so reads written after the requests could only follow all of them.
    const int rs = (t + 1) % S, ws = t % S, kt = t + S < nkt ? t + S : nkt - 1;     // (tail: re-request the last K-tile into a free
#pragma unroll                                                                      //  slot: keeps the vmcnt bookkeeping uniform)
    for (int g = 0; g < 8; ++g) {
      const int p = g - (8 - PIECES);                // the slice's request, if any
      if (T128_ABLATE != 3) {
        if (g < 4) read_a(na[g], rs, g);
        else if constexpr (!W_NMAJOR) read_w(nw.f[g - 4], rs, g - 4);
        else {
          if (T128_SCHED) __builtin_amdgcn_sched_barrier(0);
          issue_w_tr(nw, rs, g - 4);
          if (T128_SCHED) __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (T128_ABLATE != 2 && p >= 0) stage_piece(kt, ws, p);
      if constexpr (T128_ABLATE < 3) {
        if constexpr (P::KS == 2) {                  // two k-blocks per K-tile: slice g = k-block g / 4, A row tile g % 4
#pragma unroll
          for (int j = 0; j < WN; ++j) acc[g & 3][j] = P::mma(cw.f[j], ca[g & 3], acc[g & 3][j], g >> 2);
        } else {                                      // one MFMA per (row tile, column tile) and K-tile (fp8 16x16x128): two per slice
#pragma unroll
          for (int j = (g & 1) * 2; j < (g & 1) * 2 + 2; ++j) acc[g >> 1][j] = P::mma(cw.f[j], ca[g >> 1], acc[g >> 1][j], 0);
        }
      } else if constexpr (T128_ABLATE == 4) {        // keep the reads alive
        acc[g & 3][0][0] += __builtin_bit_cast(float, ca[g & 3][0] ^ cw.f[g & 3][0]);
      }
      if constexpr (T128_SCHED != 0 && T128_ABLATE < 3) {
        if (W_NMAJOR && g >= 4) {                    // (the transposed reads sit in front of the slice, pinned)
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          if (p >= 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        } else if constexpr (P::KS == 2) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          if (p >= 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        } else {                                      // (fp8's MFMA is inline asm: the groups see only the reads and the request)
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          if (p >= 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
#pragma unroll
  for (int p = 0; p < S; ++p) stage(p, p);
  __builtin_amdgcn_s_waitcnt(waitcnt_imm(PIECES * (S - 1), 15));
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < 4; ++i) read_a(f0a[i], 0, i);
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    if constexpr (W_NMAJOR) issue_w_tr(w0, 0, j); else read_w(w0.f[j], 0, j);
  }
  int t = 0;
  for (; t + 1 < nkt; t += 2) {
    step(t, f0a, w0, f1a, w1);
    step(t + 1, f1a, w1, f0a, w0);
  }
  if (t < nkt) step(t, f0a, w0, f1a, w1);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // (the tail's surplus requests and reads)
  // The last step's transposed reads fetch a K-tile nobody multiplies: to the compiler their destinations are dead the moment
  // the asm ends, and it reused them (for the epilogue's addresses) while the reads were still in flight — the LDS data then
  // landed in an address register (a fault at K = 64, M 129, [K,N]).  Naming both buffers behind the wait keeps every
  // destination allocated until the reads have retired.  tests/test_isa_async_reads.py checks the compiled kernels for it.
  if constexpr (W_NMAJOR) { retire_w_tr(w0); retire_w_tr(w1); }

  // ---- epilogue ------------------------------------------------------------------------------------------------------------
  // (Row-staged stores — the wave's 64 x 64 tile transposed through 8 KiB of the free ring and written as whole 128-byte rows,
  // 8 instructions instead of 16 — were measured and NOT kept: 24.5 / 32.1 us against 24.1 / 32.0 on 128 x 128 tiles at K 4096,
  // 12.7 against 12.2 at K 1024, 2 % better on the eight-wave shape: the extra barrier and the LDS round trip cost what the
  // store pattern saves.  DESIGN Appendix A 29.)
  // direct stores: a lane owns row m = ... + (lane & 15) and 4 consecutive columns of each 16 x 16 tile
  if (a.splitk > 1) {                                  // raw accumulators (fp32 / int32) of this K slice -> slab[kslice][m][n]: the caller's
    acc_t* slab = static_cast<acc_t*>(a.slab);         // finalize launch sums the slices in index order (and dequantises)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wm * 64 + i * 16 + (lane & 15);
      if (m >= m_end) continue;
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int n = n0 + wn * (WN * 16) + j * 16 + (lane >> 4) * 4;
        if (n + 4 <= a.N) slab[((static_cast<int64_t>(kslice) * a.slab_rows + m) * a.N + n) / 4] = acc[i][j];
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + (lane & 15);
    if (m >= m_end) continue;
    const int mc = map_row(m, a.c_rc, a.c_ml, a.c_off, a.c_mul);
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int n = n0 + wn * (WN * 16) + j * 16 + (lane >> 4) * 4;
      if (n >= a.N) continue;
      if constexpr (T128_ABLATE == 1) {              // timing only: keep the accumulators alive, store (almost) nothing
        if (__builtin_bit_cast(int, acc[i][j][0]) != 0x7fc12345) continue;
      }
      epi.store_pre(pre, i, j, mc, n, a.N, acc[i][j]);
    }
  }
}

template <typename P, typename Epi, int NWN, int S>
inline int launch_shape(const GemmArgs& a, const Epi& epi, int64_t m_total, hipStream_t s) {
  constexpr int BN = NWN * 64, LDS = S * (TILE_A_BYTES + BN * KT_BYTES);
  const int64_t tiles = (a.uniform_rows > 0 ? static_cast<int64_t>(a.G) * ceil_div(a.uniform_rows, BM) : ceil_div(m_total, BM) + a.G) * ceil_div(a.N, BN) * a.splitk;   // (ragged: an upper bound; surplus workgroups exit)
  MOJO_REQUIRE(a.splitk == 1 || (a.G == 1 && a.uniform_rows > 0 && a.slab && a.N % 4 == 0 && a.K / (KT_BYTES / P::EB) >= a.splitk),
               MOJO_EUNSUPPORTED, "gemm(128-row tiles): split-K preconditions not met");
  MOJO_REQUIRE(tiles < (1LL << 31), MOJO_EUNSUPPORTED, "gemm(128-row tiles): grid too large");
  // [K,N] weights: 16-bit and int8.  (fp8: its MFMA is inline asm with the accumulator tied in place, the kernel then sits at
  // 256 registers with spills, and hipcc splits the live ranges of the transposed reads' destinations — copies them while
  // the reads are in flight: scripts/check_async_lds_reads.py flags 30 such moves.  Not instantiated; the 256 x 256 kernel runs it.)
  if constexpr (P::EB == 2 || P::KS == 2) {
    if (a.w_n == 1) {                                 // [K,N]
      auto* fn = gemm128_kernel<P, Epi, NWN, S, true>;
      static std::atomic<uint64_t> attr_set{0};
      if (first_call_on_device(attr_set)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      hipLaunchKernelGGL(fn, dim3(static_cast<unsigned>(tiles)), dim3(128 * NWN), LDS, s, a, epi);
      MOJO_CHECK_LAUNCH("gemm(128-row tiles)");
      note_launch("gemm128:%dx%d:KN%s", BM, BN, a.splitk > 1 ? ":splitk" : "");
      return MOJO_OK;
    }
  }
  MOJO_REQUIRE(a.w_k == 1, MOJO_EUNSUPPORTED, "gemm(128-row tiles): weight layout");
  auto* fn = gemm128_kernel<P, Epi, NWN, S, false>;
  static std::atomic<uint64_t> attr_set{0};
  if (first_call_on_device(attr_set)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  hipLaunchKernelGGL(fn, dim3(static_cast<unsigned>(tiles)), dim3(128 * NWN), LDS, s, a, epi);
  MOJO_CHECK_LAUNCH("gemm(128-row tiles)");
  note_launch("gemm128:%dx%d:NK%s", BM, BN, a.splitk > 1 ? ":splitk" : "");
  return MOJO_OK;
}

// 128 x 128 tiles (ring of 4 stages, 128 KiB) while they number at most one per CU; beyond, 128 x 256 tiles (ring of 3, 144 KiB):
// a 128 x 128 tile is bound by its fill (32 KiB per K-tile at ~90 GB/s per CU = 0.36 us for 0.21-0.29 us of MFMAs), the wider tile
// loads 48 KiB for twice the MFMAs.  MOJO_HIP_GEMM_TILE128 = 128 / 256 forces a shape.
template <typename P, typename Epi>
inline int launch(const GemmArgs& a, const Epi& epi, int64_t m_total, hipStream_t s) {
  const long long f = MOJO_SWITCH("MOJO_HIP_GEMM_TILE128", -1);
  const bool wide = f == 256 || (f != 128 && (ceil_div(m_total, BM) + (a.uniform_rows > 0 ? 0 : a.G / 2)) * ceil_div(a.N, 128) > g256::device_cu_count());
  return wide ? launch_shape<P, Epi, 4, 3>(a, epi, m_total, s) : launch_shape<P, Epi, 2, 4>(a, epi, m_total, s);
}

// the switch's verdict: 0 = never, 1 = always (where the caller's preconditions hold), -1 = the caller's time model decides
inline int forced_choice() {
  const long long f = MOJO_SWITCH("MOJO_HIP_GEMM_TILE128", -1);
  if (f == 0) return 0;
  if (f > 0) return 1;                                              // 1; 128 / 256 also force the tile shape
  if (MOJO_SWITCH("MOJO_HIP_GEMM_SPLITK", 0) > 1) return 0;         // a forced split alone is a split of the 256 kernel
  return -1;
}

}  // namespace g128
}  // namespace mojo
