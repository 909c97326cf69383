// 128 x 128-tile MFMA GEMM for UNDER-FILLED launches of the 256 x 256 kernel (round 5).
//
// Role: the dense products behind `MojoGemm.forward` = `F.linear(input, weight, bias)` (core/operators/gemm.py:45-46) and the
// GEMM halves of the GEMM + collective operators with `[N, K]` weights (core/operators/compute_with_comm.py:12-24) at MID-SIZE M —
// a chunked prefill of 256..2048 tokens, config 4 at M 1024.  There a product has fewer 256 x 256 tiles than the chip has CUs,
// and gemm256_core.h either runs on part of the chip or cuts K into slices whose fp32 slabs cost more than the matrix work
// (M 1024 x 4096 x 4096: 4 slices, 27 of 48 us are slab traffic).  hipBLASLt switches to 128-row macro tiles for these shapes
// and was 15-35 % faster (profiles/r5_gemm_mid_m.txt).  This kernel is that tile shape, built for one launch shape only
// (measured against both: profiles/r5_gemm_tile128_ab.txt — M 1024 x 4096 x 4096 47.5 -> 34.4 us, hipBLASLt 37.2):
//
//   * 128 x 128 output tile, K-tiles of 128 bytes, FOUR waves (2 x 2), each 64 x 64 of C on v_mfma_f32_16x16x32 (64 accumulator
//     registers); both operands K-major in LDS as row-blocks of 16 rows x 128 bytes, filled by LDS-DMA in pieces of 8 WHOLE rows
//     (a lane's 16-byte chunk goes to slot chunk ^ (row / 2 % 8) of its row: the swizzle is applied on the SOURCE side, and
//     every ds_read_b128 of a fragment is conflict-free).  Whole 128-byte lines per request fill 15 % faster than gemm256_core.h's
//     16-row x 64-byte pieces (0.37 vs 0.44 us per K-tile with nothing else in the loop);
//   * a ring of 4 stages of 32 KiB (three K-tiles in flight), ONE barrier per K-tile, fragments double-buffered in registers:
//     wait for stage t + 1 and for the own reads of K-tile t, barrier, then the 32 MFMAs of K-tile t with the 16 fragment reads
//     of K-tile t + 1 spread over the first half of them and the 8 LDS-DMA requests of stage t + 4 (into slot t % 4: everybody
//     has its fragments of t in registers) over the second half.  The issue order is the point: with one wave per SIMD an
//     instruction in front of the MFMAs costs its whole issue time with the matrix unit idle — reads and requests ahead of the
//     MFMAs, the compiler's own order, took 0.57 us per K-tile, interleaved 0.41 (K 4096, one tile per CU on an eighth of the
//     chip; scripts/probes/tile128_anatomy.hip, profiles/r5_tile128_anatomy.txt).  What bounds it now is the per-CU fill rate:
//     32 KiB per K-tile at ~90 GB/s per CU = 0.36 us, against 0.21-0.29 us of MFMA time;
//   * beyond one tile per CU the SAME wave code on a 128 x 256 tile: EIGHT waves (2 x 4, two per SIMD), ring of 3 stages of
//     48 KiB — half again the fill for twice the MFMAs.  (A 64 x 128 wave tile on four waves needs the accumulators in AGPRs,
//     and hipcc then shuttles fragments through them; two workgroups of the 128 x 128 shape per CU on a two-stage ring were
//     the first form of this range: 63 us at M 2048 x 4096 x 4096 against 57 now, 71 on 256 x 256 tiles, hipBLASLt 58);
//   * tiles in the 256 kernel's order (bijective XCD remap, panels of 8 n-tiles, m-tile by m-tile inside a panel); row maps on
//     both sides; bias with the golden's rounding (F.linear: in the accumulator, one rounding; x @ w + b: after the rounding).
//
// Same fp32 accumulation per output element over K in index order as the 256 kernel's unsplit form.
#include "gemm256_core.h"

#ifndef T128_SCHED             // issue order of a step: 0 = the compiler's, 1 = fragment reads between the MFMAs, 2 = LDS-DMA requests too
#define T128_SCHED 2
#endif
#ifndef T128_ABLATE            // scripts/probes/tile128_anatomy.hip compiles this file with 2 / 3 / 4 (timing only, wrong results)
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
template <typename P, typename Epi, int NWN, int S>
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
  const int m_tiles = (a.uniform_rows + BM - 1) / BM;
  const int total = m_tiles * n_tiles;
  const int bid = blockIdx.x;
  if (bid >= total) return;
  int tile;
  {  // bijective XCD remap: blocks b, b + 8, ... share an XCD; each XCD gets one contiguous run of tiles
    const int q = total >> 3, r = total & 7, x = bid & 7, i = bid >> 3;
    tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
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
  const int m0 = mi * BM, m_end = a.uniform_rows, n0 = ni * BN;
  const int nkt = a.K / BK;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / NWN, wn = wave % NWN;

  // ---- staging: wave w fills A's row-blocks AB * w ... and W's row-blocks 2w, 2w + 1 (16 rows x 128 bytes each), in pieces of
  // 8 WHOLE rows: lane l -> row l / 8 of the piece, LDS slot l % 8 of that row, which holds the row's 16-byte chunk (l % 8) ^ (row / 2 % 8)
  const char* srcA[2 * AB];
  const char* srcW[2 * WB];
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
      int n = n0 + (WB * wave + (h >> 1)) * 16 + r;
      if (n >= a.N) n = a.N - 1;
      srcW[h] = static_cast<const char*>(a.W) + static_cast<int64_t>(n) * a.w_n * EB + chunk * 16;
    }
  }
  auto stage = [&](int kt, int slot) {               // PIECES LDS-DMA instructions per wave
    if (kt >= nkt) kt = nkt - 1;                     // (keeps the vmcnt bookkeeping uniform at the tail; the slot is a free one)
    lds_char* dst = smem + slot * STAGE_BYTES;
    const int64_t ko = static_cast<int64_t>(kt) * KT_BYTES;
#pragma unroll
    for (int h = 0; h < 2 * AB; ++h) glds16(srcA[h] + ko, dst + (AB * wave) * 2048 + h * 1024);
#pragma unroll
    for (int h = 0; h < 2 * WB; ++h) glds16(srcW[h] + ko, dst + TILE_A_BYTES + (WB * wave) * 2048 + h * 1024);
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
  auto read_a = [&](frag32 (&f)[4], int slot) {
    const lds_char* base = smem + slot * STAGE_BYTES + (wm * 4) * 2048;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      f[i] = join(*reinterpret_cast<lds_frag_ptr>(base + i * 2048 + frag_off[0]), *reinterpret_cast<lds_frag_ptr>(base + i * 2048 + frag_off[1]));
  };
  auto read_w = [&](frag32 (&f)[WN], int slot) {
    const lds_char* base = smem + slot * STAGE_BYTES + TILE_A_BYTES + (wn * WN) * 2048;
#pragma unroll
    for (int j = 0; j < WN; ++j)
      f[j] = join(*reinterpret_cast<lds_frag_ptr>(base + j * 2048 + frag_off[0]), *reinterpret_cast<lds_frag_ptr>(base + j * 2048 + frag_off[1]));
  };

  acc_t acc[4][WN];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[i][j] = acc_t{0, 0, 0, 0};

  // ---- K loop -----------------------------------------------------------------------------------------------------------
  // The fragments are double-buffered in registers: while the MFMAs of K-tile t run, the fragment reads of K-tile t + 1 are in
  // flight.  Once a wave's reads of K-tile t have retired and the barrier is passed, slot t % S is free for stage t + S.
  frag32 f0a[4], f0w[WN], f1a[4], f1w[WN];
  auto step = [&](int t, const frag32 (&ca)[4], const frag32 (&cw)[WN], frag32 (&na)[4], frag32 (&nw)[WN]) {
    // stages t + 2 .. t + S - 1 may still be in flight; this wave's reads of K-tile t have retired
    // (the builtin, not asm: the compiler's own wait-count pass has to see that the older reads have retired, or it puts an
    // lgkmcnt(0) in front of the MFMAs — behind the reads just issued)
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(PIECES * (S - 2), 0));
    __builtin_amdgcn_s_barrier();                    // stage t + 1 has landed for every wave; nobody reads slot t % S any more
    // (reads first: the compiler orders every LDS read behind every earlier LDS-DMA request — it cannot tell the slots apart —
    // so with the requests in front the reads could not be spread over the MFMAs)
    if constexpr (T128_ABLATE != 3) {
      read_a(na, (t + 1) % S);
      read_w(nw, (t + 1) % S);
    }
    if constexpr (T128_ABLATE != 2) stage(t + S, t % S);
    if constexpr (T128_ABLATE < 3) {
      if constexpr (T128_SCHED == 0) __builtin_amdgcn_s_setprio(1);     // (s_setprio is a scheduling boundary: not with an issue order)
#pragma unroll
      for (int ks = 0; ks < P::KS; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j) acc[i][j] = P::mma(cw[j], ca[i], acc[i][j], ks);
      if constexpr (T128_SCHED == 0) __builtin_amdgcn_s_setprio(0);
      // issue order inside the step: the fragment reads of K-tile t + 1 and the LDS-DMA requests BETWEEN the MFMAs of K-tile t —
      // in front of them they cost their full issue time with the matrix unit idle (one wave per SIMD issues in order)
      constexpr int READS = 2 * (4 + WN), MFMAS = 8 * WN;
      if constexpr (T128_SCHED == 1) {
#pragma unroll
        for (int g = 0; g < READS; ++g) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
      } else if constexpr (T128_SCHED == 2) {        // reads over the first half of the MFMAs, requests over the second half
#pragma unroll
        for (int g = 0; g < READS / 2; ++g) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); }
#pragma unroll
        for (int g = 0; g < PIECES; ++g) { __builtin_amdgcn_sched_group_barrier(0x008, (MFMAS - READS) / PIECES, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
      }
    } else if constexpr (T128_ABLATE == 4) {          // keep the reads alive
#pragma unroll
      for (int i = 0; i < 4; ++i) { acc[i][0][0] += __builtin_bit_cast(float, ca[i][0] ^ cw[i][0]); acc[i][1][0] += __builtin_bit_cast(float, ca[i][7] ^ cw[i][7]); }
    }
  };
#pragma unroll
  for (int p = 0; p < S; ++p) stage(p, p);
  __builtin_amdgcn_s_waitcnt(waitcnt_imm(PIECES * (S - 1), 15));
  __builtin_amdgcn_s_barrier();
  read_a(f0a, 0);
  read_w(f0w, 0);
  int t = 0;
  for (; t + 1 < nkt; t += 2) {
    step(t, f0a, f0w, f1a, f1w);
    step(t + 1, f1a, f1w, f0a, f0w);
  }
  if (t < nkt) step(t, f0a, f0w, f1a, f1w);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // (the tail's surplus requests and reads)

  // ---- epilogue: a lane owns row m = ... + (lane & 15) and 4 consecutive columns of each 16 x 16 tile ----------------------
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + (lane & 15);
    if (m >= m_end) continue;
    epi.row_begin(m);
    const int mc = map_row(m, a.c_rc, a.c_ml, a.c_off, a.c_mul);
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int n = n0 + wn * (WN * 16) + j * 16 + (lane >> 4) * 4;
      if (n >= a.N) continue;
      epi.store(mc, n, a.N, acc[i][j]);
    }
  }
}

template <typename P, typename T, int NWN, int S>
static int launch_shape(const GemmArgs& a, int64_t m_total, hipStream_t s) {
  constexpr int BN = NWN * 64, LDS = S * (TILE_A_BYTES + BN * KT_BYTES);
  g256::EpiloguePlain<T> epi{static_cast<T*>(a.C), a.ldc, static_cast<const T*>(a.bias), a.bias_fused != 0};
  const int64_t tiles = ceil_div(m_total, BM) * ceil_div(a.N, BN);
  MOJO_REQUIRE(tiles < (1LL << 31), MOJO_EUNSUPPORTED, "gemm(128-row tiles): grid too large");
  auto* fn = gemm128_kernel<P, g256::EpiloguePlain<T>, NWN, S>;
  static std::atomic<uint64_t> attr_set{0};
  if (first_call_on_device(attr_set)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  hipLaunchKernelGGL(fn, dim3(static_cast<unsigned>(tiles)), dim3(128 * NWN), LDS, s, a, epi);
  MOJO_CHECK_LAUNCH("gemm(128-row tiles)");
  note_launch("gemm128:%dx%d", BM, BN);
  return MOJO_OK;
}

// 128 x 128 tiles (ring of 4 stages, 128 KiB) while they number at most one per CU; beyond, 128 x 256 tiles (ring of 3, 144 KiB):
// a 128 x 128 tile is bound by its fill (32 KiB per K-tile at ~90 GB/s per CU = 0.36 us for 0.21-0.29 us of MFMAs), the wider tile
// loads 48 KiB for twice the MFMAs.  MOJO_HIP_GEMM_TILE128 = 128 / 256 forces a shape.
template <typename P, typename T>
static int launch(const GemmArgs& a, int64_t m_total, hipStream_t s) {
  const long long f = MOJO_SWITCH("MOJO_HIP_GEMM_TILE128", -1);
  const bool wide = f == 256 || (f != 128 && ceil_div(m_total, BM) * ceil_div(a.N, 128) > g256::device_cu_count());
  return wide ? launch_shape<P, T, 4, 3>(a, m_total, s) : launch_shape<P, T, 2, 4>(a, m_total, s);
}

}  // namespace g128

// One dense product (G = 1), 16-bit, [N, K] weights, whole K-tiles, unsplit.
bool gemm_tile128_ok(const GemmArgs& a, int dtype) {
  if (dtype != MOJO_BF16 && dtype != MOJO_F16) return false;
  return a.G == 1 && a.uniform_rows > 0 && a.w_k == 1 && a.K >= 64 && a.K % 64 == 0 && a.lda % 8 == 0 && a.w_n % 8 == 0 && a.ldc % 4 == 0 &&
         a.splitk == 1 && !a.glu && a.a_k_wrap == 0 && aligned_to(a.A, 16) && aligned_to(a.W, 16) && aligned_to(a.C, 8) &&
         (!a.bias || aligned_to(a.bias, 2));
}

// Where the 128-row tiles are taken: more than 128 rows (the decode-sized kernels own those) and the caller's time model
// (gemm_api.hip, gemm_dense_prefers_tile128) says so; MOJO_HIP_GEMM_TILE128 = 1 / 0: wherever they apply / never.
bool gemm_tile128_use(const GemmArgs& a, int dtype, int64_t m_total, bool model_prefers) {
  if (!gemm_tile128_ok(a, dtype) || m_total <= 128) return false;
  const long long f = MOJO_SWITCH("MOJO_HIP_GEMM_TILE128", -1);
  if (f == 0) return false;
  if (f > 0) return true;                                           // 1; 128 / 256 also force the tile shape
  if (MOJO_SWITCH("MOJO_HIP_GEMM_SPLITK", 0) > 1) return false;     // a forced split is a split of the 256 kernel
  return model_prefers;
}

int launch_gemm_tile128(const GemmArgs& a, int dtype, int64_t m_total, hipStream_t s) {
  MOJO_REQUIRE(gemm_tile128_ok(a, dtype), MOJO_EUNSUPPORTED, "gemm(128-row tiles): preconditions not met");
  return dtype == MOJO_BF16 ? g128::launch<g256::PolBF16, bf16_t>(a, m_total, s) : g128::launch<g256::PolF16, f16_t>(a, m_total, s);
}

}  // namespace mojo
