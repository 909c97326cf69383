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
//   * a ring of S stages of 32 KiB: S = 4 (128 KiB, one workgroup per CU, three K-tiles in flight) when the launch has at most
//     one tile per CU, S = 2 (64 KiB, two workgroups per CU) otherwise.  ONE barrier per K-tile, fragments double-buffered in
//     registers: wait for stage t + 1 and for the own reads of K-tile t, barrier, then the 32 MFMAs of K-tile t with the 16
//     fragment reads of K-tile t + 1 spread over the first half of them and the 8 LDS-DMA requests of stage t + S (into slot
//     t % S: everybody has its fragments of t in registers) over the second half.  The issue order is the point: with one
//     wave per SIMD an instruction in front of the MFMAs costs its whole issue time with the matrix unit idle — reads and
//     requests ahead of the MFMAs, the compiler's own order, took 0.57 us per K-tile, interleaved 0.41 (K 4096, one tile per
//     CU on an eighth of the chip; scripts/probes/tile128_anatomy.hip, profiles/r5_tile128_anatomy.txt).  What bounds it now is
//     the per-CU fill rate: 32 KiB per K-tile at ~90 GB/s per CU = 0.36 us, against 0.21-0.29 us of MFMA time;
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

constexpr int BM = 128, BN = 128;
constexpr int KT_BYTES = g256::KT_BYTES;            // 128 bytes of K per row and K-tile
constexpr int TILE_BYTES = 128 * KT_BYTES;          // one operand's K-tile: 16 KiB = 16 sub-tiles of 16 rows x 64 bytes
constexpr int STAGE_BYTES = 2 * TILE_BYTES;         // A | W
constexpr int PANEL = 8;                            // n-tiles per panel

template <typename P, typename Epi, int S>
__global__ __launch_bounds__(256, 2) void gemm128_kernel(GemmArgs a, Epi epi) {
  typedef typename P::acc_t acc_t;
  constexpr int EB = P::EB;
  constexpr int BK = KT_BYTES / EB;
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
  const int wm = wave >> 1, wn = wave & 1;

  // ---- staging: wave w fills row-blocks 2w and 2w + 1 (16 rows x 128 bytes each) of BOTH operands, in pieces of 8 WHOLE rows ----
  // lane l -> row l / 8 of the piece, LDS slot l % 8 of that row, which holds the row's 16-byte chunk (l % 8) ^ ((row / 2) % 8)
  const char* srcA[4];
  const char* srcW[4];
  {
    const int rr = lane >> 3, p = lane & 7;
#pragma unroll
    for (int h = 0; h < 4; ++h) {                    // h = row-block (h / 2), piece (h % 2)
      const int r = (h & 1) * 8 + rr;
      const int chunk = p ^ ((r >> 1) & 7);
      int m = m0 + (2 * wave + (h >> 1)) * 16 + r;
      if (m >= m_end) m = m_end - 1;                 // rows past the end: re-read a valid row, never stored
      srcA[h] = static_cast<const char*>(a.A) + static_cast<int64_t>(map_row(m, a.a_rc, a.a_ml, a.a_off, a.a_mul)) * a.lda * EB + chunk * 16;
      int n = n0 + (2 * wave + (h >> 1)) * 16 + r;
      if (n >= a.N) n = a.N - 1;
      srcW[h] = static_cast<const char*>(a.W) + static_cast<int64_t>(n) * a.w_n * EB + chunk * 16;
    }
  }
  auto stage = [&](int kt, int slot) {               // 8 LDS-DMA instructions per wave
    if (kt >= nkt) kt = nkt - 1;                     // (keeps the vmcnt bookkeeping uniform at the tail; the slot is a free one)
    lds_char* dst = smem + slot * STAGE_BYTES + (2 * wave) * 2048;
    const int64_t ko = static_cast<int64_t>(kt) * KT_BYTES;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      glds16(srcA[h] + ko, dst + h * 1024);
      glds16(srcW[h] + ko, dst + TILE_BYTES + h * 1024);
    }
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
  auto read4 = [&](frag32 (&f)[4], int slot, int operand, int rb0) {
    const lds_char* base = smem + slot * STAGE_BYTES + operand * TILE_BYTES + rb0 * 2048;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      f[i] = join(*reinterpret_cast<lds_frag_ptr>(base + i * 2048 + frag_off[0]), *reinterpret_cast<lds_frag_ptr>(base + i * 2048 + frag_off[1]));
  };

  acc_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = acc_t{0, 0, 0, 0};

  // ---- K loop -----------------------------------------------------------------------------------------------------------
  // The fragments are double-buffered in registers: while the 32 MFMAs of K-tile t run, the 16 fragment reads of K-tile t + 1
  // are in flight (with one wave per SIMD nothing else would cover them: 64 KiB of LDS reads per K-tile and CU are half the
  // MFMA time).  Once a wave's reads of K-tile t have retired and the barrier is passed, slot t % S is free for stage t + S.
  frag32 f0a[4], f0w[4], f1a[4], f1w[4];
  auto step = [&](int t, const frag32 (&ca)[4], const frag32 (&cw)[4], frag32 (&na)[4], frag32 (&nw)[4]) {
    // stages t + 2 .. t + S - 1 may still be in flight (8 requests each); this wave's reads of K-tile t have retired
    // (the builtin, not asm: the compiler's own wait-count pass has to see that the older reads have retired, or it puts an
    // lgkmcnt(0) in front of the MFMAs — behind the reads just issued.  gfx9 encoding: vmcnt [3:0] + [15:14], expcnt [6:4], lgkmcnt [11:8])
    if constexpr (S == 4) __builtin_amdgcn_s_waitcnt(0x4070);      // vmcnt(16) lgkmcnt(0)
    else __builtin_amdgcn_s_waitcnt(0x0070);                        // vmcnt(0) lgkmcnt(0)
    __builtin_amdgcn_s_barrier();                    // stage t + 1 has landed for every wave; nobody reads slot t % S any more
    // (reads first: the compiler orders every LDS read behind every earlier LDS-DMA request — it cannot tell the slots apart —
    // so with the requests in front the reads could not be spread over the MFMAs)
    if constexpr (T128_ABLATE != 3) {
      read4(na, (t + 1) % S, 0, wm * 4);
      read4(nw, (t + 1) % S, 1, wn * 4);
    }
    if constexpr (T128_ABLATE != 2) stage(t + S, t % S);
    if constexpr (T128_ABLATE < 3) {
      if constexpr (T128_SCHED == 0) __builtin_amdgcn_s_setprio(1);     // (s_setprio is a scheduling boundary: not with an issue order)
#pragma unroll
      for (int ks = 0; ks < P::KS; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = P::mma(cw[j], ca[i], acc[i][j], ks);
      if constexpr (T128_SCHED == 0) __builtin_amdgcn_s_setprio(0);
      // issue order inside the step: the fragment reads of K-tile t + 1 (and the LDS-DMA requests) BETWEEN the MFMAs of K-tile
      // t — in front of them they cost their full issue time with the matrix unit idle (one wave per SIMD issues in order)
      if constexpr (T128_SCHED == 1) {
#pragma unroll
        for (int g = 0; g < 16; ++g) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
      } else if constexpr (T128_SCHED == 2) {
#pragma unroll
        for (int g = 0; g < 8; ++g) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); }
#pragma unroll
        for (int g = 0; g < 8; ++g) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
      }
    } else if constexpr (T128_ABLATE == 4) {          // keep the reads alive
#pragma unroll
      for (int i = 0; i < 4; ++i) { acc[i][0][0] += __builtin_bit_cast(float, ca[i][0] ^ cw[i][0]); acc[i][1][0] += __builtin_bit_cast(float, ca[i][7] ^ cw[i][7]); }
    }
  };
#pragma unroll
  for (int p = 0; p < S; ++p) stage(p, p);
  if constexpr (S == 4) __builtin_amdgcn_s_waitcnt(0x4F78);        // vmcnt(24)
  else __builtin_amdgcn_s_waitcnt(0x0F78);                          // vmcnt(8)
  __builtin_amdgcn_s_barrier();
  read4(f0a, 0, 0, wm * 4);
  read4(f0w, 0, 1, wn * 4);
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
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
      if (n >= a.N) continue;
      epi.store(mc, n, a.N, acc[i][j]);
    }
  }
}

template <typename P, typename T>
static int launch(const GemmArgs& a, int64_t m_total, hipStream_t s) {
  g256::EpiloguePlain<T> epi{static_cast<T*>(a.C), a.ldc, static_cast<const T*>(a.bias), a.bias_fused != 0};
  const int64_t tiles = ceil_div(m_total, BM) * ceil_div(a.N, BN);
  MOJO_REQUIRE(tiles < (1LL << 31), MOJO_EUNSUPPORTED, "gemm(128-row tiles): grid too large");
  const bool one_per_cu = tiles <= g256::device_cu_count();
  if (one_per_cu) {
    auto* fn = gemm128_kernel<P, g256::EpiloguePlain<T>, 4>;
    static std::atomic<uint64_t> attr_set{0};
    if (first_call_on_device(attr_set)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * STAGE_BYTES);
    hipLaunchKernelGGL(fn, dim3(static_cast<unsigned>(tiles)), dim3(256), 4 * STAGE_BYTES, s, a, epi);
  } else {
    auto* fn = gemm128_kernel<P, g256::EpiloguePlain<T>, 2>;
    static std::atomic<uint64_t> attr_set{0};
    if (first_call_on_device(attr_set)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
    hipLaunchKernelGGL(fn, dim3(static_cast<unsigned>(tiles)), dim3(256), 2 * STAGE_BYTES, s, a, epi);
  }
  MOJO_CHECK_LAUNCH("gemm(128-row tiles)");
  note_launch("gemm128:%s", one_per_cu ? "ring4" : "ring2");
  return MOJO_OK;
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
  if (f == 1) return true;
  if (MOJO_SWITCH("MOJO_HIP_GEMM_SPLITK", 0) > 1) return false;     // a forced split is a split of the 256 kernel
  return model_prefers;
}

int launch_gemm_tile128(const GemmArgs& a, int dtype, int64_t m_total, hipStream_t s) {
  MOJO_REQUIRE(gemm_tile128_ok(a, dtype), MOJO_EUNSUPPORTED, "gemm(128-row tiles): preconditions not met");
  return dtype == MOJO_BF16 ? g128::launch<g256::PolBF16, bf16_t>(a, m_total, s) : g128::launch<g256::PolF16, f16_t>(a, m_total, s);
}

}  // namespace mojo
