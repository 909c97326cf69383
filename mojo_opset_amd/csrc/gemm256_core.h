// 256x256 MFMA GEMM core for gfx950, shared by the bf16/fp16 grouped/dense GEMM and the int8/fp8
// quantised GEMM.  See gemm_mfma256.hip for the design notes (LDS images, staggered 8-phase schedule).
//
// Everything is expressed in BYTES along K: a K-tile is 128 bytes of K per row (64 bf16 / 128 int8|fp8
// elements), a fragment is 16 bytes per lane, so the staging, the LDS images and the read addresses are
// identical for every element type; only the MFMA call (Policy) and the epilogue differ.
#pragma once
#include <stdlib.h>

#include "gemm.h"

#ifndef GEMM256_LINE_PIECES       // K-major LDS-DMA pieces: 1 = 8 whole rows of 128 bytes (gemm_tile128_core.h's image), 0 = 16 rows x 64 bytes
#define GEMM256_LINE_PIECES 1
#endif

namespace mojo {
namespace g256 {

constexpr int BM = 256, BN = 256;
constexpr int KT_BYTES = 128;                     // bytes of K per row per K-tile
constexpr int HALF_BYTES = 128 * KT_BYTES;        // 16 KiB
constexpr int KTILE_BYTES = 4 * HALF_BYTES;       // A0 A1 W0 W1
constexpr int LDS_BYTES = 2 * KTILE_BYTES;        // 128 KiB
constexpr int PANEL = 4;                          // n-tiles per panel (A/B: 4 beats 8 by 0.5-2.5 %, 2 and 16 lose)
constexpr int PERSIST_STAGE_BYTES = 8 * 4096;     // persistent form: 4 KiB of epilogue staging per wave, behind the tile buffers
constexpr int PERSIST_EPI_STORES = 16;            // global store instructions per wave of a full tile's row-staged epilogue

typedef __attribute__((address_space(3))) char lds_char;
typedef i32x4 frag16;                             // 16 bytes of K for one row / column
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef i32x8 frag32;                             // both 64-byte k-blocks of a K-tile: [ks=0 | ks=1]

__device__ __forceinline__ frag16 half_of(const frag32& f, int ks) {
  return ks ? __builtin_shufflevector(f, f, 4, 5, 6, 7) : __builtin_shufflevector(f, f, 0, 1, 2, 3);
}
__device__ __forceinline__ frag32 join(frag16 lo, frag16 hi) {
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// ---- element policies --------------------------------------------------------------------------------
struct PolBF16 {
  typedef bf16_t elem; typedef f32x4 acc_t; static constexpr int EB = 2, KS = 2;
  static __device__ __forceinline__ acc_t mma(const frag32& w, const frag32& a, acc_t c, int ks) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, half_of(w, ks)), __builtin_bit_cast(bf16x8, half_of(a, ks)), c, 0, 0, 0);
  }
};
struct PolF16 {
  typedef f16_t elem; typedef f32x4 acc_t; static constexpr int EB = 2, KS = 2;
  static __device__ __forceinline__ acc_t mma(const frag32& w, const frag32& a, acc_t c, int ks) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, half_of(w, ks)), __builtin_bit_cast(f16x8, half_of(a, ks)), c, 0, 0, 0);
  }
};
struct PolI8 {
  typedef int8_t elem; typedef i32x4 acc_t; static constexpr int EB = 1, KS = 2;
  static __device__ __forceinline__ acc_t mma(const frag32& w, const frag32& a, acc_t c, int ks) {
    return __builtin_amdgcn_mfma_i32_16x16x64_i8(half_of(w, ks), half_of(a, ks), c, 0, 0, 0);
  }
};
struct PolF8 {   // OCP e4m3 on v_mfma_f32_16x16x128_f8f6f4 (the CDNA4 form: a whole 128-byte K-tile per instruction, 2x the bf16 rate)
  // A lane's 32 bytes are the four 8-byte pieces the K=32 form took one per instruction; the K=128 form sums over all of
  // them, and since the activation and the weight fragment are cut from their rows the same way, every product pairs the
  // same k on both sides (the order inside the sum is free).  asm with the accumulator as "+v": through the builtin
  // (mfma_scale with zero scales, which the backend turns into this instruction) hipcc does not accumulate in place and
  // the 128 accumulator registers of this tile shape spill.  s_nop 1: a fragment register freshly written by the vector
  // unit (the [K,N] path assembles its fragments) must not be read by an MFMA in the next two states.
  typedef uint8_t elem; typedef f32x4 acc_t; static constexpr int EB = 1, KS = 1;
  static __device__ __forceinline__ acc_t mma(const frag32& w, const frag32& a, acc_t c, int) {
    asm("s_nop 1\n\tv_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0" : "+v"(c) : "v"(w), "v"(a));
    return c;
  }
};

__device__ __forceinline__ void glds16(const void* gsrc, lds_char* dst_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc),
                                   (__attribute__((address_space(3))) void*)(dst_wave_base), 16, 0, 0);
}

// ---- epilogues ---------------------------------------------------------------------------------------
// A lane owns row m = ... + (lane & 15) and the 4 consecutive columns n .. n+3 of each 16x16 tile.
template <typename T>
struct EpiloguePlain {       // C = round_T(acc) (+ bias after the rounding: the golden's `x @ w + b`), or round_T(acc + bias) (its F.linear)
  static constexpr bool kRowStaged = sizeof(T) == 2;   // may go through the wave-private LDS transpose (see the kernel's epilogue)
  // 256 x 256 kernel, row-staged epilogue WITH a bias: every thread of the second half of the workgroup fetches one of the tile's
  // 256 bias values before the K loop and parks it in LDS (EpilogueDequant's mechanism, below).  Until round 5 a bias sent the
  // tile through the direct 8-byte stores with the bias fetched in the epilogue: + 18-25 % on a prefill-sized projection with a
  // bias (8192 x 4096 x 6144: 281 -> 332 us), + 59 % at K 1024.
  static constexpr bool kLdsScales = sizeof(T) == 2;
  __host__ __device__ __forceinline__ bool lds_values() const { return bias != nullptr; }   // (no bias: the epilogue as before)
  __device__ __forceinline__ float scale_for_thread(int t, int, int, int n0, int n_limit) const {
    return (bias && t >= 256) ? static_cast<float>(bias[min(n0 + t - 256, n_limit - 1)]) : 0.f;
  }
  __device__ __forceinline__ typename vec_of<T, 4>::type to4_scaled(f32x4 acc, float, f32x4 col) const {
    typename vec_of<T, 4>::type o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = round_with_bias<T>(acc[e], static_cast<T>(col[e]), bias_fused);   // (a 16-bit value through fp32: exact)
    return o;
  }
  typedef T out_t;
  T* C; int64_t ldc; const T* bias; bool bias_fused = false;
  __host__ __device__ __forceinline__ bool has_bias() const { return bias != nullptr; }
  __device__ __forceinline__ typename vec_of<T, 4>::type to4(int, f32x4 acc) const {
    typename vec_of<T, 4>::type o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = static_cast<T>(acc[e]);
    return o;
  }
  __device__ __forceinline__ void row_begin(int) {}
  // gemm_tile128_core.h: what the epilogue needs from memory, requested BEFORE the K loop: a lane's sixteen bias values (a
  // 64 x 64 wave tile: columns n_first + 16 j + e) — fetched in the epilogue they are an exposed memory latency per launch
  struct Pre { T b[4][4]; };
  __device__ __forceinline__ void preload(Pre& p, int, int, int n_first, int n_limit) const {
    if (!bias) return;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) p.b[j][e] = bias[min(n_first + j * 16 + e, n_limit - 1)];
  }
  __device__ __forceinline__ typename vec_of<T, 4>::type cvt_pre(const Pre& p, int, int j, f32x4 acc) const {
    typename vec_of<T, 4>::type o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = bias ? round_with_bias<T>(acc[e], p.b[j][e], bias_fused) : static_cast<T>(acc[e]);
    return o;
  }
  __device__ __forceinline__ void store_pre(const Pre& p, int i, int j, int m, int n, int n_limit, f32x4 acc) const {
    typedef typename vec_of<T, 4>::type V4;
    const V4 o = cvt_pre(p, i, j, acc);
    T* dst = C + static_cast<int64_t>(m) * ldc + n;
    if (n + 4 <= n_limit) {
      *reinterpret_cast<V4*>(dst) = o;
    } else {
      for (int e = 0; e < 4 && n + e < n_limit; ++e) dst[e] = o[e];
    }
  }
  __device__ __forceinline__ void store(int m, int n, int n_limit, f32x4 acc) const {
    typedef typename vec_of<T, 4>::type V4;
    V4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = static_cast<T>(acc[e]);
    if (bias) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n + e < n_limit) o[e] = round_with_bias<T>(acc[e], bias[n + e], bias_fused);
    }
    T* dst = C + static_cast<int64_t>(m) * ldc + n;
    if (n + 4 <= n_limit) {
      *reinterpret_cast<V4*>(dst) = o;
    } else {
      for (int e = 0; e < 4 && n + e < n_limit; ++e) dst[e] = o[e];
    }
  }
};

struct EpilogueF32 {         // C (fp32) = acc, or C += acc: two-pass products (x @ w_hi, then + x @ w_lo) of the MoE router
  static constexpr bool kRowStaged = false;
  static constexpr bool kLdsScales = false;
  __host__ __device__ __forceinline__ bool lds_values() const { return false; }
  float* C; int64_t ldc; int accumulate;
  __device__ __forceinline__ void row_begin(int) {}
  __device__ __forceinline__ void store(int m, int n, int n_limit, f32x4 acc) const {
    float* dst = C + static_cast<int64_t>(m) * ldc + n;
    if (n + 4 <= n_limit) {
      f32x4 o = acc;
      if (accumulate) o += *reinterpret_cast<const f32x4*>(dst);
      *reinterpret_cast<f32x4*>(dst) = o;
    } else {
      for (int e = 0; e < 4 && n + e < n_limit; ++e) dst[e] = accumulate ? dst[e] + acc[e] : acc[e];
    }
  }
};

template <typename TO, typename ACC>
struct EpilogueDequant {     // C = round_TO( float(acc) * row_scale[m] * col_scale[n] )   (golden: gemm.py:213-223)
  static constexpr bool kRowStaged = sizeof(TO) == 2;
  // 256 x 256 kernel, row-staged epilogue: every thread fetches ONE of the tile's 256 row / 256 column scales before the K loop
  // and parks it in 2 KiB of LDS behind the tile buffers when the loop is over; the epilogue reads its scales from there.  Fetched
  // inside the epilogue they were an exposed memory latency per tile: 3.4 % of the M 4096 x 7168 x 36864 product, 9 % at K 4096
  // (scripts/probes/quant_headline_scales.py: the same kernel with constants in place of the fetches).
  static constexpr bool kLdsScales = sizeof(TO) == 2;
  __host__ __device__ __forceinline__ bool lds_values() const { return true; }
  typedef TO out_t;
  TO* C; int64_t ldc; const float* row_scale; const bf16_t* col_scale;
  float rs;
  bool vec4 = false;         // store(): four outputs in one store where they fit (the caller vouches for ldc % 4 == 0 and an aligned C)
  __host__ __device__ __forceinline__ bool has_bias() const { return false; }
  __device__ __forceinline__ typename vec_of<TO, 4>::type to4(int n, ACC acc) const {   // full tiles only: n + 4 <= N
    typename vec_of<TO, 4>::type o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#ifdef QG_NO_SCALE_LOADS                                  // timing only (scripts/probes/quant_headline_scales.py): what the scale fetches cost
      float v = __fmul_rn(__fmul_rn(static_cast<float>(acc[e]), rs), 0.5f);
#else
      float v = __fmul_rn(__fmul_rn(static_cast<float>(acc[e]), rs), static_cast<float>(col_scale[n + e]));
#endif
      asm volatile("" : "+v"(v));                          // see store(): the fp32 product is a value of its own
      o[e] = elt<TO>::from_f(v);
    }
    return o;
  }
#ifdef QG_NO_SCALE_LOADS
  __device__ __forceinline__ void row_begin(int m) { rs = 0.25f + m * 1e-6f; }
#else
  __device__ __forceinline__ void row_begin(int m) { rs = row_scale[m]; }
#endif
  // thread t of the 512: t < 256 the scale of row m0 + t, else of column n0 + t - 256 (clamped to valid entries)
  __device__ __forceinline__ float scale_for_thread(int t, int m0, int m_limit, int n0, int n_limit) const {
    return t < 256 ? row_scale[min(m0 + t, m_limit - 1)] : static_cast<float>(col_scale[min(n0 + t - 256, n_limit - 1)]);
  }
  __device__ __forceinline__ typename vec_of<TO, 4>::type to4_scaled(ACC acc, float row, f32x4 col) const {
    typename vec_of<TO, 4>::type o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = __fmul_rn(__fmul_rn(static_cast<float>(acc[e]), row), col[e]);
      asm volatile("" : "+v"(v));                          // see store(): the fp32 product is a value of its own
      o[e] = elt<TO>::from_f(v);
    }
    return o;
  }
  // gemm_tile128_core.h: a lane's four row scales and sixteen column scales (a 64 x 64 wave tile: rows m_first + 16 i, columns
  // n_first + 16 j + e), requested BEFORE the K loop — fetched in the epilogue they cost two dependent memory latencies
  // (~6 us of a 20 us launch at K 4096)
  struct Pre { float rs[4]; float cs[4][4]; };
  __device__ __forceinline__ void preload(Pre& p, int m_first, int m_limit, int n_first, int n_limit) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) p.rs[i] = row_scale[min(m_first + i * 16, m_limit - 1)];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) p.cs[j][e] = static_cast<float>(col_scale[min(n_first + j * 16 + e, n_limit - 1)]);
  }
  __device__ __forceinline__ typename vec_of<TO, 4>::type cvt_pre(const Pre& p, int i, int j, ACC acc) const {
    typename vec_of<TO, 4>::type o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = __fmul_rn(__fmul_rn(static_cast<float>(acc[e]), p.rs[i]), p.cs[j][e]);
      asm volatile("" : "+v"(v));                          // see store(): the fp32 product is a value of its own
      o[e] = elt<TO>::from_f(v);
    }
    return o;
  }
  __device__ __forceinline__ void store_pre(const Pre& p, int i, int j, int m, int n, int n_limit, ACC acc) const {
    TO* dst = C + static_cast<int64_t>(m) * ldc + n;
    const typename vec_of<TO, 4>::type o = cvt_pre(p, i, j, acc);
    if (vec4 && n + 4 <= n_limit) {
      *reinterpret_cast<typename vec_of<TO, 4>::type*>(dst) = o;
    } else {
      for (int e = 0; e < 4 && n + e < n_limit; ++e) dst[e] = o[e];
    }
  }
  __device__ __forceinline__ void store(int m, int n, int n_limit, ACC acc) const {
    TO* dst = C + static_cast<int64_t>(m) * ldc + n;
    if (vec4 && n + 4 <= n_limit) {
      *reinterpret_cast<typename vec_of<TO, 4>::type*>(dst) = to4(n, acc);
      return;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (n + e < n_limit) {
        float v = __fmul_rn(__fmul_rn(static_cast<float>(acc[e]), rs), static_cast<float>(col_scale[n + e]));
        // keep the fp32 product a value of its own: hipcc would otherwise fuse multiply + narrowing into
        // v_fma_mixlo_f16 (ONE rounding), while the golden rounds to fp32 first and to the output type second
        asm volatile("" : "+v"(v));
        dst[e] = elt<TO>::from_f(v);
      }
    }
  }
};

#ifdef GEMM_STAMPS
// In-kernel cycle anatomy of the K loop (build with MOJO_HIP_EXTRA_CXXFLAGS=-DGEMM_STAMPS; scripts/probes/gemm_stamps.py):
// lane i of `tacc` accumulates the cycles between stamp i-1 and stamp i; 5 stamps per phase x 4 phases.  Timing tool only.
static __device__ unsigned g_gemm_stamps[8192 * 8 * 32];
#define G_STAMP(i)                                                                \
  do {                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                            \
    const unsigned long long t_ = __builtin_readcyclecounter();                   \
    const unsigned d_ = static_cast<unsigned>(t_ - t_prev);                       \
    t_prev = t_;                                                                  \
    tacc += (lane == (i)) ? d_ : 0u;                                              \
    __builtin_amdgcn_sched_barrier(0);                                            \
  } while (0)
#else
#define G_STAMP(i)
#endif

// ---- the kernel ------------------------------------------------------------------------------------------
// PERSIST: one workgroup per CU walks tiles bid, bid + gridDim.x, ...; the first six half-tiles of the NEXT tile are
// requested before the epilogue of the current one (which then stages through its own 32 KiB instead of the tile
// buffers), so the ~2 us of load latency in front of a tile's first MFMA and the ~3 us of epilogue overlap, and the
// per-tile workgroup launch / teardown disappears.  The fixed cost per tile was 6.8 us of 108.7 at K = 4096 and of 19.6
// at K = 512 (the MLA decompression GEMM).  Only the row-staged 16-bit epilogue without bias / GLU / split-K has this form.
template <typename P, bool W_NMAJOR /* true: W is [K,N] (n contiguous); false: [N,K] */, typename Epi, bool PERSIST = false>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(GemmArgs a, Epi epi) {
  typedef typename P::elem E;
  typedef typename P::acc_t acc_t;
  constexpr int EB = P::EB;
  constexpr int BK = KT_BYTES / EB;                 // elements of K per K-tile
  extern __shared__ __attribute__((aligned(1024))) char smem_generic[];
  lds_char* smem = (lds_char*)smem_generic;

  // ---- which tile ------------------------------------------------------------------------------------
  // glu: a tile is 128 gate columns + the 128 up columns that pair with them (W half-tile 1 starts N/2 columns further)
  const int n_tiles = a.glu ? (a.N / 2) / 128 : (a.N + BN - 1) / BN;
  const int m_tiles = gemm_m_tiles(a, BM);
  const int tiles_mn = m_tiles * n_tiles;
  const int total = tiles_mn * a.splitk;
  int bid = blockIdx.x;
  if (bid >= total) return;
  if constexpr (!PERSIST) {
    // Start stagger (short-K products with many rounds; chosen in gemm256_launch): every tile of such a launch takes the same
    // time, so the workgroups of a round finish together and the chip alternates between a phase where every CU multiplies
    // and one where every CU writes its 128 KiB tile.  The first round starts in eight phases; later workgroups inherit the
    // phase of the slot they take over.  (Whether the gain comes from the write bursts or from the power headroom the idle
    // phases leave is not settled: the product alone gains 12 %, inside the MLA prefill operator the GEMM gains 2 % and the
    // attention kernel behind it 5 %.)
    if (a.stagger_ticks > 0 && bid < a.stagger_blocks) {
      const unsigned phase = (static_cast<unsigned>(bid) >> 3) & 7u;
      if (phase) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long d = static_cast<unsigned long long>(phase) * static_cast<unsigned>(a.stagger_ticks);
        while (__builtin_amdgcn_s_memrealtime() - t0 < d) __builtin_amdgcn_s_sleep(16);
      }
    }
  }
  const int hoff = a.glu ? a.N / 2 : 128;              // column distance between the two W half-tiles
  const int nkt_all = a.K / BK;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const char* A = static_cast<const char*>(a.A);

  // ---- per-tile state: where the tile is, and this lane's global source pointers (bytes) ------------------------------
  int g, m0, m_end, n0;                               // m_end exclusive; m0 < m_end by construction
  int kslice, kt0, nkt;
  const char* srcA[2];
  const char* srcW[2];
  int64_t w_step = 0;                                 // byte advance per K-tile
  int w2_off[2] = {64, 64};                           // byte offset of the wave's second glds
  int a2_off[2] = {64, 64};                           // the same for A (whole-line pieces: eight rows further down)
  auto locate = [&](int bid_) {
    int tile;
    if (a.tile_order == 1) {
      tile = bid_;                                     // measurement: no XCD remap — the eight XCDs share every panel
    } else {  // bijective XCD remap: blocks b, b+8, ... share an XCD; give each XCD one contiguous run of tiles
      const int q = total >> 3, r = total & 7, x = bid_ & 7, i = bid_ >> 3;
      tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    kslice = tile / tiles_mn;                          // split-K slice (0 when splitk == 1)
    tile -= kslice * tiles_mn;
    int mi, ni;
    {  // panel-major order: panels of PANEL n-tiles; inside a panel m-tile by m-tile
      const int PANEL = a.tile_order == 2 ? 2 : mojo::g256::PANEL;
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
    gemm_locate_tile(a, mi, BM, g, m0, m_end);
    n0 = a.glu ? ni * 128 : ni * BN;
    kt0 = static_cast<int>(static_cast<int64_t>(nkt_all) * kslice / a.splitk);
    nkt = static_cast<int>(static_cast<int64_t>(nkt_all) * (kslice + 1) / a.splitk) - kt0;   // K-tiles of this slice
    // K-major half-tile h: wave w fills row-block w (16 rows) with two glds (64-byte k-blocks 0,1).
    //   lane l -> row l/4, 16-byte chunk (l%4) ^ (2 if row >= 8)           [st_16x32 on the source side]
    const char* W = static_cast<const char*>(a.W) + static_cast<int64_t>(g) * a.w_group * EB;
#if GEMM256_LINE_PIECES
    // (round 5) K-major row-block of 16 rows x 128 bytes, filled in two pieces of 8 WHOLE rows: lane l -> row l / 8 (+ 8), LDS slot
    // l % 8 of that row, which holds the row's 16-byte chunk (l % 8) ^ (row / 2 % 8) — half-line requests cost the fill rate 15 %
    // in gemm_tile128_core.h, where this image comes from
    {
      const int rr = lane >> 3, pslot = lane & 7;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int64_t off[2];
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
          const int r = pc * 8 + rr;
          int m = m0 + h * 128 + wave * 16 + r;
          if (m >= m_end) m = m_end - 1;               // rows past the group: re-read a valid row, never stored
          off[pc] = (static_cast<int64_t>(map_row(m, a.a_rc, a.a_ml, a.a_off, a.a_mul)) * a.lda) * EB + (pslot ^ ((r >> 1) & 7)) * 16;
        }
        srcA[h] = A + off[0] + static_cast<int64_t>(a.a_k_wrap ? kt0 % a.a_k_wrap : kt0) * KT_BYTES;
        a2_off[h] = static_cast<int>(off[1] - off[0]);
      }
    }
    if constexpr (!W_NMAJOR) {
      const int rr = lane >> 3, pslot = lane & 7;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int64_t off[2];
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
          const int r = pc * 8 + rr;
          int n = n0 + h * hoff + wave * 16 + r;
          if (n >= a.N) n = a.N - 1;
          off[pc] = (static_cast<int64_t>(n) * a.w_n) * EB + (pslot ^ ((r >> 1) & 7)) * 16;
        }
        srcW[h] = W + off[0];
        w2_off[h] = static_cast<int>(off[1] - off[0]);
      }
      w_step = KT_BYTES;
    } else if constexpr (EB == 2) {
#else
    {
      const int row = lane >> 2;
      const int chunk = (lane & 3) ^ ((row & 8) ? 2 : 0);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int m = m0 + h * 128 + wave * 16 + row;
        if (m >= m_end) m = m_end - 1;                 // rows past the group: re-read a valid row, never stored
        srcA[h] = A + (static_cast<int64_t>(map_row(m, a.a_rc, a.a_ml, a.a_off, a.a_mul)) * a.lda) * EB + chunk * 16 + static_cast<int64_t>(a.a_k_wrap ? kt0 % a.a_k_wrap : kt0) * KT_BYTES;
      }
    }
    if constexpr (!W_NMAJOR) {
      const int row = lane >> 2;
      const int chunk = (lane & 3) ^ ((row & 8) ? 2 : 0);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int n = n0 + h * hoff + wave * 16 + row;
        if (n >= a.N) n = a.N - 1;
        srcW[h] = W + (static_cast<int64_t>(n) * a.w_n) * EB + chunk * 16;
      }
      w_step = KT_BYTES;
    } else if constexpr (EB == 2) {
#endif
      // [k/8][n/16][8 k][16 n] image (256-byte blocks): wave w fills k-block w with two glds (n-blocks 0-3,
      // 4-7); lane l -> n-block l/16, stored row (l%16)/2, columns (l%2)*8..+8; odd k-blocks hold rows 4-7 first
      const int rr = ((lane & 15) >> 1) ^ ((wave & 1) ? 4 : 0);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int n = n0 + h * hoff + (lane >> 4) * 16 + (lane & 1) * 8;
        const int n_a = n > a.N - 8 ? a.N - 8 : n;                     // partial n-tile: stay inside the row
        const int n_b = n + 64 > a.N - 8 ? a.N - 8 : n + 64;
        srcW[h] = W + (static_cast<int64_t>(wave * 8 + rr) * a.w_k + n_a) * 2;
        w2_off[h] = (n_b - n_a) * 2;
      }
      w_step = static_cast<int64_t>(BK) * a.w_k * 2;
    } else {
      // 1-byte elements: [k/8][n/16][8 k][16 n] image (128-byte blocks), block (kb, nb) stored at
      // kb*8 + (nb ^ ((kb>>1)&1)); wave w fills k-blocks 2w and 2w+1 (one glds each = 8 k-rows x 128 n)
      const int rr = lane & 7;
      const int nb = (lane >> 3) ^ (wave & 1);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int n = n0 + h * hoff + nb * 16;
        if (n > a.N - 16) n = a.N - 16;
        srcW[h] = W + static_cast<int64_t>(wave * 16 + rr) * a.w_k + n;
        w2_off[h] = static_cast<int>(8 * a.w_k);                      // k-block 2w+1: eight rows further down
      }
      w_step = static_cast<int64_t>(BK) * a.w_k;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) srcW[h] += static_cast<int64_t>(kt0) * w_step;
  };
  locate(bid);
  float pre_scale = 0.f;                            // (EpilogueDequant: see kLdsScales)
  if constexpr (Epi::kLdsScales && !PERSIST) {
    if (a.stage_rows && epi.lds_values()) pre_scale = epi.scale_for_thread(threadIdx.x, m0, m_end, n0, a.N);
  }

  // stage half-tile `which` (0:A0 1:A1 2:W0 3:W1) of K-tile kt (relative to this slice) into buffer buf
  auto stage = [&](int which, int kt, int buf) {
    if (kt >= nkt) kt = nkt - 1;                    // keep the vmcnt bookkeeping uniform at the tail
    lds_char* dst = smem + buf * KTILE_BYTES + which * HALF_BYTES + wave * 2048;
    if (which < 2) {
      const char* p = srcA[which] + static_cast<int64_t>(kt) * KT_BYTES;
      glds16(p, dst);
      glds16(p + a2_off[which], dst + 1024);
    } else {
      const int h = which - 2;
      const char* p = srcW[h] + static_cast<int64_t>(kt) * w_step;
      glds16(p, dst);
      glds16(p + w2_off[h], dst + 1024);
    }
  };

  // ---- fragment read offsets ------------------------------------------------------------------------------
  // K-major: sub-tile (rb, ks) at (rb*2+ks)*1024; lane reads row l&15, chunk (l>>4) ^ (2 if row >= 8)
#if GEMM256_LINE_PIECES
  // lane reads row l & 15, chunk ks * 4 + (l >> 4), stored in slot chunk ^ (row / 2 % 8): conflict-free in each of ds_read_b128's
  // four 16-lane groups
  const int kmaj_lane = (lane & 15) * 128 + (((lane >> 4) ^ (((lane & 15) >> 1) & 7)) * 16);
  const int kmaj_lane1 = (lane & 15) * 128 + (((4 + (lane >> 4)) ^ (((lane & 15) >> 1) & 7)) * 16);
#else
  const int kmaj_lane = (lane & 15) * 64 + (((lane >> 4) ^ ((lane & 8) ? 2 : 0)) * 16);
  const int kmaj_lane1 = kmaj_lane + 1024;
#endif
  const int grp = lane >> 4;

  typedef const __attribute__((address_space(3))) frag16* lds_frag_ptr;
  auto read_a = [&](frag32 (&fa)[4], int h, int buf) {
    const lds_char* base = smem + buf * KTILE_BYTES + h * HALF_BYTES + (wm * 4) * 2048;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      fa[i] = join(*reinterpret_cast<lds_frag_ptr>(base + i * 2048 + kmaj_lane), *reinterpret_cast<lds_frag_ptr>(base + i * 2048 + kmaj_lane1));
  };
  auto read_w = [&](frag32 (&fw)[2], int h, int buf) {              // K-major W ([N,K])
    const lds_char* base = smem + buf * KTILE_BYTES + (2 + h) * HALF_BYTES + (wn * 2) * 2048;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      fw[j] = join(*reinterpret_cast<lds_frag_ptr>(base + j * 2048 + kmaj_lane), *reinterpret_cast<lds_frag_ptr>(base + j * 2048 + kmaj_lane1));
  };
  // N-major W ([K,N]): transposed reads.  hipcc drains vmcnt(0) in front of the ds_read_tr builtins (it
  // cannot prove the read independent of the LDS-DMA writes in flight), which serialises the pipeline;
  // so the reads are issued from inline asm and retired by an explicit lgkmcnt wait that names every
  // destination register (the compiler may not touch them in between).  Outputs are EARLY-CLOBBER: a destination
  // that shares a register with an address operand is overwritten (asynchronously) while later reads of the same
  // statement still need the address — observed as wrong tiles in the tail K-tile of odd K-tile counts.
  struct TrRegs { i32x2 r[8]; };                                   // [j][ks][first | second 8 bytes of K]
  const unsigned smem_u32 = static_cast<unsigned>(reinterpret_cast<size_t>(smem));
  // EB == 2: block (kb, nb) at (kb*8+nb)*256, kb = ks*4 + grp; lane 4q+p -> stored row q (+4), cols 4p
  // EB == 1: block (kb, nb) at (kb*8 + (nb ^ ((kb>>1)&1)))*128, kb = ks*8 + 2*grp + half; lane 2q+p -> row q, cols 8p
  unsigned tr_lane[2];
  if constexpr (EB == 2) {
    const int qq = (lane & 15) >> 2, pp = lane & 3;
    tr_lane[0] = grp * 2048 + (qq + ((grp & 1) ? 4 : 0)) * 32 + pp * 8;     // k rows 0-3 of the block
    tr_lane[1] = grp * 2048 + (qq + ((grp & 1) ? 0 : 4)) * 32 + pp * 8;     // k rows 4-7
  } else {
    const int qq = (lane & 15) >> 1, pp = lane & 1;
    tr_lane[0] = grp * 2048 + ((0 ^ (grp & 1)) * 128) + qq * 16 + pp * 8;   // j = 0
    tr_lane[1] = grp * 2048 + ((1 ^ (grp & 1)) * 128) + qq * 16 + pp * 8;   // j = 1
  }
  auto issue_w_tr = [&](TrRegs& t, int h, int buf) {
    const unsigned base = smem_u32 + buf * KTILE_BYTES + (2 + h) * HALF_BYTES + (wn * 2) * (EB == 2 ? 256 : 128);
    const unsigned a0 = base + tr_lane[0], a1 = base + tr_lane[1];
    if constexpr (EB == 2) {
      asm volatile(
          "ds_read_b64_tr_b16 %0, %8\n\t"
          "ds_read_b64_tr_b16 %1, %9\n\t"
          "ds_read_b64_tr_b16 %2, %8 offset:8192\n\t"
          "ds_read_b64_tr_b16 %3, %9 offset:8192\n\t"
          "ds_read_b64_tr_b16 %4, %8 offset:256\n\t"
          "ds_read_b64_tr_b16 %5, %9 offset:256\n\t"
          "ds_read_b64_tr_b16 %6, %8 offset:8448\n\t"
          "ds_read_b64_tr_b16 %7, %9 offset:8448"
          : "=&v"(t.r[0]), "=&v"(t.r[1]), "=&v"(t.r[2]), "=&v"(t.r[3]), "=&v"(t.r[4]), "=&v"(t.r[5]), "=&v"(t.r[6]), "=&v"(t.r[7])
          : "v"(a0), "v"(a1)
          : "memory");
    } else {
      asm volatile(
          "ds_read_b64_tr_b8 %0, %8\n\t"
          "ds_read_b64_tr_b8 %1, %8 offset:1024\n\t"
          "ds_read_b64_tr_b8 %2, %8 offset:8192\n\t"
          "ds_read_b64_tr_b8 %3, %8 offset:9216\n\t"
          "ds_read_b64_tr_b8 %4, %9\n\t"
          "ds_read_b64_tr_b8 %5, %9 offset:1024\n\t"
          "ds_read_b64_tr_b8 %6, %9 offset:8192\n\t"
          "ds_read_b64_tr_b8 %7, %9 offset:9216"
          : "=&v"(t.r[0]), "=&v"(t.r[1]), "=&v"(t.r[2]), "=&v"(t.r[3]), "=&v"(t.r[4]), "=&v"(t.r[5]), "=&v"(t.r[6]), "=&v"(t.r[7])
          : "v"(a0), "v"(a1)
          : "memory");
    }
  };
  auto retire_w_tr = [&](TrRegs& t, frag32 (&fw)[2]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(t.r[0]), "+v"(t.r[1]), "+v"(t.r[2]), "+v"(t.r[3]), "+v"(t.r[4]), "+v"(t.r[5]), "+v"(t.r[6]), "+v"(t.r[7])
                 :
                 : "memory");
#pragma unroll
    for (int j = 0; j < 2; ++j)
      fw[j] = frag32{t.r[j * 4][0], t.r[j * 4][1], t.r[j * 4 + 1][0], t.r[j * 4 + 1][1],
                     t.r[j * 4 + 2][0], t.r[j * 4 + 2][1], t.r[j * 4 + 3][0], t.r[j * 4 + 3][1]};
  };

  // acc[mt][nt]: mt = h_m*4 + i (16-row tiles of this wave), nt = h_n*2 + j (16-col tiles)
  acc_t acc[8][4];

  auto quadrant = [&](const frag32 (&fa)[4], const frag32 (&fw)[2], int hm, int hn) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < P::KS; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[hm * 4 + i][hn * 2 + j] = P::mma(fw[j], fa[i], acc[hm * 4 + i][hn * 2 + j], ks);
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- schedule -------------------------------------------------------------------------------------
  // A K-tile is 4 phases; a phase is two segments separated by barriers:
  //     R_p : issue one half-tile of a future K-tile (2 glds) + this phase's fragment reads
  //     M_p : 16 MFMAs (one 64x32 C-quadrant), then s_waitcnt vmcnt(6)
  // Waves 4-7 (wm = 1) run ONE barrier behind waves 0-3, so on every SIMD one wave is in its M
  // segment while its partner is in R: the matrix pipe and the LDS/VMEM pipes ping-pong.
  // Hazards under that stagger (a lagging reader / stager is one segment late):
  //   WAR: a half-tile is restaged >= 2 phases after the phase that last ds_read it;
  //   RAW: the counted wait that retires a half-tile sits at the end of the phase TWO before the
  //        phase that first reads it (wait -> barrier -> barrier -> read, for either group).
  // Steady state, K-tile t in buffer b (reads: P1 A0+W0, P2 W1, P3 A1, P4 none - W0 stays in VGPRs):
  //     P1 stages W1(t+1)->b^1   P2 stages A1(t+1)->b^1   P3 stages A0(t+2)->b   P4 stages W0(t+2)->b
  // After every phase "all but the last 3 half-tiles issued" have landed, which is exactly what the
  // read two phases later needs (GroupGemm schedule table).
  auto prologue = [&]() {
    stage(0, 0, 0); stage(1, 0, 0); stage(2, 0, 0); stage(3, 0, 0);
    stage(0, 1, 1); stage(2, 1, 1);
  };
  prologue();

  frag32 fa0[4], fa1[4], fw0[2], fw1[2];          // A0 / A1 fragments live in their own registers: A0 of the NEXT K-tile is read in P4
  TrRegs tr;

  auto seg_end = [&]() {
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
#ifdef GEMM_STAMPS
  unsigned tacc = 0;
  unsigned long long t_prev = __builtin_readcyclecounter();
#endif
  auto seg_end_s = [&](int p) {
    G_STAMP(5 * p + 2);                                  // MFMA segment
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    G_STAMP(5 * p + 3);                                  // counted wait
    __builtin_amdgcn_s_barrier();
    G_STAMP(5 * p + 4);                                  // second barrier
  };
  auto ktile = [&](int t, int buf) {
    // P1   (A0 of this K-tile was read in P4 of the previous one, or ahead of the loop)
    stage(3, t + 1, buf ^ 1);
    if constexpr (W_NMAJOR) issue_w_tr(tr, 0, buf); else read_w(fw0, 0, buf);
#ifdef GEMM_A0_IN_P1                                     // A/B switch: round 1's placement (12 reads in P1, none in P4)
    read_a(fa0, 0, buf);
#endif
    G_STAMP(0);                                          // stage + read issue
    __builtin_amdgcn_s_barrier();
    G_STAMP(1);                                          // first barrier
    if constexpr (W_NMAJOR) retire_w_tr(tr, fw0);
    quadrant(fa0, fw0, 0, 0);
    seg_end_s(0);
    // P2
    stage(1, t + 1, buf ^ 1);
    if constexpr (W_NMAJOR) issue_w_tr(tr, 1, buf); else read_w(fw1, 1, buf);
    G_STAMP(5);
    __builtin_amdgcn_s_barrier();
    G_STAMP(6);
    if constexpr (W_NMAJOR) retire_w_tr(tr, fw1);
    quadrant(fa0, fw1, 0, 1);
    seg_end_s(1);
    // P3
    stage(0, t + 2, buf);
    read_a(fa1, 1, buf);
    G_STAMP(10);
    __builtin_amdgcn_s_barrier();
    G_STAMP(11);
    quadrant(fa1, fw1, 1, 1);
    seg_end_s(2);
    // P4   reads A0 of K-tile t+1 (retired by the wait at the end of P2; restaged two phases after this read at the earliest)
    stage(2, t + 2, buf);
#ifndef GEMM_A0_IN_P1
    read_a(fa0, 0, buf ^ 1);
#endif
    G_STAMP(15);
    __builtin_amdgcn_s_barrier();
    G_STAMP(16);
    quadrant(fa1, fw0, 1, 0);
    seg_end_s(3);
  };

  // stores the previous tile's epilogue put behind this tile's prologue loads (vmcnt counts both, in issue order)
  bool stores_behind_prologue = false;
  bool first_tile = true;
  for (;;) {
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = acc_t{0, 0, 0, 0};
  if (PERSIST && !first_tile) {
    if (stores_behind_prologue) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");   // 4 + PERSIST_EPI_STORES: K-tile 0 has landed
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            // a partial tile's stores were not counted
  } else {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");    // K-tile 0 has landed
  }
  first_tile = false;
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();            // the stagger
#ifndef GEMM_A0_IN_P1
  read_a(fa0, 0, 0);                                    // A0 of K-tile 0 (every later A0 is read one phase ahead, in P4)
#endif

  int t = 0;
  for (; t + 1 < nkt; t += 2) {
    ktile(t, 0);
    ktile(t + 1, 1);
  }
  if (t < nkt) ktile(t, 0);
  if (wm == 0) __builtin_amdgcn_s_barrier();            // pair the stagger barrier
  // (the nops: an MFMA issued from asm is not padded by hipcc in front of the first vector read of its result)
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");

#ifdef GEMM_STAMPS
  if (bid < 8192) {
    if (lane < 20) g_gemm_stamps[(bid * 8 + wave) * 32 + lane] = tacc;
    if (lane == 31) g_gemm_stamps[(bid * 8 + wave) * 32 + 31] = static_cast<unsigned>(nkt);
    if (lane == 30) g_gemm_stamps[(bid * 8 + wave) * 32 + 30] = static_cast<unsigned>(__builtin_amdgcn_s_memrealtime());
  }
#endif
  // Wait states between the K loop's last MFMA and the first instruction that reads an accumulator.  hipcc places them for the
  // MFMAs it emits itself; the fp8 policy issues its MFMA from inline asm, which the hazard recogniser cannot see into, and a
  // first read that follows a branch can go unprotected even for emitted ones (found in the MLA latent kernels).
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
  if constexpr (PERSIST) {
    // ---- persistent form: request the next tile, then write this one out through the wave's own staging area ----------
    static_assert(Epi::kRowStaged, "the persistent form has the row-staged epilogue only");
    const int e_m0 = m0, e_mend = m_end, e_n0 = n0;
    const int next = bid + static_cast<int>(gridDim.x);
    const bool has_next = next < total;
    if (has_next) {
      locate(next);
      prologue();
    }
    typedef typename Epi::out_t OT;
    typedef typename vec_of<OT, 4>::type V4;
    typedef typename vec_of<OT, 8>::type V8;
    const int l15 = lane & 15, g4 = lane >> 4;
    OT* C = epi.C;
    if (e_m0 + BM <= e_mend && e_n0 + BN <= a.N) {
      // full tile: exactly PERSIST_EPI_STORES store instructions per wave, no lane conditions (the next tile's first wait counts on it)
      lds_char* reg = smem + LDS_BYTES + wave * 4096;
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
#pragma unroll
        for (int mh = 0; mh < 2; ++mh) {
          const int mt = pass * 2 + mh;
          const int row = mh * 16 + l15;                 // row inside the pass's 32
          const int sw = ((row >> 1) & 3) << 2;
          epi.row_begin(e_m0 + (mt >> 2) * 128 + wm * 64 + (mt & 3) * 16 + l15);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            const V4 o = epi.to4(e_n0 + (nt >> 1) * 128 + wn * 32 + (nt & 1) * 16 + g4 * 4, acc[mt][nt]);
            const int slot = ((nt >> 1) * 8 + (nt & 1) * 4 + g4) ^ sw;
            *reinterpret_cast<__attribute__((address_space(3))) V4*>(reg + row * 128 + slot * 8) = o;
          }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int piece = it * 16 + (lane >> 2);           // (row, column half) inside the pass: 4 lanes x 16 B = 64 B
          const int row = piece >> 1, half = piece & 1, c = lane & 3;
          const int wrow = pass * 32 + row;                  // row of this wave: (wrow >> 6) picks the M half-tile
          const int m = e_m0 + (wrow >> 6) * 128 + wm * 64 + (wrow & 63);
          const int pair = (half * 4 + c) ^ (((row >> 1) & 3) << 1);
          const V8 v = *reinterpret_cast<const __attribute__((address_space(3))) V8*>(reg + row * 128 + pair * 16);
          const int mc = map_row(m, a.c_rc, a.c_ml, a.c_off, a.c_mul);
          *reinterpret_cast<V8*>(C + static_cast<int64_t>(mc) * epi.ldc + e_n0 + half * 128 + wn * 32 + c * 8) = v;
        }
      }
      stores_behind_prologue = true;
    } else {
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) {
        const int m = e_m0 + (mt >> 2) * 128 + wm * 64 + (mt & 3) * 16 + (lane & 15);
        if (m >= e_mend) continue;
        epi.row_begin(m);
        const int mc = map_row(m, a.c_rc, a.c_ml, a.c_off, a.c_mul);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const int n = e_n0 + (nt >> 1) * 128 + wn * 32 + (nt & 1) * 16 + (lane >> 4) * 4;
          if (n >= a.N) continue;
          epi.store(mc, n, a.N, acc[mt][nt]);
        }
      }
      stores_behind_prologue = false;
    }
    if (!has_next) return;
    bid = next;
    continue;
  }
  break;
  }

  // ---- epilogue ---------------------------------------------------------------------------------------
  if (a.splitk > 1) {                                  // raw accumulators of this K slice -> slab[kslice][m][n]
    acc_t* slab = static_cast<acc_t*>(a.slab);
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      const int m = m0 + (mt >> 2) * 128 + wm * 64 + (mt & 3) * 16 + (lane & 15);
      if (m >= m_end) continue;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int n = n0 + (nt >> 1) * 128 + wn * 32 + (nt & 1) * 16 + (lane >> 4) * 4;
        if (n + 4 <= a.N) slab[((static_cast<int64_t>(kslice) * a.slab_rows + m) * a.N + n) / 4] = acc[mt][nt];
      }
    }
    return;
  }
  if constexpr (EB == 2 && std::is_same<acc_t, f32x4>::value) {
    if (a.glu && a.stage_rows) {                       // row-staged (see below): 128 rows x 32 columns per wave, 64-byte pieces
      __builtin_amdgcn_s_barrier();
      lds_char* reg = smem + wave * 16384;
      typedef typename vec_of<E, 4>::type V4;
      typedef typename vec_of<E, 8>::type V8;
      const int l15 = lane & 15, g4 = lane >> 4;
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) {
        const int row = (mt >> 2) * 64 + (mt & 3) * 16 + l15;
        const int sw = ((row >> 1) & 3) << 2;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          V4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float gf = elt<E>::to_f(elt<E>::from_f(acc[mt][nt][e]));
            const float uf = elt<E>::to_f(elt<E>::from_f(acc[mt][nt + 2][e]));
            o[e] = elt<E>::from_f(elt<E>::to_f(elt<E>::from_f(silu_f(gf))) * uf);
          }
          *reinterpret_cast<__attribute__((address_space(3))) V4*>(reg + row * 128 + ((nt * 4 + g4) ^ sw) * 8) = o;
        }
      }
      E* C = static_cast<E*>(a.C);
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = it * 16 + (lane >> 2), c = lane & 3;
        const int m = m0 + (row >> 6) * 128 + wm * 64 + (row & 63);
        if (m >= m_end) continue;
        const V8 v = *reinterpret_cast<const __attribute__((address_space(3))) V8*>(reg + row * 128 + (c ^ (((row >> 1) & 3) << 1)) * 16);
        const int mc = map_row(m, a.c_rc, a.c_ml, a.c_off, a.c_mul);
        *reinterpret_cast<V8*>(C + static_cast<int64_t>(mc) * a.ldc + n0 + wn * 32 + c * 8) = v;
      }
      return;
    }
    if (a.glu) {                                       // accumulator tiles nt and nt + 2 hold gate and up of the same columns
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) {
        const int m = m0 + (mt >> 2) * 128 + wm * 64 + (mt & 3) * 16 + (lane & 15);
        if (m >= m_end) continue;
        const int mc = map_row(m, a.c_rc, a.c_ml, a.c_off, a.c_mul);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const int n = n0 + wn * 32 + nt * 16 + (lane >> 4) * 4;
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float gf = elt<E>::to_f(elt<E>::from_f(acc[mt][nt][e]));
            const float uf = elt<E>::to_f(elt<E>::from_f(acc[mt][nt + 2][e]));
            o[e] = elt<E>::to_f(elt<E>::from_f(silu_f(gf))) * uf;
          }
          epi.store(mc, n, a.N / 2, o);
        }
      }
      return;
    }
  }
  if constexpr (Epi::kRowStaged) {
    // Row-staged stores.  The accumulators hold a row as 8-byte pieces (4 columns per lane), 64 store instructions per
    // wave touching 64 lines each; the epilogue is fully exposed (one workgroup per CU: 6 % of the kernel at K = 4096).
    // Each wave instead transposes its 128 rows x (2 x 32 columns) through a private 16 KiB LDS region (the tile buffers
    // are free behind the barrier) and stores 64-byte row pieces, 16 bytes per lane: 16 instructions.  LDS image: row
    // stride 128 B, 8-byte slot s of row r at s ^ (((r >> 1) & 3) << 2) (2-way = minimal conflicts for the writes,
    // 16-byte pairs stay together for the reads).
    bool lds_vals = false;                                 // the epilogue's per-row / per-column values come from LDS (uniform)
    if constexpr (Epi::kLdsScales && !PERSIST) lds_vals = epi.lds_values();
    if (a.stage_rows && n0 + BN <= a.N && (!epi.has_bias() || lds_vals)) {
      typedef __attribute__((address_space(3))) float lds_float;
      lds_float* scales = reinterpret_cast<lds_float*>(smem + LDS_BYTES);     // [256 row values | 256 column values], kLdsScales only
      if constexpr (Epi::kLdsScales && !PERSIST) {
        if (lds_vals) scales[threadIdx.x] = pre_scale;
      }
      __builtin_amdgcn_s_barrier();                        // every wave is done with the tile buffers (and its DMA has landed)
      lds_char* reg = smem + wave * 16384;
      typedef typename Epi::out_t OT;
      typedef typename vec_of<OT, 4>::type V4;
      typedef typename vec_of<OT, 8>::type V8;
      const int l15 = lane & 15, g4 = lane >> 4;
      f32x4 col_scale_v[4];                               // (read once: behind the staging stores below the compiler would re-read them per row tile)
      if constexpr (Epi::kLdsScales && !PERSIST) {
        if (lds_vals) {
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            col_scale_v[nt] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(scales + 256 + (nt >> 1) * 128 + wn * 32 + (nt & 1) * 16 + g4 * 4);
        }
      }
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) {
        const int row = (mt >> 2) * 64 + (mt & 3) * 16 + l15;
        const int sw = ((row >> 1) & 3) << 2;
        float row_scale_v = 0.f;
        if constexpr (Epi::kLdsScales && !PERSIST) {
          if (lds_vals) row_scale_v = scales[(mt >> 2) * 128 + wm * 64 + (mt & 3) * 16 + l15];
        }
        if (!lds_vals) epi.row_begin(min(m0 + (mt >> 2) * 128 + wm * 64 + (mt & 3) * 16 + l15, m_end - 1));   // rows past the group: staged, never stored
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          V4 o;
          bool done = false;
          if constexpr (Epi::kLdsScales && !PERSIST) {
            if (lds_vals) { o = epi.to4_scaled(acc[mt][nt], row_scale_v, col_scale_v[nt]); done = true; }
          }
          if (!done) o = epi.to4(n0 + (nt >> 1) * 128 + wn * 32 + (nt & 1) * 16 + g4 * 4, acc[mt][nt]);
          const int slot = ((nt >> 1) * 8 + (nt & 1) * 4 + g4) ^ sw;
          *reinterpret_cast<__attribute__((address_space(3))) V4*>(reg + row * 128 + slot * 8) = o;
        }
      }
      OT* C = epi.C;
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int piece = it * 16 + (lane >> 2);             // (row, column half) of this wave: 4 lanes x 16 B = 64 B
        const int row = piece >> 1, half = piece & 1, c = lane & 3;
        const int m = m0 + (row >> 6) * 128 + wm * 64 + (row & 63);
        if (m >= m_end) continue;
        const int pair = (half * 4 + c) ^ (((row >> 1) & 3) << 1);
        const V8 v = *reinterpret_cast<const __attribute__((address_space(3))) V8*>(reg + row * 128 + pair * 16);
        const int mc = map_row(m, a.c_rc, a.c_ml, a.c_off, a.c_mul);
        *reinterpret_cast<V8*>(C + static_cast<int64_t>(mc) * epi.ldc + n0 + half * 128 + wn * 32 + c * 8) = v;
      }
      return;
    }
  }
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) {
    const int m = m0 + (mt >> 2) * 128 + wm * 64 + (mt & 3) * 16 + (lane & 15);
    if (m >= m_end) continue;
    if (a.ablate == 1 && m != m0) continue;
    epi.row_begin(m);
    const int mc = map_row(m, a.c_rc, a.c_ml, a.c_off, a.c_mul);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n = n0 + (nt >> 1) * 128 + wn * 32 + (nt & 1) * 16 + (lane >> 4) * 4;
      if (n >= a.N) continue;
      epi.store(mc, n, a.N, acc[mt][nt]);
    }
  }
}

// shared precondition checks (bytes along K); eb = element bytes
inline bool gemm256_layout_ok(const GemmArgs& a, int eb) {
  const int bk = KT_BYTES / eb, al = 16 / eb;
  if (a.K < bk || a.K % bk != 0 || a.N < al) return false;
  if (a.lda % al != 0) return false;
  if (!aligned_to(a.A, 16) || !aligned_to(a.W, 16)) return false;
  if (a.w_n == 1) {                                   // [K,N]
    if (a.w_k % al != 0 || a.w_group % al != 0 || a.N % al != 0) return false;
  } else if (a.w_k == 1) {                            // [N,K]
    if (a.w_n % al != 0 || a.w_group % al != 0) return false;
  } else {
    return false;
  }
  return true;
}

inline int device_cu_count() {
  static std::atomic<int> cached[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  int v = cached[dev & 63].load(std::memory_order_relaxed);
  if (v == 0) {
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    cached[dev & 63].store(v, std::memory_order_relaxed);
  }
  return v;
}

template <typename P, typename Epi, bool ALLOW_PERSISTENT = false>
inline int gemm256_launch(const GemmArgs& a_in, const Epi& epi, int64_t m_total, hipStream_t s) {
  GemmArgs a = a_in;
  const int64_t n_tiles = a.glu ? (a.N / 2) / 128 : ceil_div(a.N, BN);
  const int64_t blocks = (ceil_div(m_total, BM) + a.G) * n_tiles * a.splitk;   // upper bound; surplus blocks exit
  // Start stagger (see the kernel): ONE-group products of at most 16 K-tiles with at least eight rounds of workgroups — the
  // MLA decompression GEMM against a long cache.  Measured on one MI355X (10-ns ticks per phase 0 / 60 / 120 / 180 / 240 / 320):
  //  * the product alone, back to back (scripts/probes/gemm_stagger_ab.py, profiles/r4_gemm_stagger_ab.txt): [2048, 512] x
  //    [32768, 512]^T 768 / 821 / 858 / 867 / 802 / 707 TF, M = 10240: 757 / 769 / 766-794 / 849-865 / 850-881 / 805-815, K = 1024:
  //    1048 / 1089 / 1108 / 1088 / 1060 / 1012, K = 4096: 1414 / 1412 / 1394 / 1382 (long-K rounds drift apart by themselves);
  //  * inside MojoPagedPrefillMLA, where the product alternates with the attention kernel (mla_prefill_stagger_ab.py, A/B in
  //    one process; kernel traces by scripts/probes/prof_stagger.sh): 4 x 512 + 2048 cached 929 -> 891 us eager, 938 -> 872
  //    under graph replay (GEMM 378 -> 370 us, attention 526 -> 500 us); 4 x 512 without a cache (4 rounds) 203 -> 202 us with
  //    the GEMM itself 88 -> 93 us; grouped products with a tile per group (the absorbed route's projections) 104 -> 112 us.
  //    Hence the restriction to one group and >= 8 rounds.  MOJO_HIP_GEMM_STAGGER=<ticks> forces (0 = off).
  if (const long long e = MOJO_SWITCH("MOJO_HIP_GEMM_STAGGER", -1); e >= 0) {
    a.stagger_ticks = static_cast<int>(e);
  } else if (a.G == 1 && a.K / (KT_BYTES / P::EB) <= 16 && blocks >= 8 * static_cast<int64_t>(device_cu_count())) {
    a.stagger_ticks = 150;
  }
  a.stagger_blocks = device_cu_count();
  MOJO_REQUIRE(blocks < (1LL << 31), MOJO_EUNSUPPORTED, "gemm: grid too large");
  if constexpr (Epi::kRowStaged && ALLOW_PERSISTENT) {
    // persistent form (one workgroup per CU, next tile requested before this tile's epilogue): row-staged 16-bit output,
    // no bias / GLU / split-K, and enough tiles that every CU gets at least two.  OPT-IN (MOJO_HIP_GEMM_PERSIST=1): A/B on
    // one MI355X in one session (round 2, random bf16, TFLOP/s plain -> persistent): Mixtral up [K,N] 1279 -> 1278-1285,
    // [N,K] 1313-1318 -> 1315-1328, down 1226-1234 -> 1219-1222, skewed 1199-1200 -> 1196-1203, M = 4096 1010-1016 ->
    // 1001-1008, reference case 1050-1068 -> 1070-1094, K = 512 (MLA decompression) 741-755 -> 748-804: inside the run-to-run
    // spread except at short K.  The hardware's own workgroup hand-over already hides most of what the loop was meant to
    // hide, and hipcc builds the loop body with 19-20 SGPR spills; results are identical (all GEMM tests pass with it on).
    // ROUND 5: ON for short-K one-group products (at most 16 K-tiles: the MLA decompression GEMM, K = 512 / 1024), where a
    // tile spends a third of its time outside the K loop.  Measured on the final binary (scripts/probes/gemm_shortk_bound.py,
    // profiles/r5_gemm_shortk_bound.txt; bf16 [N,K], one group): 2048 x 512 x 32768 80.9 -> 73.9 us (849 -> 930 TF), 10240 x 512 x 32768
    // 388.2 (with the start stagger; 411.4 without) -> 371.7 us, 2048 x 1024 x 32768 121.3 -> 114.6 us; K = 4096 373.1 -> 369.3 us
    // (inside the spread: left on the classic launch).  The same probe bounds what ANY scheme that hides the epilogue can
    // reach: with the C stores removed (timing only) the three short-K products take 58.1 / 261.5 / 103.1 us.
    bool off = !(a.G == 1 && a.K / (KT_BYTES / P::EB) <= 16);
#ifdef MOJO_HIP_BUILD_EXPERIMENTS
    if (const long long force = MOJO_SWITCH("MOJO_HIP_GEMM_PERSIST", -1); force >= 0) off = force != 1;
#endif
    const int cus = device_cu_count();
    if (!off && a.stage_rows && !a.glu && a.splitk == 1 && !epi.has_bias() && !a.ablate && blocks >= 2 * cus) {
      constexpr int LDS_P = LDS_BYTES + PERSIST_STAGE_BYTES;
      if (a.w_n == 1) {
        auto* fn = gemm256_kernel<P, true, Epi, true>;
        static std::atomic<uint64_t> attr_set{0};
        if (first_call_on_device(attr_set)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_P);
        hipLaunchKernelGGL(fn, dim3(static_cast<unsigned>(cus)), dim3(512), LDS_P, s, a, epi);
      } else {
        auto* fn = gemm256_kernel<P, false, Epi, true>;
        static std::atomic<uint64_t> attr_set{0};
        if (first_call_on_device(attr_set)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_P);
        hipLaunchKernelGGL(fn, dim3(static_cast<unsigned>(cus)), dim3(512), LDS_P, s, a, epi);
      }
      MOJO_CHECK_LAUNCH("gemm256(persistent)");
      note_launch("gemm256:persistent");
      return MOJO_OK;
    }
  }
  constexpr int LDS_K = LDS_BYTES + (Epi::kLdsScales ? 2048 : 0);      // + the epilogue's scales
  if (a.w_n == 1) {
    auto* fn = gemm256_kernel<P, true, Epi>;
    static std::atomic<uint64_t> attr_set{0};
    if (first_call_on_device(attr_set)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_K);
    hipLaunchKernelGGL(fn, dim3(static_cast<unsigned>(blocks)), dim3(512), LDS_K, s, a, epi);
  } else {
    auto* fn = gemm256_kernel<P, false, Epi>;
    static std::atomic<uint64_t> attr_set{0};
    if (first_call_on_device(attr_set)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_K);
    hipLaunchKernelGGL(fn, dim3(static_cast<unsigned>(blocks)), dim3(512), LDS_K, s, a, epi);
  }
  MOJO_CHECK_LAUNCH("gemm256");
  note_launch("gemm256:%s:%s%s%s%s", a.stage_rows ? "staged" : "direct", a.w_n == 1 ? "KN" : "NK", a.splitk > 1 ? ":splitk" : "",
              a.stagger_ticks ? ":stagger" : "", a.glu ? ":glu" : "");
  return MOJO_OK;
}

}  // namespace g256
}  // namespace mojo
