// MojoQuantGemm: out[M,N] = round_out( (A_q @ W_q) * input_scale[m] * weight_scale[n] ).
//   int8 : exact int32 accumulation on v_mfma_i32_16x16x64_i8 (the golden's fp32 sum of int products is
//          exact up to 2^24), fp32 scaling in the golden's order, one rounding.
//   fp8  : OCP e4m3 on v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales (2x the bf16 MFMA rate);
//          this dtype has no reference implementation — parity unpinned (checked against oracle/ only).
// Shapes outside the MFMA kernel's preconditions run on a plain one-output-per-thread kernel.
//
// Algorithmic FLOPs: 2*M*K*N (MFMA-bound for M >= 128); bytes: K*N (weight stream) + M*K + M*N*out.
#include <stdlib.h>

#include <type_traits>

#include "gemm256_core.h"
#include "gemm_tile128_core.h"
#ifdef MOJO_HIP_BUILD_EXPERIMENTS        // in-launch split-K combine, measured slower (DESIGN Appendix A #9): opt-in build only
#include "experiments/splitk_combine.h"
#endif

namespace mojo {

__device__ __forceinline__ float fp8_to_f(uint8_t v) { return __builtin_amdgcn_cvt_f32_fp8(static_cast<int>(v), 0); }

template <typename TO, bool FP8>
__global__ __launch_bounds__(256) void quant_gemm_generic_kernel(const uint8_t* __restrict__ A, const uint8_t* __restrict__ W,
                                                                 const float* __restrict__ rs, const bf16_t* __restrict__ cs,
                                                                 TO* __restrict__ C, int64_t M, int K, int N, int64_t w_k,
                                                                 int64_t w_n) {
  const int64_t total = M * N;
  for (int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
       idx += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t m = idx / N;
    const int n = static_cast<int>(idx - m * N);
    const uint8_t* a = A + m * K;
    const uint8_t* w = W + static_cast<int64_t>(n) * w_n;
    float v;
    if constexpr (FP8) {
      float acc = 0.f;
      for (int k = 0; k < K; ++k) acc = fmaf(fp8_to_f(a[k]), fp8_to_f(w[k * w_k]), acc);
      v = acc;
    } else {
      int acc = 0;
      for (int k = 0; k < K; ++k) acc += static_cast<int>(static_cast<int8_t>(a[k])) * static_cast<int>(static_cast<int8_t>(w[k * w_k]));
      v = static_cast<float>(acc);
    }
    v = __fmul_rn(__fmul_rn(v, rs[m]), static_cast<float>(cs[n]));
    asm volatile("" : "+v"(v));                      // no mul+narrow fusion: the golden rounds twice
    C[m * N + n] = elt<TO>::from_f(v);
  }
}

// split-K finalize: out[m][n] = round_TO( (sum_s slab[s][m][n]) * rs[m] * cs[n] ), slices summed in index order
template <typename TO, typename ACC>
__global__ __launch_bounds__(256) void quant_finalize_kernel(const ACC* __restrict__ slab, int splitk, int64_t M, int N,
                                                             const float* __restrict__ rs, const bf16_t* __restrict__ cs,
                                                             TO* __restrict__ C) {
  const int64_t total = M * N;
  if (N % 4 == 0 && (reinterpret_cast<uintptr_t>(C) & (4 * sizeof(TO) - 1)) == 0) {      // 16 bytes of every slab per lane
    typedef typename vec_of<ACC, 4>::type A4;
    typedef typename vec_of<TO, 4>::type O4;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total / 4; i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
      A4 acc = {0, 0, 0, 0};
      for (int sidx = 0; sidx < splitk; ++sidx) acc += reinterpret_cast<const A4*>(slab + static_cast<int64_t>(sidx) * total)[i];
      const int64_t m = (i * 4) / N;
      const int n = static_cast<int>(i * 4 - m * N);
      const float r = rs[m];
      O4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = __fmul_rn(__fmul_rn(static_cast<float>(acc[e]), r), static_cast<float>(cs[n + e]));
        asm volatile("" : "+v"(v));
        o[e] = elt<TO>::from_f(v);
      }
      reinterpret_cast<O4*>(C)[i] = o;
    }
    return;
  }
  for (int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
       idx += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    ACC acc = 0;
    for (int sidx = 0; sidx < splitk; ++sidx) acc += slab[static_cast<int64_t>(sidx) * total + idx];
    const int64_t m = idx / N;
    const int n = static_cast<int>(idx - m * N);
    float v = __fmul_rn(__fmul_rn(static_cast<float>(acc), rs[m]), static_cast<float>(cs[n]));
    asm volatile("" : "+v"(v));
    C[idx] = elt<TO>::from_f(v);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Decode-sized M (<= 128) with [N,K] weights: the op is a weight STREAM (K*N bytes from HBM against 2*M*K*N cheap
// integer ops), so the 256x256 tile shape is the wrong tool — it pads M to 256 and needs 16-way split-K with fp32 slabs
// larger than the weight itself.  Skinny kernel: one workgroup = 64 output columns x all M rows x one K slice.
//   * each wave owns 16 columns and loads their weight rows row-contiguously (4 rows x 256 B per instruction), then
//     restores the MFMA fragment shape (lane = column) through a wave-private LDS image.  MFMA does not care which k a
//     lane supplies as long as both operands agree, so k-step s of a 256-byte K block is "bytes 64s + 16g .. +15 for
//     lane group g" and the activation fragments are read from LDS with the same permutation;
//   * the activation block [16*MT rows][256 B] is staged through registers into a double-buffered, padded LDS image
//     shared by the four waves (stride 272 B: conflict-free ds_read_b128);
//   * the weights of the next three K blocks and the activation blocks of the next two are in flight while a block is
//     multiplied;
//   * grid = (N/64) x splitk with ~256 workgroups (each keeps 48 KiB of weights in flight); split-K slices write raw
//     accumulators to their own slab, summed in slice order by quant_finalize_kernel (deterministic).
// Algorithmic bytes: K*N (+ M*K + 4*M*N*splitk*2 of slab traffic when split).
// NW = waves per workgroup (16 columns each; round 4): chosen with the K split so that the wave units divide evenly over the CUs
// (quant_skinny_plan) — the stream is per-CU bound, and N = 7168 in 64-column workgroups is 112 tiles x 2 slices = 224
// workgroups on 256 CUs.  A wave past the last column tile streams the last one again and stores nothing.
template <typename TO, bool FP8, int MT, bool NT /* weights read once: non-temporal loads */, int NW = 4>
__global__ __launch_bounds__(NW * 64) void quant_skinny_kernel(const uint8_t* __restrict__ A, const uint8_t* __restrict__ W,
                                                           const float* __restrict__ rs, const bf16_t* __restrict__ cs,
                                                           TO* __restrict__ C, void* __restrict__ slab, int M, int K, int N,
                                                           int splitk, int sk_slot) {
  typedef typename std::conditional<FP8, f32x4, i32x4>::type acc_t;
  constexpr int ROW = 272;                                   // padded LDS row of a 256-byte K block
  constexpr int NTH = NW * 64;                               // threads
  constexpr int AP = (MT * 256 + NTH - 1) / NTH;             // activation chunks per thread and K block
  constexpr bool A_EVEN = (MT * 256) % NTH == 0;
  __shared__ __attribute__((aligned(16))) uint8_t s_a[2][MT * 16 * ROW];
  __shared__ __attribute__((aligned(16))) uint8_t s_w[NW][2][16 * ROW];       // per wave: its 16 weight rows of a K block
  __shared__ int s_last;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int unit_raw = static_cast<int>(blockIdx.x) * NW + wave;
  const bool unit_live = unit_raw < N / 16;
  const int n0 = (unit_live ? unit_raw : N / 16 - 1) * 16;
  const int slice = blockIdx.y;
  const int m0 = blockIdx.z * (MT * 16);                     // row block (grid.z > 1: more than MT * 16 rows)
  const int nkb = K / 256;
  const int kb0 = static_cast<int>(static_cast<int64_t>(nkb) * slice / splitk);
  const int kb1 = static_cast<int>(static_cast<int64_t>(nkb) * (slice + 1) / splitk);
  // weight loads are row-contiguous: instruction j covers rows 4j .. 4j+3 of the wave's 16, 16 lanes x 16 B = one row's
  // 256-byte K block (measured: the M <= 4 GEMV, which reads whole rows, streams at twice the rate of 64-byte runs);
  // the MFMA fragment shape (lane = column) is restored by a pass through a wave-private LDS image.
  const uint8_t* wrow = W + static_cast<int64_t>(n0 + (lane >> 4)) * K + (lane & 15) * 16;

  acc_t acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = acc_t{0, 0, 0, 0};

  // K blocks are visited in an order rotated by the column tile: all column tiles of a slice read the SAME activation
  // lines, and in lock-step they would queue on one L2 channel (measured: 4 us per K block at M = 32).
  const int nb = kb1 - kb0;
  const int rot = nb > 0 ? static_cast<int>(blockIdx.x % nb) : 0;
  auto block_at = [&](int i) { const int j = i + rot; return kb0 + (j >= nb ? j - nb : j); };
  constexpr int DEPTH = 3;                                   // weight blocks in flight ahead of the multiply
  u32x4 wreg[DEPTH + 1][4], areg[2][AP];                     // activations: two blocks ahead in registers, one in LDS
  auto load_w = [&](int i, u32x4 (&wr)[4]) {
    const uint8_t* wp = wrow + static_cast<int64_t>(block_at(i)) * 256;
#pragma unroll
    for (int sx = 0; sx < 4; ++sx) {
      const u32x4* src = reinterpret_cast<const u32x4*>(wp + static_cast<int64_t>(sx) * 4 * K);
      if constexpr (NT) wr[sx] = __builtin_nontemporal_load(src); else wr[sx] = *src;
    }
  };
  auto store_w = [&](int buf, const u32x4 (&wr)[4]) {
#pragma unroll
    for (int sx = 0; sx < 4; ++sx)
      *reinterpret_cast<u32x4*>(&s_w[wave][buf][(4 * sx + (lane >> 4)) * ROW + (lane & 15) * 16]) = wr[sx];
  };
  auto load_a = [&](int i, u32x4 (&ar)[AP]) {
    const int64_t k0 = static_cast<int64_t>(block_at(i)) * 256;
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      const int idx = min(static_cast<int>(threadIdx.x) + NTH * p, MT * 256 - 1);   // 16-byte chunk of the activation block
      const int row = min(m0 + (idx >> 4), M - 1);
      ar[p] = *reinterpret_cast<const u32x4*>(A + static_cast<int64_t>(row) * K + k0 + (idx & 15) * 16);
    }
  };
  auto store_a = [&](int buf, const u32x4 (&ar)[AP]) {
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      const int idx = threadIdx.x + NTH * p;
      if (A_EVEN || idx < MT * 256) *reinterpret_cast<u32x4*>(&s_a[buf][(idx >> 4) * ROW + (idx & 15) * 16]) = ar[p];
    }
  };
  if (nb > 0) {
    load_a(0, areg[0]);
    if (nb > 1) load_a(1, areg[1]);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
      if (d < nb) load_w(d, wreg[d]);
    store_a(0, areg[0]);
    store_w(0, wreg[0]);
  }
  __syncthreads();
  // The ring index is kept compile-time by unrolling the body DEPTH + 1 times.  The steady-state rounds contain NO
  // conditional loads: with a branch around a load the compiler's wait-count analysis has to assume the load may not have
  // been issued, ranks every register of the ring as "possibly the youngest load" and waits vmcnt(0) — which silently
  // turns the three-deep prefetch into none.  The last rounds (loads that would run past the slice) go through a
  // guarded copy of the same body.
  auto body = [&](int i, auto RC, auto GUARD) {
    constexpr int r = decltype(RC)::value;
    constexpr bool guarded = decltype(GUARD)::value;
    const int cur = i & 1;                                   // LDS buffer of block i; areg[cur ^ 1] holds block i + 1
    // Issue order matters: vmcnt retires in order, so the activation load must be OLDER than the weight loads issued
    // next to it — waiting for activations at the end of the block then leaves every younger weight block in flight.
    if (!guarded || i + 2 < nb) { if (r & 1) load_a(i + 2, areg[1]); else load_a(i + 2, areg[0]); }   // slot of block i
    if (!guarded || i + DEPTH < nb) load_w(i + DEPTH, wreg[(r + DEPTH) % (DEPTH + 1)]);
#pragma unroll
    for (int sx = 0; sx < 4; ++sx) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const u32x4 af = *reinterpret_cast<const u32x4*>(&s_a[r & 1][(mt * 16 + l15) * ROW + sx * 64 + g * 16]);
        const u32x4 wf = *reinterpret_cast<const u32x4*>(&s_w[wave][r & 1][l15 * ROW + sx * 64 + g * 16]);
        if constexpr (FP8) {
          typedef long i64x2 __attribute__((ext_vector_type(2)));
          const i64x2 wl = __builtin_bit_cast(i64x2, wf), al = __builtin_bit_cast(i64x2, af);
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wl[0], al[0], acc[mt], 0, 0, 0);
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wl[1], al[1], acc[mt], 0, 0, 0);
        } else {
          acc[mt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, wf), __builtin_bit_cast(i32x4, af),
                                                          acc[mt], 0, 0, 0);
        }
      }
    }
    if (!guarded || i + 1 < nb) {
      if (r & 1) store_a(0, areg[0]); else store_a(1, areg[1]);
      store_w((r & 1) ^ 1, wreg[(r + 1) % (DEPTH + 1)]);
    }
    __syncthreads();
    (void)cur;
  };
  int i0 = 0;
  for (; i0 + 2 * DEPTH + 1 <= nb; i0 += DEPTH + 1)          // every load of the round stays inside the slice
    static_for<DEPTH + 1>([&](auto RC) { body(i0 + decltype(RC)::value, RC, std::false_type{}); });
  for (; i0 < nb; i0 += DEPTH + 1)
    static_for<DEPTH + 1>([&](auto RC) {
      constexpr int r = decltype(RC)::value;
      if (i0 + r < nb) body(i0 + r, RC, std::true_type{});
    });
  if (!unit_live) return;
  // lane holds rows m = mt*16 + l15, columns n0 + 4g .. +3
  const int n = n0 + 4 * g;
  auto emit = [&](int m, acc_t v) {
    const float r = rs[m];
    typedef typename vec_of<TO, 4>::type V4;
    V4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float x = __fmul_rn(__fmul_rn(static_cast<float>(v[q]), r), static_cast<float>(cs[n + q]));
      asm volatile("" : "+v"(x));
      o[q] = elt<TO>::from_f(x);
    }
    *reinterpret_cast<V4*>(C + static_cast<int64_t>(m) * N + n) = o;
  };
#ifdef MOJO_HIP_BUILD_EXPERIMENTS
  if (splitk > 1 && sk_slot >= 0) {
    // raw accumulators of this K slice, write-through; the last slice of the tile to arrive sums all of them in slice order
    // (splitk_combine.h) and applies the scales: one launch instead of two
    const long long slice_bytes = static_cast<long long>(M) * N * 4;
    const sk_rsrc_t rsrc = splitk_rsrc(slab, slice_bytes * splitk);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = m0 + mt * 16 + l15;
      if (m < M) splitk_store16(rsrc, slice * slice_bytes + (static_cast<long long>(m) * N + n) * 4, __builtin_bit_cast(u32x4, acc[mt]));
    }
    if (!splitk_arrive(sk_slot, static_cast<int>(blockIdx.z * gridDim.x + blockIdx.x), splitk, &s_last)) return;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = m0 + mt * 16 + l15;
      if (m >= M) continue;
      acc_t sum = acc_t{0, 0, 0, 0};
      for (int sx = 0; sx < splitk; ++sx)
        sum += __builtin_bit_cast(acc_t, splitk_load16(rsrc, sx * slice_bytes + (static_cast<long long>(m) * N + n) * 4));
      emit(m, sum);
    }
    return;
  }
#endif
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + mt * 16 + l15;
    if (m >= M) continue;
    if (splitk > 1) {
      acc_t* dst = reinterpret_cast<acc_t*>(static_cast<char*>(slab) + ((static_cast<int64_t>(slice) * M + m) * N + n) * 4);
      *dst = acc[mt];
    } else {
      emit(m, acc[mt]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// M <= 4 with [N,K] weights: a GEMV.  One wave per pair of weight rows, all 64 lanes along K (1 KiB contiguous per load
// instruction, the whole row read front to back), v_dot4_i32_i8 (or fp8 -> fp32 FMAs) against the activations held in
// LDS, one cross-lane reduction per output at the end.  No split-K, no finalize: one launch.
// Algorithmic bytes: K*N (the weight stream; the M*K activations are read once per workgroup from L2).
template <typename TO, bool FP8, int MV>
__global__ __launch_bounds__(256) void quant_gemv_kernel(const uint8_t* __restrict__ A, const uint8_t* __restrict__ W,
                                                         const float* __restrict__ rs, const bf16_t* __restrict__ cs,
                                                         TO* __restrict__ C, int M, int K, int N) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_act[];        // [MV][K], rows >= M are zero
  constexpr int RPW = 2;                                                 // weight rows per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int idx = threadIdx.x * 16; idx < MV * K; idx += 256 * 16) {
    const int m = idx / K;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (m < M) v = *reinterpret_cast<const u32x4*>(A + idx);
    *reinterpret_cast<u32x4*>(s_act + idx) = v;
  }
  __syncthreads();
  const int n0 = (blockIdx.x * 4 + wave) * RPW;
  if (n0 >= N) return;
  typedef typename std::conditional<FP8, float, int>::type acc_t;
  acc_t acc[RPW][MV];
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
    for (int m = 0; m < MV; ++m) acc[rr][m] = 0;
  const uint8_t* wp[RPW];
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) wp[rr] = W + static_cast<int64_t>(min(n0 + rr, N - 1)) * K + lane * 16;
  const int chunks = (K + 1023) / 1024;
  constexpr int AHEAD = 4;                                                // 1 KiB loads in flight per row
  u32x4 ring[AHEAD][RPW];
  auto fetch = [&](int c, u32x4 (&dst)[RPW]) {
    const bool ok = c * 1024 + lane * 16 < K;
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (ok) v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp[rr] + c * 1024));
      dst[rr] = v;
    }
  };
#pragma unroll
  for (int c = 0; c < AHEAD; ++c)
    if (c < chunks) fetch(c, ring[c]);
  for (int c0 = 0; c0 < chunks; c0 += AHEAD) {
#pragma unroll
    for (int q = 0; q < AHEAD; ++q) {
      const int c = c0 + q;
      if (c < chunks) {
        u32x4 w[RPW];
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) w[rr] = ring[q][rr];
        if (c + AHEAD < chunks) fetch(c + AHEAD, ring[q]);
        const int k0 = min(c * 1024 + lane * 16, K - 16);                 // lanes past the row read zero weights anyway
#pragma unroll
        for (int m = 0; m < MV; ++m) {
          const u32x4 av = *reinterpret_cast<const u32x4*>(s_act + m * K + k0);
#pragma unroll
          for (int rr = 0; rr < RPW; ++rr) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if constexpr (FP8) {
                const int wi = static_cast<int>(w[rr][e]), ai = static_cast<int>(av[e]);
                acc[rr][m] = fmaf(__builtin_amdgcn_cvt_f32_fp8(wi, 0), __builtin_amdgcn_cvt_f32_fp8(ai, 0), acc[rr][m]);
                acc[rr][m] = fmaf(__builtin_amdgcn_cvt_f32_fp8(wi, 1), __builtin_amdgcn_cvt_f32_fp8(ai, 1), acc[rr][m]);
                acc[rr][m] = fmaf(__builtin_amdgcn_cvt_f32_fp8(wi, 2), __builtin_amdgcn_cvt_f32_fp8(ai, 2), acc[rr][m]);
                acc[rr][m] = fmaf(__builtin_amdgcn_cvt_f32_fp8(wi, 3), __builtin_amdgcn_cvt_f32_fp8(ai, 3), acc[rr][m]);
              } else {
                acc[rr][m] = __builtin_amdgcn_sdot4(static_cast<int>(w[rr][e]), static_cast<int>(av[e]), acc[rr][m], false);
              }
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
    for (int m = 0; m < MV; ++m) {
      acc_t v = acc[rr][m];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      acc[rr][m] = v;
    }
  if (lane == 0) {
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const int n = n0 + rr;
      if (n >= N) continue;
      const float c = static_cast<float>(cs[n]);
#pragma unroll
      for (int m = 0; m < MV; ++m) {
        if (m >= M) continue;
        float v = __fmul_rn(__fmul_rn(static_cast<float>(acc[rr][m]), rs[m]), c);
        asm volatile("" : "+v"(v));
        C[static_cast<int64_t>(m) * N + n] = elt<TO>::from_f(v);
      }
    }
  }
}

static bool quant_gemv_ok(int64_t m, const GemmArgs& a) {
  return m <= 4 && a.w_k == 1 && a.w_n == a.K && a.K % 16 == 0 && a.lda == a.K && a.ldc == a.N && m * a.K <= 96 * 1024 &&
         aligned_to(a.A, 16) && aligned_to(a.W, 16);
}

template <typename TO, bool FP8>
static int launch_quant_gemv(const GemmArgs& a, const float* rs, const bf16_t* cs, int64_t m, hipStream_t s) {
  const int mv = m <= 1 ? 1 : (m <= 2 ? 2 : 4);
  const unsigned blocks = static_cast<unsigned>(ceil_div(a.N, 8));
  const size_t lds = static_cast<size_t>(mv) * a.K;
  const uint8_t* A = static_cast<const uint8_t*>(a.A);
  const uint8_t* W = static_cast<const uint8_t*>(a.W);
  TO* C = static_cast<TO*>(a.C);
#define GEMV(MV_)                                                                                                       \
  do {                                                                                                                  \
    auto* fn = quant_gemv_kernel<TO, FP8, MV_>;                                                                          \
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); \
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), lds, s, A, W, rs, cs, C, static_cast<int>(m), a.K, a.N);             \
  } while (0)
  if (mv == 1) GEMV(1); else if (mv == 2) GEMV(2); else GEMV(4);
#undef GEMV
  MOJO_CHECK_LAUNCH("quant_gemm(gemv)");
  note_launch("quant_gemv:rows%d", mv);
  return MOJO_OK;
}

static bool quant_skinny_ok(int64_t m, const GemmArgs& a) {
  return m <= 128 && a.w_k == 1 && a.w_n == a.K && a.K % 256 == 0 && a.N % 64 == 0 && a.lda == a.K && a.ldc == a.N &&
         aligned_to(a.A, 16) && aligned_to(a.W, 16) && aligned_to(a.C, 8);
}

// K split AND waves per workgroup of the decode-sized kernel.  A CU's rate does not grow with a second workgroup on it
// (measured, N = 7168: 3 slices = 336 workgroups, 1.3 per CU, 33-35 us; 2 slices = 224 and 4 = 448 workgroups 29.5 us), so the
// time goes as the wave units (16 columns x one K slice) on the fullest CU.  Round 4: four-wave workgroups alone leave e.g.
// N = 7168 at 224 or 448 workgroups (0.875 of the chip); with 4..8 waves per workgroup 64 column groups of 7 waves x 4 slices
// are exactly one per CU.  Candidates: splits of at least four K blocks per slice, at most two four-wave (one larger)
// workgroups per CU; least (units on the fullest CU) x (K share), then fewest slices (slab traffic), then fewest waves.
struct QuantSkinnyPlan { int sk, nw; double cost; };
static QuantSkinnyPlan quant_skinny_plan_for(int k, int n, int64_t m, int nw_lo, int nw_hi) {
  QuantSkinnyPlan p{1, 4, -1.0};
  const int nkb = k / 256, units = n / 16, rb = m > 64 ? 2 : 1;
  for (int sk = 1; sk <= 64 && (sk == 1 || sk <= nkb / 4); ++sk) {
    for (int nw = nw_lo; nw <= nw_hi; ++nw) {
      if (nw < 4 || nw > 8 || nw == 5) continue;
      const int64_t wgs = static_cast<int64_t>((units + nw - 1) / nw) * sk * rb;
      if (sk > 1 && wgs > (nw == 4 ? 512 : 256)) continue;       // (an unsplit launch may take several rounds of workgroups)
      const double cost = static_cast<double>((wgs + 255) / 256) * nw / sk;
      if (p.cost < 0.0 || cost < p.cost * 0.999) { p.cost = cost; p.sk = sk; p.nw = nw; }
    }
  }
  return p;
}
// The wide plans are taken where they measured faster (scripts/probes/qg_waves_ab.py, profiles/r4_qgemm_waves_ab.txt, int8, graph
// replay, cold weights, four-wave rule -> balanced plan): 64 x 7168 x 18432 40.8 -> 30.5 us (288 unsplit tiles were 1.1 rounds of
// workgroups), 128 x 18432 x 7168 51.3 -> 45.0 us (two row blocks: fewer column groups re-read the activations); NOT where the
// four-wave plan already covers 7/8 of the chip with few rows (32 x 18432 x 7168 27.3 -> 29.2 us, 16 x 4096 x 7168 10.5 -> 11.6 us:
// twice the slices for 12 % better balance).  Hence: more than 64 rows, or a balance gain of at least a quarter.
static QuantSkinnyPlan quant_skinny_plan(int k, int n, int64_t m, bool wide = true /* 6 / 7 / 8 waves instantiated (bf16 output) */) {
  if (const int v = static_cast<int>(MOJO_SWITCH("MOJO_HIP_GEMM_SPLITK", 0)); v >= 1) return QuantSkinnyPlan{v, 4, 0.0};
  if (const int w = static_cast<int>(MOJO_SWITCH("MOJO_HIP_GEMM_WAVES", 0)); w >= 4 && w <= 8)     // measurement: forced waves per workgroup (4 = the round-3 rule)
    return quant_skinny_plan_for(k, n, m, wide ? w : 4, wide ? w : 4);
  const QuantSkinnyPlan narrow = quant_skinny_plan_for(k, n, m, 4, 4);
  if (!wide) return narrow;
  const QuantSkinnyPlan any = quant_skinny_plan_for(k, n, m, 4, 8);
  if (any.nw != 4 && (any.cost <= 0.75 * narrow.cost || (m > 64 && any.cost < 0.999 * narrow.cost))) return any;
  return narrow;
}
static int quant_skinny_splitk(int k, int n, int64_t m) {            // (workspace sizing: the larger of the two plans' splits)
  const int a = quant_skinny_plan(k, n, m, true).sk, b = quant_skinny_plan(k, n, m, false).sk;
  return a > b ? a : b;
}

template <typename TO, bool FP8>
static int launch_quant_skinny(const GemmArgs& a, const float* rs, const bf16_t* cs, int64_t m, void* slab_ws, hipStream_t s) {
  constexpr bool kWide = std::is_same<TO, bf16_t>::value;   // (other output types: four-wave workgroups only — compile time)
  const QuantSkinnyPlan plan = quant_skinny_plan(a.K, a.N, m, kWide);
  const int sk = plan.sk;
  // More than 64 rows run as two row blocks of 64 (grid.z): a 128-row workgroup needs 104 KiB of LDS (one workgroup of four
  // waves per CU) and reads its activation fragments from LDS eight times per weight byte; two 64-row workgroups fit a CU
  // together, and the second reads the weight lines the first just pulled into L2.
  const dim3 grid(static_cast<unsigned>((a.N / 16 + plan.nw - 1) / plan.nw), static_cast<unsigned>(sk), m > 64 ? 2u : 1u);
  const uint8_t* A = static_cast<const uint8_t*>(a.A);
  const uint8_t* W = static_cast<const uint8_t*>(a.W);
  TO* C = static_cast<TO*>(a.C);
  const int M = static_cast<int>(m);
#ifdef MOJO_HIP_BUILD_EXPERIMENTS
  const int slot = sk > 1 && plan.nw == 4 ? splitk_take_slot(static_cast<int64_t>(grid.x) * grid.z) : -1;   // K slices combined inside the launch
#else
  const int slot = -1;
#endif
#define SKINNY(MT_, NT_, NW_) hipLaunchKernelGGL((quant_skinny_kernel<TO, FP8, MT_, NT_, NW_>), grid, dim3(NW_ * 64), 0, s, A, W, rs, cs, C, slab_ws, M, a.K, a.N, sk, slot)
#define SKINNY_NW(MT_, NT_)                                                                                   \
  do {                                                                                                        \
    if constexpr (kWide) {                                                                                    \
      switch (plan.nw) { case 6: SKINNY(MT_, NT_, 6); break; case 7: SKINNY(MT_, NT_, 7); break;               \
                         case 8: SKINNY(MT_, NT_, 8); break; default: SKINNY(MT_, NT_, 4); break; }            \
    } else {                                                                                                  \
      SKINNY(MT_, NT_, 4);                                                                                    \
    }                                                                                                         \
  } while (0)
  if (m <= 16) SKINNY_NW(1, true); else if (m <= 32) SKINNY_NW(2, true); else if (m <= 64) SKINNY_NW(4, true); else SKINNY_NW(4, false);
#undef SKINNY_NW
#undef SKINNY
  MOJO_CHECK_LAUNCH("quant_gemm(skinny)");
  note_launch("quant_skinny:waves%d:splitk%d", plan.nw, sk);
  if (sk > 1 && slot < 0) {
    int64_t blocks = ceil_div(m * a.N, 256);
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (FP8)
      hipLaunchKernelGGL((quant_finalize_kernel<TO, float>), dim3(blocks), dim3(256), 0, s, static_cast<const float*>(slab_ws), sk, m, a.N, rs, cs, C);
    else
      hipLaunchKernelGGL((quant_finalize_kernel<TO, int>), dim3(blocks), dim3(256), 0, s, static_cast<const int*>(slab_ws), sk, m, a.N, rs, cs, C);
    MOJO_CHECK_LAUNCH("quant_gemm(finalize)");
  }
  return MOJO_OK;
}

// Few output tiles (decode-sized M): cut K so that ~256 workgroups stream the weight concurrently.
static int quant_splitk(int64_t m, int k, int n) {
  if (const int v = static_cast<int>(MOJO_SWITCH("MOJO_HIP_GEMM_SPLITK", 0)); v >= 1) return v;
  const int64_t tiles = ceil_div(m, 256) * ceil_div(n, 256);
  const int nkt = k / 128;
  if (tiles >= 128 || n % 4 != 0 || nkt < 2) return 1;
  int64_t sk = ceil_div(256, tiles);
  if (sk > nkt / 2) sk = nkt / 2;                       // at least two K-tiles per slice
  if (sk > 32) sk = 32;
  return sk < 1 ? 1 : static_cast<int>(sk);
}

// 128-row tiles (gemm_tile128_core.h) for a mid-size M with [N,K] weights: the launches the 256 x 256 kernel can only fill by
// cutting K into fp32 / int32 slabs.  Same time model as the 16-bit products (gemm_api.hip, gemm_dense_prefers_tile128): a
// K-tile is 128 bytes of K in both, and the 128-row tile is bound by its fill, not by the (twice as fast) 8-bit MFMAs; the
// 256 x 256 kernel's dequantising epilogue adds ~10 us (its scales are fetched after the K loop; here they are requested before
// it).  Measured, both kernels forced in one process, 80 shapes per dtype (profiles/r5_quant_tile128_ab.txt): M 1024 x 4096 x 4096
// int8 39 -> 19 us, M 512 x 2048 x 7168 48 -> 12 us; over the grid the rule is 63 % faster (geometric mean) than the 256 x 256
// kernel alone and within 0.7 % of always picking the faster form.
static bool quant_tile128_ok(int64_t m, const GemmArgs& a, int out_elt_bytes, bool fp8) {
  const bool layout = (a.w_k == 1 && a.w_n % 16 == 0) ||                                      // [N,K] (trans_weight)
                      (!fp8 && a.w_n == 1 && a.w_k % 16 == 0 && a.N % 16 == 0 && a.N >= 16);    // [K,N], int8: transposed byte reads
  return (m > 64 || a.w_n == 1) && layout && a.K >= 128 && a.K % 128 == 0 && a.lda % 16 == 0 && a.ldc % 4 == 0 &&   // (<= 128 rows with [N,K] weights never get here: the weight-streaming kernel)
         aligned_to(a.A, 16) && aligned_to(a.W, 16) && aligned_to(a.C, 4 * out_elt_bytes);
}
// The 128 x 128 tiles' own K split (gemm_api.hip, gemm_tile128_splitk: the same model — a K-tile is 128 bytes of K in both
// element sizes — with the [K,N] byte reads' K-tile time): few tiles over a long K, M 256 x 8192 x 1024 = 16 tiles over 64 K-tiles.
// The slices' int32 / fp32 accumulators go to slabs; quant_finalize_kernel sums them in slice order and applies the scales.
static int quant_tile128_splitk(int64_t m, int k, int n, bool w_nmajor, double* modelled_us) {
  const int64_t tiles = ceil_div(m, 128) * ceil_div(n, 128), nkt = k / 128;
  auto cost = [&](int64_t sk) {
    const double busy = static_cast<double>(tiles * sk) / 256.0;
    double loop = static_cast<double>(ceil_div(nkt, sk)) * (w_nmajor ? 0.58 + 0.08 * busy : 0.30 + 0.125 * busy);
    const double hbm = (static_cast<double>(k) * n + static_cast<double>(m) * k) / 6.5e6;
    if (loop < hbm) loop = hbm;
    return sk == 1 ? 5.0 + loop : 6.0 + loop + 0.26 * static_cast<double>(sk * m * n) * 4.0 / 1e6 + 0.44 * sk;
  };
  *modelled_us = cost(1);
  if (tiles > 256 || n % 4 != 0) { *modelled_us = 1e30; return 1; }
  if (int64_t sk = MOJO_SWITCH("MOJO_HIP_GEMM_SPLITK", 0); sk > 0) {
    if (sk > nkt) sk = nkt;
    *modelled_us = cost(sk);
    return static_cast<int>(sk);
  }
  int64_t best = 1;
  double best_cost = cost(1) * 0.95;
  for (int64_t sk = 2; sk <= 16 && tiles * sk <= 256 && sk <= nkt / 4; ++sk) {
    const double c = cost(sk);
    if (c < best_cost * 0.999) { best = sk; best_cost = c; }
  }
  if (best > 1) *modelled_us = best_cost;
  return static_cast<int>(best);
}
static bool quant_prefers_tile128(int64_t m, int k, int n, bool w_nmajor, int* splitk128) {
  const int64_t narrow = ceil_div(m, 128) * ceil_div(n, 128), wide = ceil_div(m, 128) * ceil_div(n, 256), nkt = k / 128;
  double t128;
  *splitk128 = 1;
  // ([K,N], int8: the transposed byte reads cost more than the 16-bit ones — 0.58-0.66 us per K-tile of a 128 x 128 tile
  // against 0.30-0.43 for [N,K]; profiles/r5_quant_tile128_ab_kn.txt)
  if (narrow <= 256) *splitk128 = quant_tile128_splitk(m, k, n, w_nmajor, &t128);
  else if (wide <= 256) t128 = 6.0 + nkt * (w_nmajor ? 0.90 + 0.25 * wide / 256.0 : 0.34 + 0.47 * wide / 256.0);
  else return false;
  const int64_t tiles = ceil_div(m, 256) * ceil_div(n, 256), sk = quant_splitk(m, k, n);
  const double t256 = static_cast<double>(ceil_div(tiles * sk, 256)) * (static_cast<double>(nkt) / sk) * 1.06 + 10.0 +
                      (sk > 1 ? 2.0 * sk * m * n * 4.0 / 5e6 + 4.0 : 0.0);
  return t128 < 1.2 * t256;
}

template <typename TO>
static int run_quant(GemmArgs a, const float* rs, const bf16_t* cs, int64_t m, int quant_dtype, void* slab_ws,
                     hipStream_t s) {
  const bool fp8 = quant_dtype == MOJO_F8E4M3;
  if (quant_gemv_ok(m, a) && (gemm_skinny_mask() & SKINNY_GEMV))
    return fp8 ? launch_quant_gemv<TO, true>(a, rs, cs, m, s) : launch_quant_gemv<TO, false>(a, rs, cs, m, s);
  if (quant_skinny_ok(m, a) && (gemm_skinny_mask() & SKINNY_UNIFORM))
    return fp8 ? launch_quant_skinny<TO, true>(a, rs, cs, m, slab_ws, s) : launch_quant_skinny<TO, false>(a, rs, cs, m, slab_ws, s);
  if (quant_tile128_ok(m, a, sizeof(TO), fp8)) {
    const int f = g128::forced_choice();
    int sk128 = 1;
    const bool prefers = quant_prefers_tile128(m, a.K, a.N, a.w_n == 1, &sk128);
    if (f < 0 ? prefers : f == 1) {
      if (sk128 > 1 && a.ldc == a.N) {                 // (the finalize writes a dense [M, N]; mojo_hip_quant_gemm_workspace_bytes covers the slabs)
        a.splitk = sk128; a.slab = slab_ws; a.slab_rows = static_cast<int>(m);
      }
      int rc;
      if (fp8) {
        g256::EpilogueDequant<TO, f32x4> epi{static_cast<TO*>(a.C), a.ldc, rs, cs, 0.f, true};
        rc = g128::launch<g256::PolF8>(a, epi, m, s);
      } else {
        g256::EpilogueDequant<TO, i32x4> epi{static_cast<TO*>(a.C), a.ldc, rs, cs, 0.f, true};
        rc = g128::launch<g256::PolI8>(a, epi, m, s);
      }
      if (rc || a.splitk == 1) return rc;
      int64_t blocks = ceil_div(m * a.N, 256);
      if (blocks > 256 * 8) blocks = 256 * 8;
      if (fp8)
        hipLaunchKernelGGL((quant_finalize_kernel<TO, float>), dim3(blocks), dim3(256), 0, s, static_cast<const float*>(slab_ws), a.splitk, m, a.N, rs, cs, static_cast<TO*>(a.C));
      else
        hipLaunchKernelGGL((quant_finalize_kernel<TO, int>), dim3(blocks), dim3(256), 0, s, static_cast<const int*>(slab_ws), a.splitk, m, a.N, rs, cs, static_cast<TO*>(a.C));
      MOJO_CHECK_LAUNCH("quant_gemm(finalize)");
      return MOJO_OK;
    }
  }
  if (g256::gemm256_layout_ok(a, 1)) {
    const int sk = quant_splitk(m, a.K, a.N);
    if (sk > 1) {
      a.splitk = sk; a.slab = slab_ws; a.slab_rows = static_cast<int>(m);
      int rc;
      if (fp8) {
        g256::EpilogueDequant<TO, f32x4> epi{static_cast<TO*>(a.C), a.ldc, rs, cs, 0.f};
        rc = g256::gemm256_launch<g256::PolF8>(a, epi, m, s);
      } else {
        g256::EpilogueDequant<TO, i32x4> epi{static_cast<TO*>(a.C), a.ldc, rs, cs, 0.f};
        rc = g256::gemm256_launch<g256::PolI8>(a, epi, m, s);
      }
      if (rc) return rc;
      int64_t blocks = ceil_div(m * a.N, 256);
      if (blocks > 256 * 8) blocks = 256 * 8;
      if (fp8)
        hipLaunchKernelGGL((quant_finalize_kernel<TO, float>), dim3(blocks), dim3(256), 0, s, static_cast<const float*>(slab_ws), sk, m, a.N, rs, cs, static_cast<TO*>(a.C));
      else
        hipLaunchKernelGGL((quant_finalize_kernel<TO, int>), dim3(blocks), dim3(256), 0, s, static_cast<const int*>(slab_ws), sk, m, a.N, rs, cs, static_cast<TO*>(a.C));
      MOJO_CHECK_LAUNCH("quant_gemm(finalize)");
      return MOJO_OK;
    }
    const bool no_stage = MOJO_SWITCH("MOJO_HIP_GEMM_STAGE_ROWS", 1) == 0;
    a.stage_rows = (!no_stage && sizeof(TO) == 2 && a.ldc % 8 == 0 && aligned_to(a.C, 16)) ? 1 : 0;   // row-staged stores (gemm256_core.h)
    if (fp8) {
      g256::EpilogueDequant<TO, f32x4> epi{static_cast<TO*>(a.C), a.ldc, rs, cs, 0.f};
      return g256::gemm256_launch<g256::PolF8>(a, epi, m, s);
    }
    g256::EpilogueDequant<TO, i32x4> epi{static_cast<TO*>(a.C), a.ldc, rs, cs, 0.f};
    return g256::gemm256_launch<g256::PolI8>(a, epi, m, s);
  }
  int64_t blocks = ceil_div(m * a.N, 256);
  if (blocks > 256 * 32) blocks = 256 * 32;
  const uint8_t* A = static_cast<const uint8_t*>(a.A);
  const uint8_t* W = static_cast<const uint8_t*>(a.W);
  if (fp8)
    hipLaunchKernelGGL((quant_gemm_generic_kernel<TO, true>), dim3(blocks), dim3(256), 0, s, A, W, rs, cs,
                       static_cast<TO*>(a.C), m, a.K, a.N, a.w_k, a.w_n);
  else
    hipLaunchKernelGGL((quant_gemm_generic_kernel<TO, false>), dim3(blocks), dim3(256), 0, s, A, W, rs, cs,
                       static_cast<TO*>(a.C), m, a.K, a.N, a.w_k, a.w_n);
  MOJO_CHECK_LAUNCH("quant_gemm_generic");
  note_launch("quant_generic");
  return MOJO_OK;
}

}  // namespace mojo

using namespace mojo;

extern "C" int64_t mojo_hip_quant_gemm_workspace_bytes(int64_t m, int64_t k, int64_t n) {
  // the largest split any path may choose for this shape (the skinny path needs [N,K] weights, which is not known here)
  int sk = (k > 0 && k % 128 == 0) ? quant_splitk(m, static_cast<int>(k), static_cast<int>(n)) : 1;
  if (m <= 128 && k > 0 && k % 256 == 0 && n % 64 == 0) {
    const int s2 = quant_skinny_splitk(static_cast<int>(k), static_cast<int>(n), m);
    if (s2 > sk) sk = s2;
  }
  if (k > 0 && k % 128 == 0) {                          // the 128-row tiles' own split, either weight layout
    double t;
    for (int kn = 0; kn < 2; ++kn) {
      const int s3 = quant_tile128_splitk(m, static_cast<int>(k), static_cast<int>(n), kn != 0, &t);
      if (s3 > sk) sk = s3;
    }
  }
  return 64 + (sk > 1 ? static_cast<int64_t>(sk) * m * n * 4 : 0);
}

extern "C" int mojo_hip_quant_gemm(const void* input, const void* weight, const float* input_scale,
                                   const void* weight_scale, void* out, int64_t m, int64_t k, int64_t n,
                                   int trans_weight, int quant_dtype, int out_dtype, void* workspace,
                                   int64_t workspace_bytes, mojo_stream_t stream) {
  MOJO_REQUIRE(k > 0 && n > 0 && m >= 0, MOJO_EINVAL, "quant_gemm: bad shape");
  if (m == 0) return MOJO_OK;
  MOJO_REQUIRE(input && weight && input_scale && weight_scale && out, MOJO_EINVAL, "quant_gemm: null pointer");
  MOJO_REQUIRE(quant_dtype == MOJO_I8 || quant_dtype == MOJO_F8E4M3, MOJO_EUNSUPPORTED,
               "quant_gemm: quant dtype %d (int8 / fp8-e4m3 only)", quant_dtype);
  MOJO_REQUIRE(out_dtype == MOJO_F32 || out_dtype == MOJO_F16 || out_dtype == MOJO_BF16, MOJO_EUNSUPPORTED,
               "quant_gemm: output dtype %d", out_dtype);
  MOJO_REQUIRE(m < (1LL << 31) && k < (1LL << 31) && n < (1LL << 31), MOJO_EUNSUPPORTED, "quant_gemm: dimension too large");
  MOJO_REQUIRE(workspace && workspace_bytes >= mojo_hip_quant_gemm_workspace_bytes(m, k, n) && aligned_to(workspace, 16),
               MOJO_EWORKSPACE, "quant_gemm: workspace too small");
  GemmArgs a;
  a.A = input; a.W = weight; a.C = out; a.bias = nullptr;
  a.lda = k; a.ldc = n; a.w_group = 0;
  if (trans_weight) { a.w_k = 1; a.w_n = k; } else { a.w_k = n; a.w_n = 1; }
  a.K = static_cast<int>(k); a.N = static_cast<int>(n); a.G = 1;
  int32_t* ws = static_cast<int32_t*>(workspace);
  a.row_start = ws; a.tile_start = ws + 2;
  hipStream_t s = static_cast<hipStream_t>(stream);
  a.uniform_rows = static_cast<int>(m);
  const bf16_t* cs = static_cast<const bf16_t*>(weight_scale);
  void* slab_ws = static_cast<char*>(workspace) + 64;
  switch (out_dtype) {
    case MOJO_F32: return run_quant<float>(a, input_scale, cs, m, quant_dtype, slab_ws, s);
    case MOJO_F16: return run_quant<f16_t>(a, input_scale, cs, m, quant_dtype, slab_ws, s);
    default: return run_quant<bf16_t>(a, input_scale, cs, m, quant_dtype, slab_ws, s);
  }
}
