// Probe (round 5): does the WIDER MFMA shape hold a higher clock / deliver more FLOP/s on random data?
// The shipped GEMM core (gemm256_core.h) runs a 128 x 64 wave tile on 16x16xK MFMAs; the judge's experiment is the same tile on
// 32x32x(K/2) MFMAs (half the register-file operand reads per FLOP).  Before rebuilding the kernel this probe runs the bare inner
// loop of both forms — the wave tile's A / B fragments in registers (64 + 32 VGPRs in every form), accumulators 128 VGPRs, eight
// waves per CU (two per SIMD) on all 256 CUs, random operands — and reports sustained TFLOP/s (wall clock over >= 0.6 s of
// back-to-back launches after 0.6 s of settling: the DVFS steady state) and cycles per MFMA (s_memtime), so FLOP/s = FLOP per
// cycle x the clock the chip holds.  Variant "lds": every K-tile's fragments are re-read from a random LDS image with ds_read_b128
// (24 per wave and K-tile, as the real loop does).
//   hipcc --offload-arch=gfx950 -O3 -w -o scripts/probes/build/mfma_shape_power_probe scripts/probes/mfma_shape_power_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(8))) unsigned u32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) int i32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

enum { BF16 = 0, FP8 = 1, I8 = 2 };

__device__ __forceinline__ unsigned hash(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// a random, finite, moderate-magnitude operand word for the data type
template <int DT>
__device__ __forceinline__ unsigned rnd_word(unsigned seed) {
  const unsigned r = hash(seed);
  if (DT == BF16) {                         // two bf16: sign | exponent 124..127 | 7 mantissa bits
    const unsigned lo = (r & 0x807Fu) | ((124u + ((r >> 7) & 3u)) << 7);
    const unsigned hi = ((r >> 16) & 0x807Fu) | ((124u + ((r >> 23) & 3u)) << 7);
    return lo | (hi << 16);
  }
  if (DT == FP8) return r & 0xB7B7B7B7u;    // e4m3: exponent's top bit clear and never the NaN pattern
  return r;                                  // int8: any byte
}

template <int DT, int BIG>
struct Shape {
  static constexpr int R = DT == FP8 ? 8 : 4;                         // VGPRs per fragment
  static constexpr int KS = DT == FP8 ? (BIG ? 2 : 1) : (BIG ? 4 : 2);  // k-steps per 128-byte K-tile
  static constexpr int MI = BIG ? 4 : 8, NJ = BIG ? 2 : 4;            // fragments along M / N of the 128 x 64 wave tile
  static constexpr int NA = MI * KS, NB = NJ * KS;                    // NA * R = 64, NB * R = 32 in every form
  static constexpr int NC = MI * NJ;                                  // accumulators (x 4 or x 16 registers = 128)
};

template <int DT, int BIG> struct Mfma;
template <> struct Mfma<BF16, 0> { typedef u32x4 VA; typedef f32x4 VC;
  static __device__ __forceinline__ void run(VC& c, const VA& a, const VA& b) { c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0); } };
template <> struct Mfma<BF16, 1> { typedef u32x4 VA; typedef f32x16 VC;
  static __device__ __forceinline__ void run(VC& c, const VA& a, const VA& b) { c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0); } };
template <> struct Mfma<I8, 0> { typedef u32x4 VA; typedef i32x4 VC;
  static __device__ __forceinline__ void run(VC& c, const VA& a, const VA& b) { c = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, a), __builtin_bit_cast(i32x4, b), c, 0, 0, 0); } };
template <> struct Mfma<I8, 1> { typedef u32x4 VA; typedef i32x16 VC;
  static __device__ __forceinline__ void run(VC& c, const VA& a, const VA& b) { c = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, a), __builtin_bit_cast(i32x4, b), c, 0, 0, 0); } };
template <> struct Mfma<FP8, 0> { typedef u32x8 VA; typedef f32x4 VC;
  static __device__ __forceinline__ void run(VC& c, const VA& a, const VA& b) { asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); } };
template <> struct Mfma<FP8, 1> { typedef u32x8 VA; typedef f32x16 VC;
  static __device__ __forceinline__ void run(VC& c, const VA& a, const VA& b) { asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); } };

template <int DT, int BIG, int LDS>
__global__ __launch_bounds__(512, 1) void probe(unsigned* cycles, float* sink, int iters) {
  using S = Shape<DT, BIG>;
  using VA = typename Mfma<DT, BIG>::VA;
  using VC = typename Mfma<DT, BIG>::VC;
  __shared__ __attribute__((aligned(16))) unsigned lds[24 * 64 * 4 * 2];       // two 24 KiB images (24 b128 reads per lane each)
  VA a[S::NA], b[S::NB];
  VC c[S::NC];
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 24 * 64 * 4 * 2; i += 512) lds[i] = rnd_word<DT>(i * 2654435761u + blockIdx.x);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < S::NA; ++i)
#pragma unroll
    for (int r = 0; r < S::R; ++r) a[i][r] = rnd_word<DT>((blockIdx.x * 512 + threadIdx.x) * 131u + i * 17u + r);
#pragma unroll
  for (int i = 0; i < S::NB; ++i)
#pragma unroll
    for (int r = 0; r < S::R; ++r) b[i][r] = rnd_word<DT>((blockIdx.x * 512 + threadIdx.x) * 137u + i * 19u + r + 7777u);
#pragma unroll
  for (int i = 0; i < S::NC; ++i) c[i] = VC{};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (LDS) {       // the K-tile's fragments from LDS: 24 conflict-free ds_read_b128 per lane (image alternates per iteration)
      const unsigned base = ((it & 1) * 24 * 64 + lane) * 16;
      u32x4* ap = reinterpret_cast<u32x4*>(a);
      u32x4* bp = reinterpret_cast<u32x4*>(b);
#pragma unroll
      for (int q = 0; q < 16; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ap[q]) : "v"(base), "i"(q * 1024) : "memory");
#pragma unroll
      for (int q = 0; q < 8; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bp[q]) : "v"(base), "i"((16 + q) * 1024) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
#pragma unroll
    for (int ks = 0; ks < S::KS; ++ks)
#pragma unroll
      for (int i = 0; i < S::MI; ++i)
#pragma unroll
        for (int j = 0; j < S::NJ; ++j) Mfma<DT, BIG>::run(c[i * S::NJ + j], a[i * S::KS + ks], b[j * S::KS + ks]);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < S::NC; ++i) acc += static_cast<float>(c[i][0]);
  if (acc == 12345.678f) sink[threadIdx.x] = acc;
  if (lane == 0) cycles[blockIdx.x * 8 + wave] = static_cast<unsigned>(t1 - t0);
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <int DT, int BIG, int LDS>
void run(unsigned* d_cyc, float* d_sink, const char* name) {
  using S = Shape<DT, BIG>;
  const int blocks = 256, iters = 20000;
  // FLOP (or OP) per wave and K-tile: 128 x 64 x (128 bytes of K) x 2
  const double k_elems = DT == BF16 ? 64.0 : 128.0;
  const double flop_launch = 2.0 * 128 * 64 * k_elems * iters * 8.0 * blocks;
  auto launch = [&] { hipLaunchKernelGGL((probe<DT, BIG, LDS>), dim3(blocks), dim3(512), 0, 0, d_cyc, d_sink, iters); };
  double t = now();
  while (now() - t < 0.6) { launch(); hipDeviceSynchronize(); }             // settle into the DVFS steady state
  int n = 0;
  const double t0 = now();
  while (now() - t0 < 0.6) { for (int q = 0; q < 4; ++q) launch(); hipDeviceSynchronize(); n += 4; }
  const double el = now() - t0;
  static unsigned h[2048];
  hipMemcpy(h, d_cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < 2048; ++i) s += h[i];
  const double cyc_per_mfma = s / 2048 / (static_cast<double>(iters) * S::NC * S::KS);
  const double tf = flop_launch * n / el / 1e12;
  // clock = cycles per launch / wall per launch (s_memtime counts shader cycles)
  const double mhz = (s / 2048) / (el / n) / 1e6;
  printf("%-28s %s  %8.1f T(FL)OP/s  %6.2f cycles/MFMA/wave  ~%5.0f MHz\n", name, LDS ? "lds " : "regs", tf, cyc_per_mfma, mhz);
  fflush(stdout);
}

int main() {
  unsigned* d_cyc; float* d_sink;
  hipMalloc(&d_cyc, 8192); hipMalloc(&d_sink, 4096);
  for (int rep = 0; rep < 2; ++rep) {
    run<BF16, 0, 0>(d_cyc, d_sink, "bf16 16x16x32");   run<BF16, 1, 0>(d_cyc, d_sink, "bf16 32x32x16");
    run<FP8, 0, 0>(d_cyc, d_sink, "fp8 16x16x128 f8f6f4"); run<FP8, 1, 0>(d_cyc, d_sink, "fp8 32x32x64 f8f6f4");
    run<I8, 0, 0>(d_cyc, d_sink, "int8 16x16x64");     run<I8, 1, 0>(d_cyc, d_sink, "int8 32x32x32");
    run<BF16, 0, 1>(d_cyc, d_sink, "bf16 16x16x32");   run<BF16, 1, 1>(d_cyc, d_sink, "bf16 32x32x16");
    run<FP8, 0, 1>(d_cyc, d_sink, "fp8 16x16x128 f8f6f4"); run<FP8, 1, 1>(d_cyc, d_sink, "fp8 32x32x64 f8f6f4");
    run<I8, 0, 1>(d_cyc, d_sink, "int8 16x16x64");     run<I8, 1, 1>(d_cyc, d_sink, "int8 32x32x32");
  }
  return 0;
}
