// Probe: can ONE wave per SIMD keep the matrix pipe full?  Cycles (s_memtime units) per MFMA per wave for 4 and 8 waves per CU
// (one and two per SIMD), 16x16x32 and 32x32x16 bf16, with 0..1 ds_read_b128 per MFMA.  Two waves per SIMD at full rate show
// what the pipe can do in the same units; a lone wave that needs more than half of that per MFMA leaves the pipe idle.
//   hipcc --offload-arch=gfx950 -O3 -w -o scripts/probes/mfma16_issue_probe.bin scripts/probes/mfma16_issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int WAVES, int FILL, int BIG>
__global__ __launch_bounds__(WAVES * 64, 1) void probe(unsigned* out, float* sink, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[32768];
  f32x4 c[16];
  f32x16 d[8];
  for (int i = 0; i < 16; ++i) c[i] = f32x4{0, 0, 0, 0};
  for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) d[i][e] = 0.f;
  u32x4 a = {threadIdx.x, 1, 2, 3}, b = {4, 5, 6, threadIdx.x};
  u32x4 ld = {0, 0, 0, 0};
  const unsigned la = (threadIdx.x & 63) * 16;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (BIG) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(d[u & 7]) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c[u]) : "v"(a), "v"(b));
      if (FILL == 1 && (u & 3) == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(ld) : "v"(la) : "memory");
      if (FILL == 2 && (u & 1) == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(ld) : "v"(la) : "memory");
      if (FILL == 3) asm volatile("ds_read_b128 %0, %1" : "=v"(ld) : "v"(la) : "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
    if (FILL) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ld) : : "memory");
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  float acc = __builtin_bit_cast(float, ld[0]);
  for (int i = 0; i < 16; ++i) acc += c[i][0];
  for (int i = 0; i < 8; ++i) acc += d[i][0];
  if (acc == 12345.678f) sink[threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = static_cast<unsigned>(t1 - t0);
}
template <int WAVES, int FILL, int BIG> void run(unsigned* d_out, float* d_sink) {
  const int iters = 2000, blocks = 256;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((probe<WAVES, FILL, BIG>), dim3(blocks), dim3(WAVES * 64), 0, 0, d_out, d_sink, iters);
  hipDeviceSynchronize();
  static unsigned h[2048];
  hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; int n = 0;
  for (int b = 0; b < 256; ++b) for (int w = 0; w < WAVES; ++w) { s += h[b * 8 + w]; ++n; }
  printf("%s bf16, %d waves/CU, ds_read_b128 every %s MFMA: %6.1f cycles per MFMA per wave\n", BIG ? "32x32x16" : "16x16x32", WAVES,
         FILL == 0 ? "- (none)" : FILL == 1 ? "4th" : FILL == 2 ? "2nd" : "1", s / n / (iters * 16.0));
}
int main() {
  unsigned* d_out; float* d_sink;
  hipMalloc(&d_out, 8192); hipMalloc(&d_sink, 4096);
  run<4, 0, 0>(d_out, d_sink); run<4, 1, 0>(d_out, d_sink); run<4, 2, 0>(d_out, d_sink); run<4, 3, 0>(d_out, d_sink);
  run<8, 0, 0>(d_out, d_sink); run<8, 1, 0>(d_out, d_sink); run<8, 2, 0>(d_out, d_sink); run<8, 3, 0>(d_out, d_sink);
  run<4, 0, 1>(d_out, d_sink); run<4, 3, 1>(d_out, d_sink); run<8, 0, 1>(d_out, d_sink); run<8, 3, 1>(d_out, d_sink);
  return 0;
}
