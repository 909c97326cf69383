// Probe: cycles per v_mfma_f32_32x32x16_bf16 for ONE wave per SIMD (256-thread workgroups, one per CU) as a function of
//   * where the accumulators live (AGPR C/D with VGPR A/B  |  VGPR C/D with AGPR A/B),
//   * how many independent accumulator chains are interleaved (1, 2, 4, 8),
//   * which vector instructions are issued between two MFMAs (none | v_fma | v_fma v_exp | v_fma v_exp v_add v_cvt v_max3 | + ds_read_b128).
// It answers "why does a gap of prefill_w64_kernel take 49 cycles": the hardware model behind the CDNA4 guide's
// 'fillers hide under the MFMA' rule, measured with the kernel's own asm forms.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_gap scripts/probes/mfma_gap_probe.hip && /tmp/mfma_gap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define MFMA_A(c, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b))
#define MFMA_V(c, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "a"(a), "a"(b))

template <int FORM /* 0: C/D AGPR, 1: C/D VGPR */, int CHAINS, int FILL>
__global__ __launch_bounds__(256, 1) void probe(unsigned* out, float* sink, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[16384];
  f32x16 c[CHAINS];
#pragma unroll
  for (int i = 0; i < CHAINS; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) c[i][e] = 0.f;
  u32x4 a = {threadIdx.x, 1, 2, 3}, b = {4, 5, 6, threadIdx.x};
  float x = threadIdx.x * 1e-3f, y = 0.f, mx = 0.f, sc = 1.0001f;
  unsigned w = 0;
  u32x4 ld = {0, 0, 0, 0}, ld2 = {0, 0, 0, 0};
  unsigned long long tr0 = 0, tr1 = 0;
  float z = 0.f;
  const unsigned la = (threadIdx.x & 63) * 16;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t0 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if constexpr (FORM == 0) MFMA_A(c[u % CHAINS], a, b); else MFMA_V(c[u % CHAINS], a, b);
      if constexpr (FILL >= 1 && FILL <= 4) { asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(sc)); }
      if constexpr (FILL >= 2 && FILL <= 4) { asm volatile("v_exp_f32 %0, %1" : "=v"(y) : "v"(x)); }
      if constexpr (FILL >= 3 && FILL <= 4) {
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(mx) : "v"(sc));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(mx), "v"(sc));
        asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(sc), "v"(x));
      }
      if constexpr (FILL == 4 || FILL == 5 || FILL == 6) { asm volatile("ds_read_b128 %0, %1" : "=v"(ld) : "v"(la) : "memory"); }
      if constexpr (FILL == 6) { asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(ld2) : "v"(la) : "memory"); }
      if constexpr (FILL == 7) {                         // the five vector fillers on INDEPENDENT registers
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(sc));
        asm volatile("v_exp_f32 %0, %1" : "=v"(y) : "v"(sc));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(mx) : "v"(sc));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(sc), "v"(sc));
        asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(z) : "v"(sc));
      }
      if constexpr (FILL == 8) {
        asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:2048" : "=v"(tr0), "=v"(tr1) : "v"(la) : "memory");
      }
      if constexpr (FILL == 9) {                         // fma + exp + add (three fillers, independent)
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(sc));
        asm volatile("v_exp_f32 %0, %1" : "=v"(y) : "v"(sc));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(mx) : "v"(sc));
      }
      if constexpr (FILL == 10) {                        // fma + exp + add + cvt (four)
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(sc));
        asm volatile("v_exp_f32 %0, %1" : "=v"(y) : "v"(sc));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(mx) : "v"(sc));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(sc), "v"(sc));
      }
      if constexpr (FILL == 11) {                        // fma + exp + ds_read_b128
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(sc));
        asm volatile("v_exp_f32 %0, %1" : "=v"(y) : "v"(sc));
        asm volatile("ds_read_b128 %0, %1" : "=v"(ld) : "v"(la) : "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (FILL == 4 || FILL == 5 || FILL == 6 || FILL == 8 || FILL == 11)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ld), "+v"(ld2), "+v"(tr0), "+v"(tr1) : : "memory");
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  float acc = y + mx + z + __builtin_bit_cast(float, w) + __builtin_bit_cast(float, ld[0]) + __builtin_bit_cast(float, ld2[1]) + static_cast<float>(tr0 + tr1);
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) acc += c[i][0];
  if (acc == 12345.678f) sink[threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = static_cast<unsigned>(t1 - t0);
}

template <int FORM, int CHAINS, int FILL>
void run(const char* name, unsigned* d_out, float* d_sink) {
  const int iters = 2000, blocks = 256;
  hipLaunchKernelGGL((probe<FORM, CHAINS, FILL>), dim3(blocks), dim3(256), 0, 0, d_out, d_sink, iters);
  hipLaunchKernelGGL((probe<FORM, CHAINS, FILL>), dim3(blocks), dim3(256), 0, 0, d_out, d_sink, iters);
  hipDeviceSynchronize();
  static unsigned h[1024];
  hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < 1024; ++i) s += h[i];
  printf("%-28s chains %d  fill %d : %6.1f cycles per MFMA\n", name, CHAINS, FILL, s / 1024 / (iters * 16.0));
}

int main() {
  unsigned* d_out; float* d_sink;
  hipMalloc(&d_out, 4096); hipMalloc(&d_sink, 4096);
  printf("fill: 0 none, 1 v_fma, 2 + v_exp, 3 + v_add v_cvt_pk v_max3 (a dependent chain), 4 + ds_read_b128; 5 ds_read_b128 alone, 6 two of them,\n"
         "      7 the five vector fillers on independent registers, 8 two ds_read_b64_tr_b16, 9 fma exp add, 10 fma exp add cvt, 11 fma exp ds_read_b128\n");
#define ROW(FORM, NAME) \
  run<FORM, 1, 0>(NAME, d_out, d_sink); run<FORM, 2, 0>(NAME, d_out, d_sink); run<FORM, 4, 0>(NAME, d_out, d_sink); run<FORM, 8, 0>(NAME, d_out, d_sink); \
  run<FORM, 4, 1>(NAME, d_out, d_sink); run<FORM, 4, 2>(NAME, d_out, d_sink); run<FORM, 4, 3>(NAME, d_out, d_sink); run<FORM, 4, 4>(NAME, d_out, d_sink); \
  run<FORM, 8, 3>(NAME, d_out, d_sink); run<FORM, 8, 4>(NAME, d_out, d_sink); run<FORM, 4, 5>(NAME, d_out, d_sink); run<FORM, 4, 6>(NAME, d_out, d_sink); \
  run<FORM, 4, 7>(NAME, d_out, d_sink); run<FORM, 4, 8>(NAME, d_out, d_sink); run<FORM, 4, 9>(NAME, d_out, d_sink); run<FORM, 4, 10>(NAME, d_out, d_sink); run<FORM, 4, 11>(NAME, d_out, d_sink);
  ROW(0, "C/D AGPR, A/B VGPR")
  ROW(1, "C/D VGPR, A/B AGPR")
  return 0;
}
