// Probe: how fast does the dispatcher place workgroups of a given footprint?  Each workgroup stamps the chip-wide
// 100 MHz clock when it starts, holds its slot for `hold_us`, and exits.  Printed: when the n-th workgroup started.
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/dispatch_rate.hip -o scripts/probes/build/dispatch_rate
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-result"
#pragma clang diagnostic ignored "-Wunused-value"
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

template <int VG>
__global__ __launch_bounds__(256, 2) void hold_kernel(unsigned* t_start, unsigned* t_end, int hold_ticks, float* sink) {
  extern __shared__ float lds[];
  const unsigned t0 = static_cast<unsigned>(__builtin_amdgcn_s_memrealtime());
  float keep[VG];                                           // occupy registers
#pragma unroll
  for (int i = 0; i < VG; ++i) keep[i] = threadIdx.x * 0.5f + i;
  if (threadIdx.x == 0) t_start[blockIdx.x] = t0;
  lds[threadIdx.x] = keep[0];
  while (static_cast<unsigned>(__builtin_amdgcn_s_memrealtime()) - t0 < static_cast<unsigned>(hold_ticks)) {
#pragma unroll
    for (int i = 0; i < VG; ++i) keep[i] = keep[i] * 1.0001f + 1.f;
    __builtin_amdgcn_s_sleep(8);
  }
  float s = lds[(threadIdx.x + 1) & 255];
#pragma unroll
  for (int i = 0; i < VG; ++i) s += keep[i];
  if (s == 12345.678f) sink[0] = s;
  if (threadIdx.x == 0) t_end[blockIdx.x] = static_cast<unsigned>(__builtin_amdgcn_s_memrealtime());
}

template <int VG>
static void run(const char* name, int grid, int lds_bytes, float hold_us) {
  unsigned *ts, *te; float* sink;
  hipMalloc(&ts, grid * 4); hipMalloc(&te, grid * 4); hipMalloc(&sink, 4);
  hipFuncSetAttribute(reinterpret_cast<const void*>(hold_kernel<VG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(hold_kernel<VG>, dim3(grid), dim3(256), lds_bytes, 0, ts, te, static_cast<int>(hold_us * 100), sink);
  hipDeviceSynchronize();
  std::vector<unsigned> s(grid), e(grid);
  hipMemcpy(s.data(), ts, grid * 4, hipMemcpyDeviceToHost);
  hipMemcpy(e.data(), te, grid * 4, hipMemcpyDeviceToHost);
  const unsigned base = *std::min_element(s.begin(), s.end());
  std::vector<unsigned> so(s);
  std::sort(so.begin(), so.end());
  unsigned last_end = 0;
  for (int i = 0; i < grid; ++i) last_end = std::max(last_end, e[i] - base);
  printf("%-34s grid %5d lds %6d hold %5.1f us | start of WG #64 %6.2f  #128 %6.2f  #256 %6.2f  #512 %6.2f  #1024 %6.2f  last %6.2f us | span %7.2f us\n",
         name, grid, lds_bytes, hold_us, (so[std::min(63, grid - 1)] - base) / 100.0, (so[std::min(127, grid - 1)] - base) / 100.0,
         (so[std::min(255, grid - 1)] - base) / 100.0, (so[std::min(511, grid - 1)] - base) / 100.0, (so[std::min(1023, grid - 1)] - base) / 100.0,
         (so[grid - 1] - base) / 100.0, last_end / 100.0);
  hipFree(ts); hipFree(te); hipFree(sink);
}

// Placement rule: blocks whose position inside their XCD ((id >> 3) % period) is 0 hold their slot `long_us`, the others
// `short_us`.  If blocks are dealt to the shader engines of an XCD in strict rotation and in order, the long blocks of
// period 4 pile onto one engine and the launch takes (long blocks per XCD / slots of one engine) rounds of long_us.
__global__ __launch_bounds__(256, 2) void pattern_kernel(unsigned* t_start, unsigned* t_end, unsigned* hw, int period, int long_ticks, int short_ticks) {
  extern __shared__ float lds[];
  const unsigned t0 = static_cast<unsigned>(__builtin_amdgcn_s_memrealtime());
  const int hold = ((blockIdx.x >> 3) % period) == 0 ? long_ticks : short_ticks;
  if (threadIdx.x == 0) {
    t_start[blockIdx.x] = t0;
    hw[blockIdx.x * 2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    hw[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
  }
  lds[threadIdx.x] = 1.f;
  while (static_cast<unsigned>(__builtin_amdgcn_s_memrealtime()) - t0 < static_cast<unsigned>(hold)) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) t_end[blockIdx.x] = static_cast<unsigned>(__builtin_amdgcn_s_memrealtime());
}

static void run_pattern(int grid, int period, float long_us, float short_us) {
  const int lds_bytes = 65536 + 4112;
  unsigned *ts, *te, *hw;
  hipMalloc(&ts, grid * 4); hipMalloc(&te, grid * 4); hipMalloc(&hw, grid * 8);
  hipFuncSetAttribute(reinterpret_cast<const void*>(pattern_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(pattern_kernel, dim3(grid), dim3(256), lds_bytes, 0, ts, te, hw, period, static_cast<int>(long_us * 100), static_cast<int>(short_us * 100));
  hipDeviceSynchronize();
  std::vector<unsigned> s(grid), e(grid), h(grid * 2);
  hipMemcpy(s.data(), ts, grid * 4, hipMemcpyDeviceToHost);
  hipMemcpy(e.data(), te, grid * 4, hipMemcpyDeviceToHost);
  hipMemcpy(h.data(), hw, grid * 8, hipMemcpyDeviceToHost);
  const unsigned base = *std::min_element(s.begin(), s.end());
  unsigned last_end = 0;
  for (int i = 0; i < grid; ++i) last_end = std::max(last_end, e[i] - base);
  int n_long = 0;
  for (int i = 0; i < grid; ++i) n_long += ((i >> 3) % period) == 0;
  const double ideal = (n_long * long_us + (grid - n_long) * short_us) / 512.0;
  printf("pattern: grid %d, every %d-th block of an XCD holds %.0f us, the others %.0f us: span %.1f us (slot-time / 512 slots = %.1f us)\n",
         grid, period, long_us, short_us, last_end / 100.0, ideal);
  // where did the long blocks of XCD 0 go?  HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
  printf("   long blocks seen by XCC 0, first 24 in block order: (se,sh,cu @ start us) ");
  int shown = 0;
  for (int i = 0; i < grid && shown < 24; ++i)
    if (((i >> 3) % period) == 0 && (i & 7) == 0) {
      printf("(%u,%u,%u xcc%u @%.1f) ", (h[2 * i] >> 13) & 7, (h[2 * i] >> 12) & 1, (h[2 * i] >> 8) & 15, h[2 * i + 1] & 15, (s[i] - base) / 100.0);
      ++shown;
    }
  printf("\n");
  hipFree(ts); hipFree(te); hipFree(hw);
}

int main() {
  run_pattern(4096, 4, 30.f, 1.f);
  run_pattern(4096, 3, 30.f, 1.f);
  run_pattern(4096, 5, 30.f, 1.f);
  run_pattern(4096, 8, 30.f, 1.f);
  run_pattern(4096, 2, 30.f, 1.f);
  run<8>("small regs, no LDS", 2048, 1024, 18.f);
  run<8>("small regs, 64 KiB LDS", 2048, 65536 + 4112, 18.f);
  run<100>("~128 VGPRs, no LDS", 2048, 1024, 18.f);
  run<100>("~128 VGPRs, 64 KiB LDS", 2048, 65536 + 4112, 18.f);
  run<8>("small regs, 64 KiB LDS, hold 40", 2048, 65536 + 4112, 40.f);
  run<8>("small regs, 64 KiB LDS, hold 5", 4096, 65536 + 4112, 5.f);
  run<8>("small regs, no LDS, hold 5", 4096, 1024, 5.f);
  return 0;
}
