// Probe: how fast can ONE workgroup per CU fill its LDS from HBM/L2 — LDS-DMA (global_load_lds b128) against register staging
// (global_load_dwordx4 + ds_write_b128) — at the same bytes in flight?  Two loader waves per workgroup (as mla512_ps_kernel),
// 1 KiB per wave-instruction, rows of 1152 bytes in 18-KiB pages picked by a shuffled table, `pairs` workgroups share a stream.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_fill scripts/probes/lds_fill_rate.hip && /tmp/lds_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <random>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
constexpr int PIECES = 18;            // KiB per loader wave and slot
constexpr int SLOT = 2 * PIECES * 1024;

template <int MODE, int DEPTH>      // MODE 0: LDS-DMA, 1: registers; DEPTH: slots in flight
__global__ __launch_bounds__(512) void fill(const char* src, const int* table, int n_slots, int share, unsigned* sink) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // `share` workgroups of ONE XCD (block ids are dealt to the 8 XCDs in rotation) read the same stream, as the two head blocks of a
  // token do in mla512_ps_kernel
  const int stream = (blockIdx.x & 7) + 8 * ((blockIdx.x >> 3) / share);
  const int* tab = table + (size_t)stream * n_slots;
  unsigned acc = 0;
  // page ids through LDS: a vector load of the table inside the loop would bring a vmcnt(0) that drains the ring
  __shared__ int s_tab[256];
  if (threadIdx.x < n_slots) s_tab[threadIdx.x] = tab[threadIdx.x];
  __syncthreads();
  if (wave < 2) {
    auto addr = [&](int s, int i) -> const char* {
      const int page = __builtin_amdgcn_readfirstlane(s_tab[s]);
      return src + (size_t)page * SLOT + (size_t)(wave * PIECES + i) * 1024 + lane * 16;
    };
    if constexpr (MODE == 0) {
      auto issue = [&](int s) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)addr(s, i),
                                           (__attribute__((address_space(3))) void*)(smem + (s % (DEPTH + 1)) * SLOT + (wave * PIECES + i) * 1024), 16, 0, 0);
      };
      for (int s = 0; s < DEPTH && s < n_slots; ++s) issue(s);
      for (int s = 0; s < n_slots; ++s) {
        if (s + DEPTH < n_slots) {
          issue(s + DEPTH);
          if constexpr (DEPTH == 1) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
          else if constexpr (DEPTH == 2) asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(54)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        acc += *(volatile unsigned*)(smem + (s % (DEPTH + 1)) * SLOT + wave * PIECES * 1024 + lane * 4);
      }
    } else {
      u32x4 buf[DEPTH][PIECES];
      auto issue = [&](int s, u32x4 (&b)[PIECES]) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) b[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(addr(s, i)));
      };
      auto drain = [&](int s, u32x4 (&b)[PIECES]) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) *reinterpret_cast<u32x4*>(smem + (s & 1) * SLOT + (wave * PIECES + i) * 1024 + lane * 16) = b[i];
      };
      if constexpr (DEPTH == 2) {
        issue(0, buf[0]);
        if (n_slots > 1) issue(1, buf[1]);
        for (int s = 0; s < n_slots; s += 2) {
          drain(s, buf[0]);
          if (s + 2 < n_slots) issue(s + 2, buf[0]);
          if (s + 1 < n_slots) {
            drain(s + 1, buf[1]);
            if (s + 3 < n_slots) issue(s + 3, buf[1]);
          }
          acc += *(volatile unsigned*)(smem + lane * 4);
        }
      } else {
        issue(0, buf[0]);
        for (int s = 0; s < n_slots; ++s) {
          drain(s, buf[0]);
          if (s + 1 < n_slots) issue(s + 1, buf[0]);
          acc += *(volatile unsigned*)(smem + lane * 4);
        }
      }
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
  const int wgs = 256, n_slots = 64;                      // 64 slots x 36 KiB = 2.36 MB per stream
  for (int share : {1, 2}) {
    const int streams = wgs / share;
    const size_t pages = (size_t)streams * n_slots;
    char* src; int* table; unsigned* sink;
    hipMalloc(&src, pages * SLOT + (1 << 20)); hipMemset(src, 1, pages * SLOT);
    hipMalloc(&table, pages * 4); hipMalloc(&sink, 4);
    std::vector<int> t(pages);
    for (size_t i = 0; i < pages; ++i) t[i] = (int)i;
    std::shuffle(t.begin(), t.end(), std::mt19937(1));
    hipMemcpy(table, t.data(), pages * 4, hipMemcpyHostToDevice);
    auto run = [&](auto kernel, const char* name, int lds) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kernel, dim3(wgs), dim3(512), lds, 0, src, table, n_slots, share, sink);
      hipEventRecord(e0);
      const int reps = 20;
      for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kernel, dim3(wgs), dim3(512), lds, 0, src, table, n_slots, share, sink);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double us = ms * 1e3 / reps;
      printf("share %d  %-28s %7.1f us   %5.1f GB/s per CU   %5.2f TB/s into LDS\n", share, name, us, n_slots * (double)SLOT / us / 1e3,
             wgs * n_slots * (double)SLOT / us / 1e6);
    };
    run(fill<0, 1>, "LDS-DMA, 1 slot in flight", 2 * SLOT);
    run(fill<0, 2>, "LDS-DMA, 2 slots in flight", 3 * SLOT);
    run(fill<0, 3>, "LDS-DMA, 3 slots in flight", 4 * SLOT);
    run(fill<1, 1>, "registers, 1 slot in flight", 2 * SLOT);
    run(fill<1, 2>, "registers, 2 slots in flight", 2 * SLOT);
    hipFree(src); hipFree(table); hipFree(sink);
  }
  return 0;
}
