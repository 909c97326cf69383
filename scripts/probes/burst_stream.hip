// Probe: HBM streaming rate of a [rows][K bytes] weight matrix read by 256 workgroups x 4 waves, each wave owning 16 rows,
// as a function of the contiguous run one wave-instruction reads per row: 256 B (4 rows per instruction, the skinny GEMM's
// pattern), 512 B (2 rows) or 1 KiB (1 row).  Same bytes in flight per wave (16 KiB) in every variant.
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/burst_stream.hip -o scripts/probes/build/burst_stream
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-result"
#include <stdio.h>
#include <stdint.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// RUN = bytes per row per instruction (256, 512, 1024).  A wave reads 16 rows x 1 KiB per step = 16 instructions.
template <int RUN, bool NT>
__global__ __launch_bounds__(256) void stream_kernel(const uint8_t* __restrict__ w, int64_t row_bytes, int k_bytes_per_wg, int slices, unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.x / slices, slice = blockIdx.x % slices;
  const int row0 = tile * 64 + wave * 16;
  constexpr int LANES_PER_ROW = RUN / 16, ROWS_PER_INST = 64 / LANES_PER_ROW, INST_PER_KB = 1024 / RUN;   // per 16 rows x 1 KiB
  const uint8_t* base = w + static_cast<int64_t>(row0 + lane / LANES_PER_ROW) * row_bytes + static_cast<int64_t>(slice) * k_bytes_per_wg + (lane % LANES_PER_ROW) * 16;
  u32x4 acc = {0, 0, 0, 0};
  for (int k = 0; k < k_bytes_per_wg; k += 1024) {
    u32x4 v[16];
#pragma unroll
    for (int rg = 0; rg < 16 / ROWS_PER_INST; ++rg)          // row groups of the wave's 16 rows
#pragma unroll
      for (int c = 0; c < INST_PER_KB; ++c) {
        const u32x4* p = reinterpret_cast<const u32x4*>(base + static_cast<int64_t>(rg * ROWS_PER_INST) * row_bytes + k + c * RUN);
        v[rg * INST_PER_KB + c] = NT ? __builtin_nontemporal_load(p) : *p;
      }
#pragma unroll
    for (int i = 0; i < 16; ++i) acc ^= v[i];
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

template <int RUN, bool NT>
static void run(const uint8_t* w, int rows, int64_t row_bytes, int slices, unsigned* sink) {
  const int tiles = rows / 64;
  const int k_per = static_cast<int>(row_bytes / slices);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream_kernel<RUN, NT>), dim3(tiles * slices), dim3(256), 0, 0, w, row_bytes, k_per, slices, sink);
  hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream_kernel<RUN, NT>), dim3(tiles * slices), dim3(256), 0, 0, w, row_bytes, k_per, slices, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = static_cast<double>(rows) * row_bytes;
  printf("rows %6d x %6lld B, %3d workgroups, run %4d B%s: %7.1f us  %6.2f TB/s\n", rows, (long long)row_bytes, tiles * slices, RUN, NT ? " nt" : "   ",
         ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
}

int main() {
  const int64_t total = 512ll << 20;
  uint8_t* w; unsigned* sink;
  hipMalloc(&w, total); hipMalloc(&sink, 4); hipMemset(w, 1, total);
  // 8192 x 16 KiB (bf16 8192 x 8192): 128 column tiles x 2 K slices = 256 workgroups.   Two matrices alternate? one 134 MB matrix
  // re-read every launch sits in the 256 MiB Infinity Cache, so read a 512 MiB buffer: 32768 rows x 16 KiB, 512 tiles
  run<256, true>(w, 32768, 16384, 1, sink);  run<512, true>(w, 32768, 16384, 1, sink);  run<1024, true>(w, 32768, 16384, 1, sink);
  run<256, false>(w, 32768, 16384, 1, sink); run<1024, false>(w, 32768, 16384, 1, sink);
  // the GEMM's shape: few column tiles, K split over workgroups (7168 rows x 18432 B: 112 tiles x 2 slices), buffer rotated by the caller? single 132 MB matrix
  run<256, true>(w, 7168, 18432, 2, sink);   run<1024, true>(w, 7168, 18432, 2, sink);
  run<256, true>(w, 8192, 16384, 2, sink);   run<1024, true>(w, 8192, 16384, 2, sink);
  return 0;
}
