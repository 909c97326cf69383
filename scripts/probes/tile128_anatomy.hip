// Round 5: anatomy of the 128 x 128-tile GEMM's K loop (csrc/gemm_tile128.hip).  The kernel file is compiled INTO this probe with
// -DT128_ABLATE=<n>: 0 = the kernel as shipped, 2 = no LDS-DMA inside the loop (fragments + MFMAs on stale LDS: the compute loop
// alone), 3 = LDS-DMA, waits and barriers only (no fragment reads, no MFMAs: the fill alone), 4 = fill + fragment reads, no MFMAs.
// Build (one binary per variant), link against the library for the error / switch plumbing:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DT128_ABLATE=2 -Iinclude -Imojo_opset_amd/csrc scripts/probes/tile128_anatomy.hip \
//         mojo_opset_amd/csrc/gemm_tile128.hip -Lmojo_opset_amd/lib -lmojo_hip -Wl,-rpath,'$ORIGIN/../../../mojo_opset_amd/lib' -o scripts/probes/_bin/t128_a2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "gemm.h"
#ifndef T128_PIECE
#define T128_PIECE 1
#endif
#ifndef T128_SCHED
#define T128_SCHED 1
#endif

int main(int argc, char** argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 4096;
  const int ms[] = {128, 256, 512, 1024, 2048, 4096};
  for (int m : ms) {
    std::vector<uint16_t> ha(static_cast<size_t>(m) * K), hw(static_cast<size_t>(N) * K);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return static_cast<uint16_t>(0x3c00u + ((s >> 9) & 0x3ffu) - ((s >> 20) & 1u) * 0x8000u); };  // ~ +-[0.5, 2) bf16-ish bit patterns
    for (auto& v : ha) v = rnd();
    for (auto& v : hw) v = rnd();
    void *A, *W, *C;
    hipMalloc(&A, ha.size() * 2); hipMalloc(&W, hw.size() * 2); hipMalloc(&C, static_cast<size_t>(m) * N * 2);
    hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    mojo::GemmArgs a;
    a.A = A; a.W = W; a.C = C; a.bias = nullptr; a.lda = K; a.ldc = N; a.w_group = 0; a.w_k = 1; a.w_n = K;
    a.K = K; a.N = N; a.G = 1; a.uniform_rows = m; a.row_start = nullptr; a.tile_start = nullptr;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) mojo::launch_gemm_tile128(a, MOJO_BF16, m, nullptr);
    hipDeviceSynchronize();
    float best = 1e9f, sum = 0;
    const int reps = 5, iters = 20;
    for (int r = 0; r < reps; ++r) {
      hipEventRecord(e0, nullptr);
      for (int i = 0; i < iters; ++i) mojo::launch_gemm_tile128(a, MOJO_BF16, m, nullptr);
      hipEventRecord(e1, nullptr);
      hipEventSynchronize(e1);
      float ms_ = 0; hipEventElapsedTime(&ms_, e0, e1);
      best = ms_ < best ? ms_ : best; sum += ms_;
    }
    const double us = best * 1e3 / iters;
    printf("shape %s ablate %d piece %d sched %d K %d N %d M %5d tiles %4d : %7.1f us  (%.3f us per K-tile, %6.0f TFLOP/s)\n", getenv("MOJO_HIP_GEMM_TILE128") ? getenv("MOJO_HIP_GEMM_TILE128") : "auto", T128_ABLATE, T128_PIECE, T128_SCHED, K, N, m,
           ((m + 127) / 128) * ((N + 127) / 128), us, us / (K / 64), 2.0 * m * K * N / us / 1e6);
    hipFree(A); hipFree(W); hipFree(C);
  }
  return 0;
}
