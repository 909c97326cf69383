// Probe (round 5): is a kernel's store to hipDeviceMallocUncached memory visible to the NEXT kernel of the same stream?
// tests/test_hip_peer_virtual_ranks.py showed stale reads on uncached peer buffers inside one process (0 failures on plain hipMalloc).
//   writer kernel: buf[i] = value(iter, i)      reader kernel: counts elements != value(iter, i)
// variants: memory kind (uncached / plain), reader width (4 B / 16 B loads), a system-scope fence at the end of the writer,
// an empty kernel between writer and reader.   hipcc --offload-arch=gfx950 -O3 -o scripts/probes/build/uc_visibility_probe ...
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void writer(unsigned* buf, size_t n, unsigned iter, int fence) {
  for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] = iter * 2654435761u + (unsigned)i;
  if (fence) __atomic_thread_fence(__ATOMIC_SEQ_CST);
}
__global__ void reader4(const unsigned* buf, size_t n, unsigned iter, unsigned* bad) {
  unsigned c = 0;
  for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += buf[i] != iter * 2654435761u + (unsigned)i;
  if (c) atomicAdd(bad, c);
}
__global__ void reader16(const uint4* buf, size_t n4, unsigned iter, unsigned* bad) {
  unsigned c = 0;
  for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = buf[i];
    const unsigned b = iter * 2654435761u + (unsigned)(4 * i);
    c += (v.x != b) + (v.y != b + 1) + (v.z != b + 2) + (v.w != b + 3);
  }
  if (c) atomicAdd(bad, c);
}
__global__ void empty_kernel() {}
int main() {
  const size_t n = 1 << 20;                                 // 4 MiB
  unsigned* bad; hipMalloc(&bad, 4);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  for (int kind = 0; kind < 2; ++kind) {
    unsigned* buf = nullptr;
    hipError_t e = kind == 0 ? hipExtMallocWithFlags((void**)&buf, n * 4, hipDeviceMallocUncached) : hipMalloc((void**)&buf, n * 4);
    if (e != hipSuccess) { printf("alloc kind %d failed: %s\n", kind, hipGetErrorString(e)); continue; }
    hipMemset(buf, 0, n * 4); hipDeviceSynchronize();
    for (int wide = 0; wide < 2; ++wide)
      for (int fence = 0; fence < 2; ++fence)
        for (int gap = 0; gap < 2; ++gap) {
          unsigned total_bad = 0, bad_iters = 0;
          for (unsigned it = 1; it <= 200; ++it) {
            hipMemsetAsync(bad, 0, 4, s);
            hipLaunchKernelGGL(writer, dim3(512), dim3(256), 0, s, buf, n, it, fence);
            if (gap) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s);
            if (wide) hipLaunchKernelGGL(reader16, dim3(64), dim3(256), 0, s, (const uint4*)buf, n / 4, it, bad);
            else hipLaunchKernelGGL(reader4, dim3(64), dim3(256), 0, s, buf, n, it, bad);
            unsigned h = 0;
            hipMemcpyAsync(&h, bad, 4, hipMemcpyDeviceToHost, s);
            hipStreamSynchronize(s);
            total_bad += h; bad_iters += h != 0;
          }
          printf("%-9s reader %2d B, writer fence %d, empty kernel between %d: %3u of 200 iterations saw stale data (%u elements)\n",
                 kind == 0 ? "uncached" : "hipMalloc", wide ? 16 : 4, fence, gap, bad_iters, total_bad);
        }
    hipFree(buf);
  }
  return 0;
}
