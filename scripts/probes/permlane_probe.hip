// probe: what do v_permlane32_swap (asm and builtin forms) leave in each operand?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  const int lane = threadIdx.x;
  int x = lane, y = 100 + lane;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
  out[lane] = x; out[64 + lane] = y;
  const auto r = __builtin_amdgcn_permlane32_swap(static_cast<unsigned>(lane), static_cast<unsigned>(100 + lane), false, false);
  out[128 + lane] = r[0]; out[192 + lane] = r[1];
}
int main() {
  int* d; hipMalloc(&d, 256 * 4);
  k<<<1, 64>>>(d);
  int h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* names[4] = {"asm x'", "asm y'", "bi r0 ", "bi r1 "};
  for (int v = 0; v < 4; ++v) { printf("%s:", names[v]); for (int l = 0; l < 64; l += 8) printf(" %d", h[64 * v + l]); printf("\n"); }
  return 0;
}
