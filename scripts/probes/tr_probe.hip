// Probe: lane->element mapping of ds_read_b64_tr_b8 / ds_read_b64_tr_b16 and of the i8 / fp8 MFMA operands.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__global__ void tr8(uint8_t* out, int stride) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint8_t)(i & 0xff);
  __syncthreads();
  // lane l supplies address l*stride within a 16-lane group block
  i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2*)(lds + threadIdx.x * stride));
  ((i32x2*)out)[threadIdx.x] = v;
}
__global__ void tr16(uint16_t* out, int stride) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)((char*)lds + threadIdx.x * stride));
  ((s16x4*)out)[threadIdx.x] = v;
}
// i8 MFMA: A[i][k] = (i==probe_row && k==probe_k), B = identity-ish -> find which lane/byte feeds which (row,k)
__global__ void mfma_i8_map(int* out) {
  // for each lane and byte j in its 16-byte A operand: set that single byte to 1, B all ones -> D[row][*] = 1 tells the row.
  // to find k: B[k][col] = k+1 for all col -> D[row][col] = k+1.
  const int lane = threadIdx.x;
  for (int sl = 0; sl < 64; ++sl)
    for (int j = 0; j < 16; ++j) {
      i32x4 a = {0, 0, 0, 0}, b;
      if (lane == sl) { int w = j / 4, s = (j % 4) * 8; a[w] = 1 << s; }
      // B operand: lane holds B[k = f(lane, jj)][col = lane&15]; we do not know f: instead set B bytes to encode (lane>>4)*16+jj + 1
      for (int w = 0; w < 4; ++w) { int val = 0; for (int bb = 0; bb < 4; ++bb) val |= (((lane >> 4) * 16 + w * 4 + bb + 1) & 0x7f) << (bb * 8); b[w] = val; }
      i32x4 c = {0, 0, 0, 0};
      c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
      // D: col = lane&15, row = (lane>>4)*4 + reg. find nonzero
      for (int r = 0; r < 4; ++r)
        if (c[r] != 0 && (lane & 15) == 0) { out[(sl * 16 + j) * 2 + 0] = (lane >> 4) * 4 + r; out[(sl * 16 + j) * 2 + 1] = c[r]; }
    }
}
int main() {
  uint8_t* d8; hipMalloc(&d8, 64 * 8);
  for (int stride : {8, 16}) {
    hipLaunchKernelGGL(tr8, dim3(1), dim3(64), 0, 0, d8, stride);
    uint8_t h[512]; hipMemcpy(h, d8, 512, hipMemcpyDeviceToHost);
    printf("tr8 stride %d (lane: 8 result bytes = LDS byte addresses)\n", stride);
    for (int l = 0; l < 32; ++l) { printf(" l%02d:", l); for (int j = 0; j < 8; ++j) printf(" %3d", h[l * 8 + j]); printf("\n"); }
  }
  uint16_t* d16; hipMalloc(&d16, 64 * 8);
  hipLaunchKernelGGL(tr16, dim3(1), dim3(64), 0, 0, d16, 8);
  uint16_t h16[256]; hipMemcpy(h16, d16, 512, hipMemcpyDeviceToHost);
  printf("tr16 stride 8 (lane: 4 results = LDS element index)\n");
  for (int l = 0; l < 20; ++l) { printf(" l%02d:", l); for (int j = 0; j < 4; ++j) printf(" %3d", h16[l * 4 + j]); printf("\n"); }
  int* dm; hipMalloc(&dm, 64 * 16 * 2 * 4); hipMemset(dm, 0xff, 64 * 16 * 2 * 4);
  hipLaunchKernelGGL(mfma_i8_map, dim3(1), dim3(64), 0, 0, dm);
  int hm[64 * 16 * 2]; hipMemcpy(hm, dm, sizeof(hm), hipMemcpyDeviceToHost);
  printf("i8 mfma A operand: lane, byte -> (row, kcode=B-lane-group*16+byte+1)\n");
  for (int sl : {0, 1, 15, 16, 17, 33, 63}) for (int j : {0, 1, 4, 15}) printf(" lane %2d byte %2d -> row %d kcode %d\n", sl, j, hm[(sl * 16 + j) * 2], hm[(sl * 16 + j) * 2 + 1]);
  return 0;
}
