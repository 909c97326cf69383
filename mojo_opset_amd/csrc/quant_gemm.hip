// MojoQuantGemm: out[M,N] = round_out( (A_q @ W_q) * input_scale[m] * weight_scale[n] ).
//   int8 : exact int32 accumulation on v_mfma_i32_16x16x64_i8 (the golden's fp32 sum of int products is
//          exact up to 2^24), fp32 scaling in the golden's order, one rounding.
//   fp8  : OCP e4m3 on v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales (2x the bf16 MFMA rate);
//          this dtype has no reference implementation — parity unpinned (checked against oracle/ only).
// Shapes outside the MFMA kernel's preconditions run on a plain one-output-per-thread kernel.
//
// Algorithmic FLOPs: 2*M*K*N (MFMA-bound for M >= 128); bytes: K*N (weight stream) + M*K + M*N*out.
#include <stdlib.h>

#include "gemm256_core.h"

namespace mojo {

__device__ __forceinline__ float fp8_to_f(uint8_t v) { return __builtin_amdgcn_cvt_f32_fp8(static_cast<int>(v), 0); }

template <typename TO, bool FP8>
__global__ __launch_bounds__(256) void quant_gemm_generic_kernel(const uint8_t* __restrict__ A, const uint8_t* __restrict__ W,
                                                                 const float* __restrict__ rs, const bf16_t* __restrict__ cs,
                                                                 TO* __restrict__ C, int64_t M, int K, int N, int64_t w_k,
                                                                 int64_t w_n) {
  const int64_t total = M * N;
  for (int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
       idx += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t m = idx / N;
    const int n = static_cast<int>(idx - m * N);
    const uint8_t* a = A + m * K;
    const uint8_t* w = W + static_cast<int64_t>(n) * w_n;
    float v;
    if constexpr (FP8) {
      float acc = 0.f;
      for (int k = 0; k < K; ++k) acc = fmaf(fp8_to_f(a[k]), fp8_to_f(w[k * w_k]), acc);
      v = acc;
    } else {
      int acc = 0;
      for (int k = 0; k < K; ++k) acc += static_cast<int>(static_cast<int8_t>(a[k])) * static_cast<int>(static_cast<int8_t>(w[k * w_k]));
      v = static_cast<float>(acc);
    }
    v = __fmul_rn(__fmul_rn(v, rs[m]), static_cast<float>(cs[n]));
    asm volatile("" : "+v"(v));                      // no mul+narrow fusion: the golden rounds twice
    C[m * N + n] = elt<TO>::from_f(v);
  }
}

// split-K finalize: out[m][n] = round_TO( (sum_s slab[s][m][n]) * rs[m] * cs[n] ), slices summed in index order
template <typename TO, typename ACC>
__global__ __launch_bounds__(256) void quant_finalize_kernel(const ACC* __restrict__ slab, int splitk, int64_t M, int N,
                                                             const float* __restrict__ rs, const bf16_t* __restrict__ cs,
                                                             TO* __restrict__ C) {
  const int64_t total = M * N;
  for (int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
       idx += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    ACC acc = 0;
    for (int sidx = 0; sidx < splitk; ++sidx) acc += slab[static_cast<int64_t>(sidx) * total + idx];
    const int64_t m = idx / N;
    const int n = static_cast<int>(idx - m * N);
    float v = __fmul_rn(__fmul_rn(static_cast<float>(acc), rs[m]), static_cast<float>(cs[n]));
    asm volatile("" : "+v"(v));
    C[idx] = elt<TO>::from_f(v);
  }
}

// Few output tiles (decode-sized M): cut K so that ~256 workgroups stream the weight concurrently.
static int quant_splitk(int64_t m, int k, int n) {
  if (const char* e = getenv("MOJO_HIP_QGEMM_SPLITK")) { const int v = atoi(e); if (v >= 1) return v; }
  const int64_t tiles = ceil_div(m, 256) * ceil_div(n, 256);
  const int nkt = k / 128;
  if (tiles >= 128 || n % 4 != 0 || nkt < 2) return 1;
  int64_t sk = ceil_div(256, tiles);
  if (sk > nkt / 2) sk = nkt / 2;                       // at least two K-tiles per slice
  if (sk > 32) sk = 32;
  return sk < 1 ? 1 : static_cast<int>(sk);
}

template <typename TO>
static int run_quant(GemmArgs a, const float* rs, const bf16_t* cs, int64_t m, int quant_dtype, void* slab_ws,
                     hipStream_t s) {
  const bool fp8 = quant_dtype == MOJO_F8E4M3;
  if (g256::gemm256_layout_ok(a, 1)) {
    const int sk = quant_splitk(m, a.K, a.N);
    if (sk > 1) {
      a.splitk = sk; a.slab = slab_ws; a.slab_rows = static_cast<int>(m);
      int rc;
      if (fp8) {
        g256::EpilogueDequant<TO, f32x4> epi{static_cast<TO*>(a.C), a.ldc, rs, cs, 0.f};
        rc = g256::gemm256_launch<g256::PolF8>(a, epi, m, s);
      } else {
        g256::EpilogueDequant<TO, i32x4> epi{static_cast<TO*>(a.C), a.ldc, rs, cs, 0.f};
        rc = g256::gemm256_launch<g256::PolI8>(a, epi, m, s);
      }
      if (rc) return rc;
      int64_t blocks = ceil_div(m * a.N, 256);
      if (blocks > 256 * 8) blocks = 256 * 8;
      if (fp8)
        hipLaunchKernelGGL((quant_finalize_kernel<TO, float>), dim3(blocks), dim3(256), 0, s, static_cast<const float*>(slab_ws), sk, m, a.N, rs, cs, static_cast<TO*>(a.C));
      else
        hipLaunchKernelGGL((quant_finalize_kernel<TO, int>), dim3(blocks), dim3(256), 0, s, static_cast<const int*>(slab_ws), sk, m, a.N, rs, cs, static_cast<TO*>(a.C));
      MOJO_CHECK_LAUNCH("quant_gemm(finalize)");
      return MOJO_OK;
    }
    if (fp8) {
      g256::EpilogueDequant<TO, f32x4> epi{static_cast<TO*>(a.C), a.ldc, rs, cs, 0.f};
      return g256::gemm256_launch<g256::PolF8>(a, epi, m, s);
    }
    g256::EpilogueDequant<TO, i32x4> epi{static_cast<TO*>(a.C), a.ldc, rs, cs, 0.f};
    return g256::gemm256_launch<g256::PolI8>(a, epi, m, s);
  }
  int64_t blocks = ceil_div(m * a.N, 256);
  if (blocks > 256 * 32) blocks = 256 * 32;
  const uint8_t* A = static_cast<const uint8_t*>(a.A);
  const uint8_t* W = static_cast<const uint8_t*>(a.W);
  if (fp8)
    hipLaunchKernelGGL((quant_gemm_generic_kernel<TO, true>), dim3(blocks), dim3(256), 0, s, A, W, rs, cs,
                       static_cast<TO*>(a.C), m, a.K, a.N, a.w_k, a.w_n);
  else
    hipLaunchKernelGGL((quant_gemm_generic_kernel<TO, false>), dim3(blocks), dim3(256), 0, s, A, W, rs, cs,
                       static_cast<TO*>(a.C), m, a.K, a.N, a.w_k, a.w_n);
  MOJO_CHECK_LAUNCH("quant_gemm_generic");
  return MOJO_OK;
}

}  // namespace mojo

using namespace mojo;

extern "C" int64_t mojo_hip_quant_gemm_workspace_bytes(int64_t m, int64_t k, int64_t n) {
  const int sk = (k > 0 && k % 128 == 0) ? quant_splitk(m, static_cast<int>(k), static_cast<int>(n)) : 1;
  return 64 + (sk > 1 ? static_cast<int64_t>(sk) * m * n * 4 : 0);
}

extern "C" int mojo_hip_quant_gemm(const void* input, const void* weight, const float* input_scale,
                                   const void* weight_scale, void* out, int64_t m, int64_t k, int64_t n,
                                   int trans_weight, int quant_dtype, int out_dtype, void* workspace,
                                   int64_t workspace_bytes, mojo_stream_t stream) {
  MOJO_REQUIRE(k > 0 && n > 0 && m >= 0, MOJO_EINVAL, "quant_gemm: bad shape");
  if (m == 0) return MOJO_OK;
  MOJO_REQUIRE(input && weight && input_scale && weight_scale && out, MOJO_EINVAL, "quant_gemm: null pointer");
  MOJO_REQUIRE(quant_dtype == MOJO_I8 || quant_dtype == MOJO_F8E4M3, MOJO_EUNSUPPORTED,
               "quant_gemm: quant dtype %d (int8 / fp8-e4m3 only)", quant_dtype);
  MOJO_REQUIRE(out_dtype == MOJO_F32 || out_dtype == MOJO_F16 || out_dtype == MOJO_BF16, MOJO_EUNSUPPORTED,
               "quant_gemm: output dtype %d", out_dtype);
  MOJO_REQUIRE(m < (1LL << 31) && k < (1LL << 31) && n < (1LL << 31), MOJO_EUNSUPPORTED, "quant_gemm: dimension too large");
  MOJO_REQUIRE(workspace && workspace_bytes >= mojo_hip_quant_gemm_workspace_bytes(m, k, n) && aligned_to(workspace, 16),
               MOJO_EWORKSPACE, "quant_gemm: workspace too small");
  GemmArgs a;
  a.A = input; a.W = weight; a.C = out; a.bias = nullptr;
  a.lda = k; a.ldc = n; a.w_group = 0;
  if (trans_weight) { a.w_k = 1; a.w_n = k; } else { a.w_k = n; a.w_n = 1; }
  a.K = static_cast<int>(k); a.N = static_cast<int>(n); a.G = 1;
  int32_t* ws = static_cast<int32_t*>(workspace);
  a.row_start = ws; a.tile_start = ws + 2;
  hipStream_t s = static_cast<hipStream_t>(stream);
  a.uniform_rows = static_cast<int>(m);
  const bf16_t* cs = static_cast<const bf16_t*>(weight_scale);
  void* slab_ws = static_cast<char*>(workspace) + 64;
  switch (out_dtype) {
    case MOJO_F32: return run_quant<float>(a, input_scale, cs, m, quant_dtype, slab_ws, s);
    case MOJO_F16: return run_quant<f16_t>(a, input_scale, cs, m, quant_dtype, slab_ws, s);
    default: return run_quant<bf16_t>(a, input_scale, cs, m, quant_dtype, slab_ws, s);
  }
}
