// 128-row-tile GEMM, 16-bit instances (bf16 / fp16, plain epilogue, both weight layouts): see gemm_tile128_core.h.
#include "gemm_tile128_core.h"

namespace mojo {

// 16-bit products, whole K-tiles; [N,K] weights or [K,N] weights with N a multiple of 8; any number of groups (ragged groups: the
// caller builds the prefix arrays for 128-row tiles); one dense product may be cut along K (fp32 slabs + launch_gemm_splitk_finalize).
bool gemm_tile128_group_ok(const GemmArgs& a, int dtype) {
  if (dtype != MOJO_BF16 && dtype != MOJO_F16) return false;
  const bool layout = (a.w_k == 1 && a.w_n % 8 == 0) || (a.w_n == 1 && a.w_k % 8 == 0 && a.N % 8 == 0 && a.N >= 8);
  return a.G >= 1 && layout && (a.G == 1 || a.w_group % 8 == 0) && a.K >= 64 && a.K % 64 == 0 && a.lda % 8 == 0 && a.ldc % 4 == 0 &&
         (a.splitk == 1 || (a.splitk > 1 && a.G == 1 && a.uniform_rows > 0 && a.slab && a.N % 4 == 0 && a.K / 64 >= a.splitk)) && !a.glu && a.a_k_wrap == 0 && aligned_to(a.A, 16) && aligned_to(a.W, 16) && aligned_to(a.C, 8) &&
         (!a.bias || aligned_to(a.bias, 2));
}
// One dense product (G = 1).
bool gemm_tile128_ok(const GemmArgs& a, int dtype) { return a.G == 1 && a.uniform_rows > 0 && gemm_tile128_group_ok(a, dtype); }
int gemm_tile128_forced() { return g128::forced_choice(); }      // MOJO_HIP_GEMM_TILE128: 0 never, 1 always, -1 the caller's model

// Where the 128-row tiles are taken: where the caller's time model (gemm_api.hip: gemm_dense_prefers_tile128, and
// gemm_rows128_prefers_tile128 against the weight-streaming kernels) says so; MOJO_HIP_GEMM_TILE128 = 1 / 0: wherever they apply / never.
bool gemm_tile128_use(const GemmArgs& a, int dtype, int64_t m_total, bool model_prefers) {
  if (!gemm_tile128_ok(a, dtype)) return false;
  const int f = g128::forced_choice();
  return f < 0 ? model_prefers : f == 1;
}

int launch_gemm_tile128(const GemmArgs& a, int dtype, int64_t m_total, hipStream_t s) {
  MOJO_REQUIRE(gemm_tile128_group_ok(a, dtype), MOJO_EUNSUPPORTED, "gemm(128-row tiles): preconditions not met");
  int rc;
  if (dtype == MOJO_BF16) {
    g256::EpiloguePlain<bf16_t> epi{static_cast<bf16_t*>(a.C), a.ldc, static_cast<const bf16_t*>(a.bias), a.bias_fused != 0};
    rc = g128::launch<g256::PolBF16>(a, epi, m_total, s);
  } else {
    g256::EpiloguePlain<f16_t> epi{static_cast<f16_t*>(a.C), a.ldc, static_cast<const f16_t*>(a.bias), a.bias_fused != 0};
    rc = g128::launch<g256::PolF16>(a, epi, m_total, s);
  }
  if (rc || a.splitk == 1 || a.defer_finalize) return rc;
  return launch_gemm_splitk_finalize(a, dtype, m_total, s);               // C = round(sum of the slices' slabs) (+ bias)
}

}  // namespace mojo
