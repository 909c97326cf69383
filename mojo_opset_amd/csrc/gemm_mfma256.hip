// Grouped / dense GEMM on MFMA for gfx950: 256x256 output tile, BK = 64, 8 waves (2 x 4), bf16/fp16
// inputs, fp32 accumulation.
//
// Structure (after the "256^2 8-phase" recipe of the CDNA4 programming guide, re-derived here):
//   * LDS = 2 K-tile buffers x 4 half-tiles (A rows 0-127 / 128-255, W cols 0-127 / 128-255) x 16 KiB.
//   * global -> LDS by direct-to-LDS loads (global_load_lds_dwordx4, 1 KiB per wave instruction).  The
//     LDS image is lane-linear, so bank-conflict avoidance lives in the per-lane SOURCE address:
//       - K-major operands (A always; W when stored [N,K]) use 16-row x 32-k sub-tiles (1 KiB) with the
//         two 32-byte halves of rows 8-15 swapped ("st_16x32"); a fragment is one ds_read_b128 per lane.
//       - W stored [K,N] uses an [k/8][n/16][8][16] image with k-rows 0-3 <-> 4-7 swapped in odd k-blocks;
//         fragments come from ds_read_b64_tr_b16 (hardware transpose), one glds = 8 k-rows x 128 B.
//   * a K-tile is 4 phases; each phase = {issue one half-tile of a future K-tile, read this phase's
//     fragments, 16 MFMAs of one 64x32 C-quadrant per wave, one barrier}.  Loads are retired with a
//     COUNTED s_waitcnt vmcnt(6) once per K-tile, so three half-tiles (48 KiB per CU) stay in flight
//     across barriers; the wait sits before the last barrier of a K-tile and the data is first read in
//     the next phase.
//   * MFMA operands are swapped (W fragment as "A", A fragment as "B") so that a lane owns 4 consecutive
//     output columns of one row -> 8-byte stores.
//   * blocks are mapped to tiles through an XCD-aware, panel-major order: the 32 blocks that run
//     concurrently on one XCD cover 8 m-tiles x 4 n-tiles and share their operands in that XCD's L2.
//
// Algorithmic FLOPs per launch: 2 * M * K * N.   Bound: MFMA (bf16 dense peak ~2.5 PFLOP/s).
#include <stdlib.h>

#include "gemm256_core.h"
#ifdef MOJO_HIP_BUILD_EXPERIMENTS
#include "experiments/gemm_w128.h"
#endif

namespace mojo {

bool gemm_mfma256_ok(const GemmArgs& a, int dtype) {
  if (dtype != MOJO_BF16 && dtype != MOJO_F16) return false;
  if (a.ldc % 4 != 0 || !aligned_to(a.C, 8)) return false;
  return g256::gemm256_layout_ok(a, 2);
}

bool gemm_mfma256_glu_ok(const GemmArgs& a, int dtype) {
  return gemm_mfma256_ok(a, dtype) && a.N % 256 == 0 && a.splitk == 1 && a.bias == nullptr;
}

int launch_gemm_mfma256(const GemmArgs& a_in, int dtype, int64_t m_total, hipStream_t s) {
  GemmArgs a = a_in;
  MOJO_REQUIRE(gemm_mfma256_ok(a, dtype), MOJO_EUNSUPPORTED, "gemm_mfma256: preconditions not met");
  const bool no_stage = MOJO_SWITCH("MOJO_HIP_GEMM_STAGE_ROWS", 1) == 0;
  a.stage_rows = (!no_stage && a.splitk == 1 && a.ldc % 8 == 0 && aligned_to(a.C, 16)) ? 1 : 0;
  MOJO_REQUIRE(!a.glu || gemm_mfma256_glu_ok(a, dtype), MOJO_EUNSUPPORTED, "gemm_mfma256: fused SwiGLU needs I %% 128 == 0");
#ifdef MOJO_HIP_BUILD_EXPERIMENTS
  {  // experiments build only, MOJO_HIP_GEMM_W128=1: four waves of 128x128 (experiments/gemm_w128.h), [N,K] weights
    const long long w128 = MOJO_SWITCH("MOJO_HIP_GEMM_W128", 0);
    const char e[2] = {static_cast<char>('0' + (w128 >= 0 && w128 <= 9 ? w128 : 0)), 0};
    if (e[0] >= '1' && e[0] <= '5' && dtype == MOJO_BF16 && a.w_k == 1 && !a.glu && a.splitk == 1 && a.K % 32 == 0 && a.K >= 128) {
      g256::EpiloguePlain<bf16_t> epi{static_cast<bf16_t*>(a.C), a.ldc, static_cast<const bf16_t*>(a.bias), a.bias_fused != 0};
      if (e[0] == '2') return w128::gemm_w128_launch<g256::PolBF16, g256::EpiloguePlain<bf16_t>, 1>(a, epi, m_total, s);   // 2-4: timing-only ablations
      if (e[0] == '3') return w128::gemm_w128_launch<g256::PolBF16, g256::EpiloguePlain<bf16_t>, 2>(a, epi, m_total, s);
      if (e[0] == '4') return w128::gemm_w128_launch<g256::PolBF16, g256::EpiloguePlain<bf16_t>, 3>(a, epi, m_total, s);
      if (e[0] == '5') return w128::gemm_w128_launch<g256::PolBF16, g256::EpiloguePlain<bf16_t>, 0, 3>(a, epi, m_total, s);   // shallower prefetch
      return w128::gemm_w128_launch<g256::PolBF16>(a, epi, m_total, s);
    }
  }
#endif
  if (dtype == MOJO_BF16) {
    g256::EpiloguePlain<bf16_t> epi{static_cast<bf16_t*>(a.C), a.ldc, static_cast<const bf16_t*>(a.bias), a.bias_fused != 0};
    return g256::gemm256_launch<g256::PolBF16, g256::EpiloguePlain<bf16_t>, true>(a, epi, m_total, s);
  }
  g256::EpiloguePlain<f16_t> epi{static_cast<f16_t*>(a.C), a.ldc, static_cast<const f16_t*>(a.bias), a.bias_fused != 0};
  return g256::gemm256_launch<g256::PolF16, g256::EpiloguePlain<f16_t>, true>(a, epi, m_total, s);
}

// 16-bit operands, fp32 output (optionally accumulated onto C): used by the MoE router for its hi/lo split product
int launch_gemm_mfma256_f32out(const GemmArgs& a, int dtype, int accumulate, int64_t m_total, hipStream_t s) {
  MOJO_REQUIRE((dtype == MOJO_BF16 || dtype == MOJO_F16) && a.ldc % 4 == 0 && aligned_to(a.C, 16) && g256::gemm256_layout_ok(a, 2),
               MOJO_EUNSUPPORTED, "gemm_mfma256_f32out: preconditions not met");
  g256::EpilogueF32 epi{static_cast<float*>(a.C), a.ldc, accumulate};
  if (dtype == MOJO_BF16) return g256::gemm256_launch<g256::PolBF16>(a, epi, m_total, s);
  return g256::gemm256_launch<g256::PolF16>(a, epi, m_total, s);
}

}  // namespace mojo

#ifdef GEMM_STAMPS
extern "C" int mojo_hip_debug_gemm_stamps(unsigned* host_out, int64_t count) {
  if (hipDeviceSynchronize() != hipSuccess) return MOJO_ELAUNCH;
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mojo::g256::g_gemm_stamps), static_cast<size_t>(count) * 4) == hipSuccess ? MOJO_OK : MOJO_ELAUNCH;
}
#endif
