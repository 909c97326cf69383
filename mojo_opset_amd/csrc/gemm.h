// Shared declarations of the grouped / dense GEMM kernels.
#pragma once
#include "common.h"

namespace mojo {

// One GEMM problem family:  C[rows of group g] = A[rows of group g] @ W[g]
//   A  [M_total, K] row-major (lda), C [M_total, N] row-major (ldc)
//   W  per group: element (k, n) lives at  W + g*w_group + k*w_k + n*w_n   (one of w_k / w_n is 1)
// Row ranges come from device-side prefix arrays (built by prefix_kernel from the row counts), so no
// host sync is needed and the launch is graph-capturable.
struct GemmArgs {
  const void* A;
  const void* W;
  void* C;
  const void* bias;          // optional [N], same dtype as C; added after rounding (golden: two ops) unless bias_fused
  int bias_fused = 0;        // 1: bias joins the fp32 accumulator, ONE rounding (F.linear semantics: the [N,K] weight layout)
  int64_t lda, ldc, w_group, w_k, w_n;
  int K, N, G;
  const int32_t* row_start;  // [G+1]   (unused when uniform_rows > 0)
  const int32_t* tile_start; // [G+1] prefix of ceil(rows_g / BM)
  int uniform_rows = 0;      // > 0: every group has exactly this many rows; the prefix arrays are not read (and no
                             // prefix kernel is launched): dense GEMMs (G = 1) and the MLA per-head projections
  // Optional row maps (rc == 0 means identity).  Logical row m of the product reads
  // A row  (m / a_rc) * a_ml + a_off + (m % a_rc) * a_mul   and writes the C row given by the c_* quadruple.
  // They let one launch consume / produce the "every rank's c-th sub-chunk" view that the chunked reduce-scatter
  // and all-gather pipelines exchange, and the head-major <-> token-major views of the MLA projections
  // (rc = tokens, ml = 1, mul = heads), without staging copies.
  int a_rc = 0, a_ml = 0, a_off = 0, a_mul = 1;
  int c_rc = 0, c_ml = 0, c_off = 0, c_mul = 1;
  // Optional split-K (dense problems with few output tiles, e.g. decode-sized M): the K range is cut into `splitk`
  // slices, slice s writes its raw fp32 / int32 accumulators to slab[s][M][N]; a finalize kernel sums the slices
  // in a fixed order (deterministic) and applies the epilogue.
  // Fused SwiGLU (MojoExperts' first projection): W has N = 2*I columns [gate | up]; an output tile pairs gate columns
  // n .. n+127 with up columns I+n .. I+n+127 and stores round(round(silu(round(gate))) * round(up)) to C [M, I] — the
  // golden's rounding points — instead of the [M, 2I] product.  256x256 MFMA kernel only, I % 128 == 0, no split-K.
  int glu = 0;
  int stage_rows = 0;        // 256x256 kernel, 16-bit C, ldc % 8 == 0, 16-byte aligned C: store whole 64-byte row pieces via LDS
  int ablate = 0;            // timing-only: 1 = skip the C stores (MOJO_HIP_GEMM_ABLATE)
  int tile_order = 0;        // 256x256 kernel (MOJO_HIP_GEMM_ORDER, read per call; measurement): 0 = every XCD walks its own run of
                             // panels (8 m x 4 n tiles per XCD at a time); 1 = the eight XCDs walk the SAME panel at the same time
                             // (64 m x 4 n tiles in flight chip-wide, an XCD takes every eighth tile); 2 = 0 with panels of 2 n-tiles
                             // (16 m x 2 n per XCD)
  int splitk = 1;
  void* slab = nullptr;
  int slab_rows = 0;
  int a_k_wrap = 0;          // 256x256 kernel: A's K-tile index wraps after this many K-tiles (0 = never): the MoE router multiplies the SAME
                             // activations by [w_hi; w_lo] stacked along K in one launch.  No K slice may straddle a multiple of it.
  int stagger_ticks = 0;     // 256x256 kernel: the first `stagger_blocks` workgroups start (block / 8 % 8) x this many 10-ns ticks late,
  int stagger_blocks = 0;    // so that the CUs do not write their tiles out in one burst per round (short-K products; gemm256_core.h)
  int defer_finalize = 0;    // split-K: leave the slabs to the caller's own finalize (launch_gemm_splitk_resnorm)
  int sk_slot = -1;          // decode-sized kernel: >= 0 = combine the K slices inside the launch (splitk_combine.h), ticket slot
};

// number of M tiles of height bm over all groups
__device__ inline int gemm_m_tiles(const GemmArgs& a, int bm) {
  return a.uniform_rows > 0 ? a.G * ((a.uniform_rows + bm - 1) / bm) : a.tile_start[a.G];
}
// group, first row and end row (exclusive) of M tile mi
__device__ inline void gemm_locate_tile(const GemmArgs& a, int mi, int bm, int& g, int& m0, int& m_end) {
  if (a.uniform_rows > 0) {
    const int tpg = (a.uniform_rows + bm - 1) / bm;
    g = mi / tpg;
    m0 = g * a.uniform_rows + (mi - g * tpg) * bm;
    m_end = (g + 1) * a.uniform_rows;
    return;
  }
  int lo = 0, hi = a.G;                      // largest g with tile_start[g] <= mi
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (a.tile_start[mid] <= mi) lo = mid; else hi = mid;
  }
  g = lo;
  m0 = a.row_start[g] + (mi - a.tile_start[g]) * bm;
  m_end = a.row_start[g + 1];
}

__host__ __device__ inline int map_row(int m, int rc, int ml, int off, int mul = 1) { return rc ? (m / rc) * ml + off + (m % rc) * mul : m; }

constexpr int GEMM_WS_INTS(int G) { return 2 * (G + 1); }

// rows [sum(counts), m_total) of C (identity row map) to be zeroed by the prefix kernel; C == nullptr: nothing
struct GemmTail {
  void* C = nullptr;
  long long ld_bytes = 0, row_bytes = 0;
};

// fills row_start / tile_start for tile height bm from int32 or int64 row counts
int launch_group_prefix(const void* counts, int counts_are_i64, int G, int bm, int64_t m_total, int32_t* row_start,
                        int32_t* tile_start, hipStream_t s, GemmTail tail = GemmTail());

// MOJO_HIP_GEMM_SKINNY: bit mask of the decode-sized (weight-stream) GEMM forms that may be taken; default all.  Without a
// form's bit its products run on the general kernels (256 x 256 tiles / separate finalize, norm and activation launches):
// the A/B switch of the tests that compare a fused or skinny form with the general path.
enum : int { SKINNY_UNIFORM = 1,   // gemm_skinny_kernel: <= 128 rows per equal-sized group, [N,K] weights (also QuantGemm's 5..128-row kernel)
             SKINNY_RAGGED = 2,    // ragged groups of <= 64 rows on average (the experts of a decode step)
             SKINNY_GLU = 4,       // SwiGLU in the skinny kernel's epilogue (mojo_hip_gemm_swiglu)
             SKINNY_RESNORM = 8,   // split-K slabs summed by the residual RMSNorm kernel (mojo_hip_gemm_residual_rmsnorm)
             SKINNY_GEMV = 16 };   // QuantGemm M <= 4: v_dot4 GEMV
inline int gemm_skinny_mask() { return static_cast<int>(MOJO_SWITCH("MOJO_HIP_GEMM_SKINNY", 31)); }

// fast MFMA path (256x256x64 tiles); returns MOJO_EUNSUPPORTED when its preconditions do not hold
int launch_gemm_mfma256(const GemmArgs& a, int dtype, int64_t m_total, hipStream_t s);
bool gemm_mfma256_glu_ok(const GemmArgs& a, int dtype);
bool gemm_skinny_ok(const GemmArgs& a, int dtype);
int gemm_skinny_splitk(int64_t m, int64_t k, int64_t n, int64_t groups);
int launch_gemm_skinny(const GemmArgs& a, int dtype, hipStream_t s);
bool gemm_rows128_takes_tile128(const GemmArgs& a, int dtype, int64_t m, int64_t k, int64_t n, int* splitk128);   // gemm_api.hip: mojo_hip_gemm's route for at most 128 rows of [N,K] weights
int launch_gemm_splitk_finalize(const GemmArgs& a, int dtype, int64_t m_total, hipStream_t s);   // C = round(sum of the fp32 K-slice slabs) (+ bias)
bool gemm_skinny_glu_ok(const GemmArgs& a, int dtype);                            // dense, <= 64 rows, W = [gate | up] rows: SwiGLU in the epilogue
int launch_gemm_skinny_glu(const GemmArgs& a, int dtype, hipStream_t s);
// split-K slabs -> round -> (+ bias) -> + residual -> RMSNorm, one row per workgroup (N <= 16384)
bool gemm_splitk_resnorm_ok(const GemmArgs& a, int dtype, const void* residual, const void* norm_w, const void* normed, const void* summed);
int launch_gemm_splitk_resnorm(const GemmArgs& a, int dtype, int64_t m_total, const void* residual, const void* norm_weight,
                               void* normed, void* summed, float eps, hipStream_t s);
bool gemm_skinny_ragged_ok(const GemmArgs& a, int dtype, int64_t m_total);        // ragged groups of <= 64 rows on average; prefix arrays for tile height 64
int launch_gemm_skinny_ragged(const GemmArgs& a, int dtype, int64_t m_total, hipStream_t s);
int launch_gemm_mfma256_f32out(const GemmArgs& a, int dtype, int accumulate, int64_t m_total, hipStream_t s);
bool gemm_mfma256_ok(const GemmArgs& a, int dtype);

// 128 x 128 tiles for one dense 16-bit product with [N,K] weights that under-fills the 256 x 256 kernel (gemm_tile128.hip)
bool gemm_tile128_ok(const GemmArgs& a, int dtype);
bool gemm_tile128_group_ok(const GemmArgs& a, int dtype);      // any number of groups (prefix arrays for 128-row tiles)
int gemm_tile128_forced();                                       // MOJO_HIP_GEMM_TILE128: 0 never, 1 always, -1 the caller's model
bool gemm_tile128_use(const GemmArgs& a, int dtype, int64_t m_total, bool model_prefers);   // MOJO_HIP_GEMM_TILE128: 1 / 0 override the model
int launch_gemm_tile128(const GemmArgs& a, int dtype, int64_t m_total, hipStream_t s);

// generic path (any dtype in {f32,f16,bf16}, any K/N, any strides)
int launch_gemm_generic(const GemmArgs& a, int dtype, int64_t m_total, hipStream_t s);

}  // namespace mojo
