// C ABI of the grouped and dense GEMM entry points: argument checks, kernel choice, prefix launch (ragged groups only).
#include <stdlib.h>

#include <algorithm>

#include "gemm.h"

namespace mojo {

static int run_gemm(GemmArgs& a, int dtype, int64_t m_total, hipStream_t s) {
#ifdef MOJO_HIP_BUILD_EXPERIMENTS              // timing-only ablations and the round-3 tile-order experiment (DESIGN Appendix A 7)
  a.ablate = static_cast<int>(MOJO_SWITCH("MOJO_HIP_GEMM_ABLATE", 0));
  a.tile_order = static_cast<int>(MOJO_SWITCH("MOJO_HIP_GEMM_ORDER", 0));
#endif
  if (gemm_skinny_ok(a, dtype)) return launch_gemm_skinny(a, dtype, s);      // <= 128 rows per (equal-sized) group, [N,K] weights
  if (gemm_skinny_ragged_ok(a, dtype, m_total)) return launch_gemm_skinny_ragged(a, dtype, m_total, s);   // ragged, <= 64 rows per group on average
  if (gemm_mfma256_ok(a, dtype)) return launch_gemm_mfma256(a, dtype, m_total, s);
  return launch_gemm_generic(a, dtype, m_total, s);
}

}  // namespace mojo

using namespace mojo;

extern "C" int64_t mojo_hip_group_gemm_workspace_bytes(int64_t num_groups) {
  return (2 * (num_groups + 1)) * static_cast<int64_t>(sizeof(int32_t)) + 64;
}

// 128-row tiles or the 256 x 256 kernel for a grouped product?  The row counts live on the device: a model on the MEAN rows per
// group (constants fitted on scripts/probes/group_gemm_sweep.py, both kernels forced, 116 grouped products:
// profiles/r5_group_gemm_sweep.txt — within 0.9 % of always picking the faster one, 15 % ahead of the 256 x 256 kernel alone).
static bool group_prefers_tile128(int64_t m_total, int64_t k, int64_t n, int64_t num_groups) {
  const int64_t rows = std::max<int64_t>(1, m_total / num_groups), nkt = k / 64;
  const int64_t t256 = num_groups * ceil_div(rows, 256) * ceil_div(n, 256);
  const int64_t t128n = num_groups * ceil_div(rows, 128) * ceil_div(n, 128), t128w = num_groups * ceil_div(rows, 128) * ceil_div(n, 256);
  const double c256 = static_cast<double>(ceil_div(t256, 256)) * nkt * (t256 <= 128 ? 1.2 : 1.45);
  const double c128 = t128n <= 256 ? 5.0 + nkt * (0.30 + 0.125 * t128n / 256.0) : 6.0 + nkt * 1.0 * std::max(1.0, t128w / 256.0);
  return c128 < c256;
}

extern "C" int mojo_hip_group_gemm_strided(const void* input, const void* weight, void* out, const void* group_list,
                                           int group_list_is_i64, int64_t m_total, int64_t k, int64_t n,
                                           int64_t num_groups, int64_t lda, int64_t ldc, int64_t w_group_stride,
                                           int64_t w_k_stride, int64_t w_n_stride, const int64_t a_map[4],
                                           const int64_t c_map[4], int dtype, void* workspace,
                                           int64_t workspace_bytes, mojo_stream_t stream) {
  MOJO_REQUIRE(num_groups > 0 && k > 0 && n > 0 && m_total >= 0, MOJO_EINVAL, "group_gemm: bad shape");
  if (m_total == 0) return MOJO_OK;
  MOJO_REQUIRE(input && weight && out, MOJO_EINVAL, "group_gemm: null pointer");
  MOJO_REQUIRE(group_list || m_total % num_groups == 0, MOJO_EINVAL,
               "group_gemm: group_list == NULL means equal groups, but %lld rows do not divide into %lld groups",
               (long long)m_total, (long long)num_groups);
  MOJO_REQUIRE(dtype == MOJO_F32 || dtype == MOJO_F16 || dtype == MOJO_BF16, MOJO_EUNSUPPORTED,
               "group_gemm: dtype %d not supported", dtype);
  MOJO_REQUIRE(m_total < (1LL << 31) && k < (1LL << 31) && n < (1LL << 31) && num_groups < (1 << 20), MOJO_EUNSUPPORTED,
               "group_gemm: dimension too large");
  MOJO_REQUIRE(lda >= k && ldc >= n && (w_k_stride == 1 || w_n_stride == 1), MOJO_EINVAL, "group_gemm: bad strides");
  MOJO_REQUIRE(workspace && workspace_bytes >= mojo_hip_group_gemm_workspace_bytes(num_groups) && aligned_to(workspace, 4),
               MOJO_EWORKSPACE, "group_gemm: workspace too small");
  GemmArgs a;
  a.A = input; a.W = weight; a.C = out; a.bias = nullptr;
  a.lda = lda; a.ldc = ldc; a.w_group = w_group_stride; a.w_k = w_k_stride; a.w_n = w_n_stride;
  a.K = static_cast<int>(k); a.N = static_cast<int>(n); a.G = static_cast<int>(num_groups);
  if (a_map) { a.a_rc = static_cast<int>(a_map[0]); a.a_ml = static_cast<int>(a_map[1]); a.a_off = static_cast<int>(a_map[2]); a.a_mul = static_cast<int>(a_map[3]); }
  if (c_map) { a.c_rc = static_cast<int>(c_map[0]); a.c_ml = static_cast<int>(c_map[1]); a.c_off = static_cast<int>(c_map[2]); a.c_mul = static_cast<int>(c_map[3]); }
  int32_t* ws = static_cast<int32_t*>(workspace);
  a.row_start = ws; a.tile_start = ws + (num_groups + 1);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!group_list) a.uniform_rows = static_cast<int>(m_total / num_groups);
  // 128-row tiles (gemm_tile128_core.h) for grouped products the 256 x 256 kernel under-fills or pads: few small experts
  // (8 groups of 2048 x 1408: 48 tiles of 256 x 256, 45 us where the weights stream in 8), or 65-250 rows per group (a
  // 256-row tile per group is mostly padding).  The row counts live on the device: the model works on the mean.
  bool use128 = false;
  if (num_groups > 1 && gemm_tile128_group_ok(a, dtype) &&
      !(group_list ? gemm_skinny_ragged_ok(a, dtype, m_total) : gemm_skinny_ok(a, dtype))) {
    const int f = gemm_tile128_forced();
    use128 = f >= 0 ? f == 1 : group_prefers_tile128(m_total, k, n, num_groups);
  }
  if (group_list) {
    const int bm = use128 ? 128 : gemm_skinny_ragged_ok(a, dtype, m_total) ? 64 : (gemm_mfma256_ok(a, dtype) ? 256 : 64);   // the tile height the chosen kernel walks
    GemmTail tail;
    const int64_t elt = dtype == MOJO_F32 ? 4 : 2;
    if (!c_map) { tail.C = out; tail.ld_bytes = ldc * elt; tail.row_bytes = n * elt; }
    int rc = launch_group_prefix(group_list, group_list_is_i64, a.G, bm, m_total, ws, ws + (num_groups + 1), s, tail);
    if (rc) return rc;
  }
  if (use128) return launch_gemm_tile128(a, dtype, m_total, s);
  return run_gemm(a, dtype, m_total, s);
}

extern "C" int mojo_hip_group_gemm(const void* input, const void* weight, void* out, const void* group_list,
                                   int group_list_is_i64, int64_t m_total, int64_t k, int64_t n, int64_t num_groups,
                                   int trans_weight, int dtype, void* workspace, int64_t workspace_bytes,
                                   mojo_stream_t stream) {
  return mojo_hip_group_gemm_strided(input, weight, out, group_list, group_list_is_i64, m_total, k, n, num_groups, k, n,
                                     k * n, trans_weight ? 1 : n, trans_weight ? k : 1, nullptr, nullptr, dtype,
                                     workspace, workspace_bytes, stream);
}

extern "C" int mojo_hip_group_gemm_swiglu(const void* input, const void* weight, void* out, const void* group_list,
                                          int group_list_is_i64, int64_t m_total, int64_t k, int64_t inter,
                                          int64_t num_groups, int trans_weight, int dtype, void* workspace,
                                          int64_t workspace_bytes, mojo_stream_t stream) {
  MOJO_REQUIRE(num_groups > 0 && k > 0 && inter > 0 && m_total >= 0, MOJO_EINVAL, "group_gemm_swiglu: bad shape");
  if (m_total == 0) return MOJO_OK;
  MOJO_REQUIRE(input && weight && out && group_list, MOJO_EINVAL, "group_gemm_swiglu: null pointer");
  MOJO_REQUIRE(dtype == MOJO_F16 || dtype == MOJO_BF16, MOJO_EUNSUPPORTED, "group_gemm_swiglu: dtype %d (bf16 / fp16 only)", dtype);
  MOJO_REQUIRE(m_total < (1LL << 31) && k < (1LL << 31) && inter < (1LL << 30) && num_groups < (1 << 20), MOJO_EUNSUPPORTED,
               "group_gemm_swiglu: dimension too large");
  MOJO_REQUIRE(workspace && workspace_bytes >= mojo_hip_group_gemm_workspace_bytes(num_groups) && aligned_to(workspace, 4),
               MOJO_EWORKSPACE, "group_gemm_swiglu: workspace too small");
  const int64_t n = 2 * inter;
  GemmArgs a;
  a.A = input; a.W = weight; a.C = out; a.bias = nullptr;
  a.lda = k; a.ldc = inter; a.w_group = k * n; a.w_k = trans_weight ? 1 : n; a.w_n = trans_weight ? k : 1;
  a.K = static_cast<int>(k); a.N = static_cast<int>(n); a.G = static_cast<int>(num_groups);
  a.glu = 1;
  MOJO_REQUIRE(gemm_mfma256_glu_ok(a, dtype), MOJO_EUNSUPPORTED,
               "group_gemm_swiglu: needs the 256x256 MFMA kernel's layout and an intermediate size that is a multiple of 128");
  {  // few small experts / groups of about a hundred rows: the product on the 128-row tiles + mojo_hip_swiglu_rows beats the fused
     // 256 x 256 tiles (8 experts of 2048 x 2816: 38 us fused against ~20); the documented fallback of this entry point
    GemmArgs p = a;
    p.glu = 0;
    const int f = gemm_tile128_forced();
    const int64_t rows = std::max<int64_t>(1, m_total / num_groups);
    const bool one_round = num_groups * ceil_div(rows, 128) * ceil_div(n, 128) <= 256;   // (beyond one round of 128 x 128 tiles the fused
    if (gemm_tile128_group_ok(p, dtype) && !gemm_skinny_ragged_ok(p, dtype, m_total) &&   //  tiles win back what the product alone loses:
        (f >= 0 ? f == 1 : (one_round && group_prefers_tile128(m_total, k, n, num_groups))))   //  8 x 300 rows of 2048 x 2816: 81 us fused, 101 not)
      MOJO_REQUIRE(false, MOJO_EUNSUPPORTED, "group_gemm_swiglu: the unfused route (group_gemm on 128-row tiles + swiglu_rows) is faster for this shape");
  }
  int32_t* ws = static_cast<int32_t*>(workspace);
  a.row_start = ws; a.tile_start = ws + (num_groups + 1);
  hipStream_t s = static_cast<hipStream_t>(stream);
  GemmTail tail;
  tail.C = out; tail.ld_bytes = inter * 2; tail.row_bytes = inter * 2;
  int rc = launch_group_prefix(group_list, group_list_is_i64, a.G, 256, m_total, ws, ws + (num_groups + 1), s, tail);
  if (rc) return rc;
  return launch_gemm_mfma256(a, dtype, m_total, s);
}

// Dense 16-bit products of more than 128 rows with few 256x256 output tiles (a prefill chunk of a few hundred tokens
// against a 4096..8192-wide projection: 16-64 tiles on 256 CUs): every tile walked the whole K and the launch took the
// same 68 us (K = 4096) from 129 to 2048 rows, 2-4 x the vendor library's time.  K is cut so that ~256 workgroups run; the
// slices go to fp32 slabs and launch_gemm_splitk_finalize sums them in slice order (deterministic).
// MOJO_HIP_GEMM_SPLITK=<n> forces the split (1 = off).
// `modelled_us`: the chosen launch's time with COLD weights — every call another weight, as a model's layers are; the one-weight
// graphs of the first A/Bs re-read up to 256 MB from the last-level cache, which flatters the kernels with little in flight per CU.
// Fitted on profiles/r5_gemm_cold_weights_ab.txt (94 launches of this kernel, 6 % rms): 12.3 us + K-tiles per slice x (0.60 per
// round + 0.77 x share of busy CUs) + 0.14 us per MB of slab + 7.1 us for the second launch of a split.
static int gemm_dense_splitk256(int64_t m, int64_t k, int64_t n, double* modelled_us = nullptr) {
  auto cold = [&](int64_t sk) {
    const double tiles_sk = static_cast<double>(ceil_div(m, 256) * ceil_div(n, 256) * sk), kts = static_cast<double>(k / 64) / sk;
    return 12.3 + std::ceil(tiles_sk / 256.0) * kts * 0.60 + tiles_sk / 256.0 * kts * 0.77 + (sk > 1 ? 0.14 * sk * m * n * 4.0 / 1e6 + 7.1 : 0.0);
  };
  if (modelled_us) *modelled_us = cold(1);
  // (m <= 128 too: decode-sized rows with [K,N] weights — `x @ w`, the GEMM + collective operators' trans_weight — have no
  // weight-streaming kernel and ran ONE round of 16-32 workgroups over the whole K: 72 us at 32 x 4096 x 4096 against 14 in the
  // vendor library, 137 us at K 8192 (profiles/r5_gemm_sweep_vs_lib.txt).  Slices of at least 4 K-tiles there.)
  if (m < 1 || n % 4 != 0 || k % 64 != 0) return 1;
  const int64_t tiles = ceil_div(m, 256) * ceil_div(n, 256), nkt = k / 64, min_slice = m <= 128 ? 4 : 8;
  if (int64_t sk = MOJO_SWITCH("MOJO_HIP_GEMM_SPLITK", 0); sk > 0) {
    if (sk > nkt / min_slice) sk = nkt / min_slice;
    if (sk < 1) sk = 1;
    if (modelled_us) *modelled_us = cold(sk);
    return static_cast<int>(sk);
  }
  // time of a split in us, from measurements on this chip: a workgroup's K-tile takes ~1.06 us (68 us for the 64 K-tiles of
  // K = 4096), one workgroup per CU, so rounds(tiles * sk) x K-tiles per slice; the slabs are written and read once each at
  // ~5 TB/s, and the second launch costs ~4 us.  The smallest split within 10 % of the best wins; at least 8 K-tiles per slice.
  auto cost = [&](int64_t sk) {
    const double gemm = static_cast<double>(ceil_div(tiles * sk, 256)) * (static_cast<double>(nkt) / sk) * 1.06;
    return sk == 1 ? gemm : gemm + 2.0 * sk * m * n * 4.0 / 5e6 + 4.0;
  };
  int64_t best = 1;
  double best_cost = cost(1) * 0.9;                     // a split has to gain 10 % before it is worth a second launch
  for (int64_t sk = 2; sk <= (m <= 128 ? 32 : 16) && sk <= nkt / min_slice; ++sk) {
    const double c = cost(sk);
    if (c < best_cost * 0.999) { best = sk; best_cost = c; }
  }
  if (modelled_us) *modelled_us = cold(best);
  return static_cast<int>(best);
}

// The 128 x 128 tiles' own K split (few tiles over a long K: a chunk of 65-1024 rows against a 1024-8192-wide projection, where
// even 128 x 128 tiles leave most CUs idle and one tile walks 64-224 K-tiles: M 256 x 8192 x 1024 took 32 us on the 256 kernel's
// split and 45 unsplit here, 17.5 with 6 slices; hipBLASLt 16.9).  Slices of at least 4 K-tiles, tiles x slices within one per CU.
// Time model for COLD weights (profiles/r5_gemm_cold_weights_ab.txt: every split of 168 shapes, each call of the graph on another
// copy of the weight; the first, one-weight A/B — r5_gemm_tile128_splitk_ab.txt — had the weights in the last-level cache): with
// cold operands a workgroup fills at ~58 GB/s instead of ~90: 0.50-0.58 us per K-tile whatever the number of busy CUs (0.30-0.43
// from the cache; a per-CU ceiling of the miss path — a fifth ring stage and a prefetch wave were measured and change nothing,
// DESIGN Appendix A 30-31), so spreading the K-tiles over more CUs pays more than it did warm: unsplit 6.75 us + K-tiles x
// (0.50 + 0.075 x share of busy CUs); split 5.6 us + K-tiles per slice x 0.55 + 0.55 us per MB of slab + 0.32 us per slice; and no
// form runs under the operands' one pass over HBM at the ~4.8 TB/s these 128-byte row pieces reach.  Returns the split
// (1 = none) and its modelled time.
// `one_shot`: the constants of the first, one-weight A/B (profiles/r5_gemm_tile128_splitk_ab.txt) instead — used for 16..64 rows,
// see gemm_rows128_prefers_tile128.
static int gemm_tile128_splitk(int64_t m, int64_t k, int64_t n, bool w_nmajor, double* modelled_us, bool one_shot = false) {
  const int64_t tiles = ceil_div(m, 128) * ceil_div(n, 128), nkt = k / 64;
  auto cost = [&](int64_t sk) {
    const double busy = static_cast<double>(tiles * sk) / 256.0, kn = w_nmajor ? 0.05 : 0.0;
    const double kts = static_cast<double>(ceil_div(nkt, sk));
    if (one_shot) {
      double loop = kts * (0.30 + kn + 0.125 * busy);
      const double hbm = (static_cast<double>(k) * n + static_cast<double>(m) * k) * 2.0 / 6.5e6;
      if (loop < hbm) loop = hbm;
      return sk == 1 ? (5.0 + loop) / 0.95 : 6.0 + loop + 0.26 * static_cast<double>(sk * m * n) * 4.0 / 1e6 + 0.44 * sk;   // (a split has to model 5 % under)
    }
    double loop = kts * (sk == 1 ? 0.50 + kn + 0.075 * busy : 0.55 + kn);
    const double hbm = (static_cast<double>(k) * n + static_cast<double>(m) * k) * 2.0 / 4.8e6;
    if (loop < hbm) loop = hbm;
    return sk == 1 ? 6.75 + loop : 5.6 + loop + 0.55 * static_cast<double>(sk * m * n) * 4.0 / 1e6 + 0.32 * sk;
  };
  *modelled_us = cost(1);
  if (tiles > 256 || n % 4 != 0) { *modelled_us = 1e30; return 1; }
  if (int64_t sk = MOJO_SWITCH("MOJO_HIP_GEMM_SPLITK", 0); sk > 0) {
    if (sk > nkt) sk = nkt;
    *modelled_us = cost(sk);
    return static_cast<int>(sk);
  }
  int64_t best = 1;
  double best_cost = cost(1);
  for (int64_t sk = 2; sk <= 16 && tiles * sk <= 256 && sk <= nkt / 4; ++sk) {
    const double c = cost(sk);
    if (c < best_cost * 0.999) { best = sk; best_cost = c; }
  }
  if (best > 1) *modelled_us = best_cost;
  return static_cast<int>(best);
}

// 128-row tiles (gemm_tile128.hip) or the 256 x 256 kernel with its best split?  The smaller of the two modelled times, both for
// COLD weights (profiles/r5_gemm_cold_weights_ab.txt, 168 shapes x every form; see gemm_tile128_splitk and gemm_dense_splitk256):
// up to 256 tiles of 128 x 128 with their own K split; beyond, up to 256 tiles of 128 x 256 (eight waves): 12 us + K-tiles x
// (0.72 + 0.12 x share of busy CUs); [K,N] weights (transposed fragment reads) 0.05 us more per K-tile.  On the cold grid the rule is
// within 1.6 % (geometric mean) of always picking the fastest measured form and level with hipBLASLt (1.00 x its time over the
// grid, rows 8 to 4096); the warm-fitted rule it replaces was 3.6 % off the best and picked 128-row tiles for long-K products the
// 256 kernel runs 20-25 % faster cold (M 1024 x 14336 x 4096: 136 against 108 us).  On the warm grid the cold rule gives up 0.9 %.
static bool gemm_dense_prefers_tile128(int64_t m, int64_t k, int64_t n, bool w_nmajor, int* splitk128) {
  const int64_t narrow = ceil_div(m, 128) * ceil_div(n, 128), wide = ceil_div(m, 128) * ceil_div(n, 256), nkt = k / 64;
  double t128;
  *splitk128 = 1;
  if (narrow <= 256) *splitk128 = gemm_tile128_splitk(m, k, n, w_nmajor, &t128);
  else if (wide <= 256) t128 = 12.0 + nkt * (0.72 + (w_nmajor ? 0.05 : 0.0) + 0.12 * wide / 256.0);
  else return false;
  double t256 = 0;
  (void)gemm_dense_splitk256(m, k, n, &t256);
  // (at most 128 rows — [K,N] weights, which have no weight-streaming kernel: the 256 kernel's split is slices of at least 4 K-tiles
  // on 16-32 workgroups, 27.6 us at 100 x 8192 x 1024 against 15.9 on 128-row tiles in 8 slices; profiles/r5_gemm_tile128_splitk_ab_kn.txt)
  return t128 < t256;
}

// At most 128 rows with [N,K] weights — the weight-streaming kernels' range — against one row of 128-row tiles (a lone m-tile:
// ceil(n / 128) tiles of 128 x 128 with their own K split, or rounds of 128 x 256 tiles beyond the chip).  65..128 rows are the
// stream's weak zone (its 128-row form reads the weights at ~3.5 TB/s: 88 us for 128 x 4096 x 28672 against 58 on the tiles, 440 us
// for a 128-row lm_head 4096 x 128256 against 231 in the vendor library); up to 64 rows the stream wins on cold weights almost
// everywhere.  Both sides modelled for COLD weights (profiles/r5_gemm_cold_weights_ab.txt; the stream, 65 launches, 9 % rms): up to 64
// rows (7.4 + 0.05 m) us + (0.144 + 0.00044 m) us per MB of weights; 65..128 rows (3.9 + 0.05 m) us + (0.264 + 0.00044 m) us per MB.
static bool gemm_rows128_prefers_tile128(int64_t m, int64_t k, int64_t n, int* splitk128) {
  *splitk128 = 1;
  if (m > 128) return false;
  const int64_t narrow = ceil_div(n, 128), wide = ceil_div(n, 256), nkt = k / 64;
  double t128;
  const double mb = static_cast<double>(k) * n * 2.0 / 1e6;
  // 16..64 rows: a 10-30 us launch between other kernels is neither of the two A/B regimes.  Measured where it runs — the Llama-3-8B
  // decode layer at B 64, six alternating repetitions (scripts/probes/decode_layer_tile128_ab.py, profiles/r5_decode_layer_tile128_ab.txt):
  // QKV / o / down projections on the tiles with their K split 290.7 -> 279.4 us for the fused layer (medians; 288-296 against
  // 279-285), 306.9 -> 299.3 us for the layer written against the reference's operators — where the back-to-back cold graphs put the
  // stream ahead (its launches overlap head and tail there) and the one-weight graphs had put the tiles ahead by 7 us.  So this range
  // takes the one-weight models of both sides (r5_gemm_tile128_splitk_ab_nk_small.txt); weights under 32 MB stay with the stream.
  const bool one_shot = m >= 16 && m <= 64;              // (B 32 / 16: 195.8 -> 190.6 / 154.2 -> 150.1 us fused with the tiles forced; B 8 and 1: level)
  if (narrow <= 256) *splitk128 = gemm_tile128_splitk(m, k, n, false, &t128, one_shot);
  else t128 = 6.0 + nkt * 0.81 * (wide <= 256 ? 1.0 : wide / 256.0);
  double t_stream;
  if (one_shot) t_stream = 6.5 + 0.07 * m + (0.16 + 0.0003 * m) * mb;
  else if (m <= 64) t_stream = 0.9 * (7.4 + 0.05 * m + (0.144 + 0.00044 * m) * mb);   // (decode rows: the tiles have to model 10 % under the stream,
  else t_stream = 3.9 + 0.05 * m + (0.264 + 0.00044 * m) * mb;                         //  whose fused decode forms — GLU, norm, RoPE + store — share its bits)
  if (m <= 64 && mb < 32.0) return false;               // (small weights at decode rows: outside the fitted range; the stream)
  if (one_shot && *splitk128 == 1) return false;        // (unsplit tiles over a wide N lost in the layer: 64 x 4096 x 28672, +15 us)
  return t128 < t_stream;
}

// (for the fused callers outside this file that must take mojo_hip_gemm's route: qkv_fused.hip)
bool mojo::gemm_rows128_takes_tile128(const GemmArgs& a, int dtype, int64_t m, int64_t k, int64_t n, int* splitk128) {
  *splitk128 = 1;
  if (!(a.w_k == 1 && m <= 128 && (dtype == MOJO_BF16 || dtype == MOJO_F16) && k % 64 == 0)) return false;
  const bool prefers = gemm_rows128_prefers_tile128(m, k, n, splitk128);
  return gemm_tile128_use(a, dtype, m, prefers);
}

extern "C" int64_t mojo_hip_gemm_workspace_bytes(int64_t m, int64_t k, int64_t n) {
  int sk = gemm_skinny_splitk(m, k, n, 1);
  const int sk256 = gemm_dense_splitk256(m, k, n);
  if (sk256 > sk) sk = sk256;
  double t;
  const int sk128 = k % 64 == 0 ? gemm_tile128_splitk(m, k, n, false, &t) : 1;   // (the same split for either weight layout)
  if (sk128 > sk) sk = sk128;
  return 64 + (sk > 1 ? static_cast<int64_t>(sk) * m * n * 4 : 0);
}

extern "C" int mojo_hip_gemm_rowmap(const void* input, const void* weight, const void* bias, void* out, int64_t m,
                                    int64_t k, int64_t n, int64_t lda, int64_t ldc, int64_t w_k_stride,
                                    int64_t w_n_stride, const int64_t a_map[3], const int64_t c_map[3], int dtype,
                                    void* workspace, int64_t workspace_bytes, mojo_stream_t stream) {
  MOJO_REQUIRE(k > 0 && n > 0 && m >= 0, MOJO_EINVAL, "gemm: bad shape");
  if (m == 0) return MOJO_OK;
  MOJO_REQUIRE(input && weight && out, MOJO_EINVAL, "gemm: null pointer");
  MOJO_REQUIRE(dtype == MOJO_F32 || dtype == MOJO_F16 || dtype == MOJO_BF16, MOJO_EUNSUPPORTED,
               "gemm: dtype %d not supported", dtype);
  MOJO_REQUIRE(m < (1LL << 31) && k < (1LL << 31) && n < (1LL << 31), MOJO_EUNSUPPORTED, "gemm: dimension too large");
  MOJO_REQUIRE(workspace && workspace_bytes >= 16 && aligned_to(workspace, 4), MOJO_EWORKSPACE, "gemm: workspace too small");
  GemmArgs a;
  a.A = input; a.W = weight; a.C = out; a.bias = bias;
  a.bias_fused = (bias && w_k_stride == 1) ? 1 : 0;                     // [N,K] weights = the golden's F.linear(input, weight, bias): one rounding
  a.lda = lda; a.ldc = ldc; a.w_group = 0; a.w_k = w_k_stride; a.w_n = w_n_stride;
  a.K = static_cast<int>(k); a.N = static_cast<int>(n); a.G = 1;
  if (a_map) {
    MOJO_REQUIRE(a_map[0] >= 0 && a_map[0] < (1LL << 31) && a_map[1] < (1LL << 31) && a_map[2] < (1LL << 31), MOJO_EINVAL, "gemm: bad A row map");
    a.a_rc = static_cast<int>(a_map[0]); a.a_ml = static_cast<int>(a_map[1]); a.a_off = static_cast<int>(a_map[2]);
  }
  if (c_map) {
    MOJO_REQUIRE(c_map[0] >= 0 && c_map[0] < (1LL << 31) && c_map[1] < (1LL << 31) && c_map[2] < (1LL << 31), MOJO_EINVAL, "gemm: bad C row map");
    a.c_rc = static_cast<int>(c_map[0]); a.c_ml = static_cast<int>(c_map[1]); a.c_off = static_cast<int>(c_map[2]);
  }
  int32_t* ws = static_cast<int32_t*>(workspace);
  a.row_start = ws; a.tile_start = ws + 2;
  hipStream_t s = static_cast<hipStream_t>(stream);
  a.uniform_rows = static_cast<int>(m);
  auto take_split128 = [&](int sk128) {                  // the 128-row tiles' own K split, if the workspace holds its slabs
    if (sk128 > 1 && workspace_bytes >= 64 + static_cast<int64_t>(sk128) * m * n * 4 && aligned_to(workspace, 16)) {
      a.splitk = sk128; a.slab = static_cast<char*>(workspace) + 64; a.slab_rows = static_cast<int>(m);
    }
  };
  if (int sk128 = 1; gemm_rows128_takes_tile128(a, dtype, m, k, n, &sk128)) {
    take_split128(sk128);
    return launch_gemm_tile128(a, dtype, m, s);
  }
  if (w_k_stride == 1 && (dtype == MOJO_BF16 || dtype == MOJO_F16)) {          // decode-sized, K-major weights: maybe split K
    const int sk = gemm_skinny_splitk(m, k, n, 1);
    if (sk > 1 && workspace_bytes >= 64 + static_cast<int64_t>(sk) * m * n * 4 && aligned_to(workspace, 16)) {
      a.splitk = sk; a.slab = static_cast<char*>(workspace) + 64; a.slab_rows = static_cast<int>(m);
      if (!gemm_skinny_ok(a, dtype)) { a.splitk = 1; a.slab = nullptr; }
    }
  }
  // (up to 64 rows of [N,K] weights the weight stream cannot take — N % 64: the 256 kernel, whose unsplit bits the fused decode forms share)
  if (a.splitk == 1 && !gemm_skinny_ok(a, dtype) && (dtype == MOJO_BF16 || dtype == MOJO_F16) && k % 64 == 0 && (m > 64 || w_k_stride != 1)) {
    int sk128 = 1;
    const bool prefers = gemm_dense_prefers_tile128(m, k, n, w_n_stride == 1, &sk128);
    if (gemm_tile128_use(a, dtype, m, prefers)) {
      take_split128(sk128);
      return launch_gemm_tile128(a, dtype, m, s);      // (split: + launch_gemm_splitk_finalize)
    }
  }
  if (a.splitk == 1 && (dtype == MOJO_BF16 || dtype == MOJO_F16) && !gemm_skinny_ok(a, dtype) && gemm_mfma256_ok(a, dtype)) {
    const int sk = gemm_dense_splitk256(m, k, n);       // few output tiles: cut K, sum the slices in a second launch
    if (sk > 1 && workspace_bytes >= 64 + static_cast<int64_t>(sk) * m * n * 4 && aligned_to(workspace, 16)) {
      a.splitk = sk; a.slab = static_cast<char*>(workspace) + 64; a.slab_rows = static_cast<int>(m);
      const int rc = launch_gemm_mfma256(a, dtype, m, s);
      if (rc) return rc;
      return launch_gemm_splitk_finalize(a, dtype, m, s);
    }
  }
  return run_gemm(a, dtype, m, s);
}

extern "C" int mojo_hip_gemm(const void* input, const void* weight, const void* bias, void* out, int64_t m, int64_t k,
                             int64_t n, int64_t lda, int64_t ldc, int64_t w_k_stride, int64_t w_n_stride, int dtype,
                             void* workspace, int64_t workspace_bytes, mojo_stream_t stream) {
  return mojo_hip_gemm_rowmap(input, weight, bias, out, m, k, n, lda, ldc, w_k_stride, w_n_stride, nullptr, nullptr,
                              dtype, workspace, workspace_bytes, stream);
}

// ---- decode-sized fusions (a decoder layer's MLP and projections at M <= 64; core/operators/moe.py:402-449 is the op chain) ----
extern "C" int mojo_hip_swiglu_rows(const void* gate, const void* up, void* out, int64_t rows, int64_t cols, int64_t ld_gate,
                                    int64_t ld_up, int64_t ld_out, int dtype, float swiglu_limit, mojo_stream_t stream);
extern "C" int mojo_hip_residual_add_rmsnorm(const void* hidden, const void* residual, const void* weight, void* normed_out,
                                             void* sum_out, int64_t rows, int64_t dim, int dtype, float eps, mojo_stream_t stream);

extern "C" int64_t mojo_hip_gemm_swiglu_workspace_bytes(int64_t m, int64_t k, int64_t inter) {
  return 64 + m * 2 * inter * 2 + mojo_hip_gemm_workspace_bytes(m, k, 2 * inter);
}

extern "C" int mojo_hip_gemm_swiglu(const void* input, const void* weight, void* out, int64_t m, int64_t k, int64_t inter,
                                    int64_t lda, int64_t ldc, int64_t w_n_stride, int dtype, void* workspace,
                                    int64_t workspace_bytes, mojo_stream_t stream) {
  MOJO_REQUIRE(k > 0 && inter > 0 && m >= 0, MOJO_EINVAL, "gemm_swiglu: bad shape");
  if (m == 0) return MOJO_OK;
  MOJO_REQUIRE(input && weight && out, MOJO_EINVAL, "gemm_swiglu: null pointer");
  MOJO_REQUIRE(dtype == MOJO_F16 || dtype == MOJO_BF16, MOJO_EUNSUPPORTED, "gemm_swiglu: dtype %d (bf16 / fp16 only)", dtype);
  MOJO_REQUIRE(m < (1LL << 31) && k < (1LL << 31) && inter < (1LL << 30), MOJO_EUNSUPPORTED, "gemm_swiglu: dimension too large");
  MOJO_REQUIRE(lda >= k && ldc >= inter && w_n_stride >= k, MOJO_EINVAL, "gemm_swiglu: bad strides");
  hipStream_t s = static_cast<hipStream_t>(stream);
  GemmArgs a;
  a.A = input; a.W = weight; a.C = out; a.bias = nullptr;
  a.lda = lda; a.ldc = ldc; a.w_group = 0; a.w_k = 1; a.w_n = w_n_stride;
  a.K = static_cast<int>(k); a.N = static_cast<int>(2 * inter); a.G = 1;
  a.row_start = nullptr; a.tile_start = nullptr;
  a.uniform_rows = static_cast<int>(m);
  a.glu = 1;
  if (gemm_skinny_glu_ok(a, dtype)) return launch_gemm_skinny_glu(a, dtype, s);
  // Prefill-sized rows: the 256 x 256 kernel's fused-SwiGLU tiles (an output tile pairs 128 gate columns with their 128 up
  // columns: MojoExperts' epilogue, here with one group) where they fill the chip — the [M, 2 I] product (117 MB written and
  // read back at M 2048, I 14336) never exists.  Below ~3/4 of a round of tiles the unfused route is faster: its product runs
  // on the 128-row tiles of gemm_tile128.hip.  MOJO_HIP_GEMM_SKINNY without bit 4 (SKINNY_GLU) turns every fused form off.
  if ((gemm_skinny_mask() & SKINNY_GLU) && m > 128 && gemm_mfma256_glu_ok(a, dtype) &&
      ceil_div(m, 256) * (inter / 128) >= 192 && lda % 8 == 0 && ldc % 8 == 0)
    return launch_gemm_mfma256(a, dtype, m, s);
  // any other shape: the product into the workspace, then the activation over its two halves (same bits: the fused
  // epilogue rounds where these two launches round)
  MOJO_REQUIRE(workspace && workspace_bytes >= mojo_hip_gemm_swiglu_workspace_bytes(m, k, inter) && aligned_to(workspace, 16),
               MOJO_EWORKSPACE, "gemm_swiglu: workspace too small");
  char* gu = static_cast<char*>(workspace) + 64;
  char* ws2 = gu + m * 2 * inter * 2;
  ws2 += (16 - (reinterpret_cast<uintptr_t>(ws2) & 15)) & 15;
  const int64_t ws2_bytes = workspace_bytes - (ws2 - static_cast<char*>(workspace));
  int rc = mojo_hip_gemm(input, weight, nullptr, gu, m, k, 2 * inter, lda, 2 * inter, 1, w_n_stride, dtype, ws2, ws2_bytes, stream);
  if (rc) return rc;
  return mojo_hip_swiglu_rows(gu, gu + inter * 2, out, m, inter, 2 * inter, 2 * inter, ldc, dtype, 0.f, stream);
}

extern "C" int64_t mojo_hip_gemm_residual_rmsnorm_workspace_bytes(int64_t m, int64_t k, int64_t n) {
  return 64 + m * n * 2 + 16 + mojo_hip_gemm_workspace_bytes(m, k, n);
}

extern "C" int mojo_hip_gemm_residual_rmsnorm(const void* input, const void* weight, const void* bias, const void* residual,
                                              const void* norm_weight, void* normed_out, void* sum_out, void* gemm_out,
                                              int64_t m, int64_t k, int64_t n, int64_t lda, int64_t w_k_stride,
                                              int64_t w_n_stride, int dtype, float eps, void* workspace,
                                              int64_t workspace_bytes, mojo_stream_t stream) {
  MOJO_REQUIRE(k > 0 && n > 0 && m >= 0, MOJO_EINVAL, "gemm_residual_rmsnorm: bad shape");
  if (m == 0) return MOJO_OK;
  MOJO_REQUIRE(input && weight && norm_weight && normed_out, MOJO_EINVAL, "gemm_residual_rmsnorm: null pointer");
  MOJO_REQUIRE(dtype == MOJO_F16 || dtype == MOJO_BF16, MOJO_EUNSUPPORTED, "gemm_residual_rmsnorm: dtype %d (bf16 / fp16 only)", dtype);
  MOJO_REQUIRE(m < (1LL << 31) && k < (1LL << 31) && n < (1LL << 30), MOJO_EUNSUPPORTED, "gemm_residual_rmsnorm: dimension too large");
  MOJO_REQUIRE(workspace && workspace_bytes >= mojo_hip_gemm_residual_rmsnorm_workspace_bytes(m, k, n) && aligned_to(workspace, 16),
               MOJO_EWORKSPACE, "gemm_residual_rmsnorm: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  char* prod = static_cast<char*>(workspace) + 64;                      // the product when the caller does not want it
  char* ws2 = prod + m * n * 2;
  ws2 += (16 - (reinterpret_cast<uintptr_t>(ws2) & 15)) & 15;
  const int64_t ws2_bytes = workspace_bytes - (ws2 - static_cast<char*>(workspace));
  // A split product's slabs go straight into the norm (the finalize launch IS the norm).  The plan below is mojo_hip_gemm's own,
  // step for step, so the fused form and the separate calls add the same slices in the same order (same bits).
  GemmArgs a;
  a.A = input; a.W = weight; a.C = gemm_out; a.bias = bias;
  a.bias_fused = (bias && w_k_stride == 1) ? 1 : 0;                      // [N,K]: F.linear semantics, as mojo_hip_gemm
  a.lda = lda; a.ldc = n; a.w_group = 0; a.w_k = w_k_stride; a.w_n = w_n_stride;
  a.K = static_cast<int>(k); a.N = static_cast<int>(n); a.G = 1;
  a.row_start = nullptr; a.tile_start = nullptr;
  a.uniform_rows = static_cast<int>(m);
  auto slabs = [&](int sk) {
    if (sk <= 1 || ws2_bytes < 64 + static_cast<int64_t>(sk) * m * n * 4) return false;
    a.splitk = sk; a.slab = ws2 + 64; a.slab_rows = static_cast<int>(m); a.defer_finalize = 1;
    return true;
  };
  auto unsplit = [&]() { a.splitk = 1; a.slab = nullptr; a.defer_finalize = 0; };
  auto tiles128_into_norm = [&](int sk128) -> int {     // 1 = launched, 0 = not this way, < 0 = error code
    if (!slabs(sk128)) return 0;
    if (!gemm_tile128_group_ok(a, dtype) || !gemm_splitk_resnorm_ok(a, dtype, residual, norm_weight, normed_out, sum_out)) { unsplit(); return 0; }
    int rc = launch_gemm_tile128(a, dtype, m, s);
    if (!rc) rc = launch_gemm_splitk_resnorm(a, dtype, m, residual, norm_weight, normed_out, sum_out, eps, s);
    return rc ? rc : 1;
  };
  bool planned = false;                                  // true: mojo_hip_gemm takes a route that is not fused here
  if (int sk128 = 1; gemm_rows128_takes_tile128(a, dtype, m, k, n, &sk128)) {   // at most 128 rows: 128-row tiles where the model says so
    const int r = tiles128_into_norm(sk128);
    if (r) return r < 0 ? r : MOJO_OK;
    planned = true;
  }
  if (!planned && w_k_stride == 1) {                                     // decode-sized split of the weight-streaming kernel
    const int sk = gemm_skinny_splitk(m, k, n, 1);
    if (slabs(sk)) {
      if (gemm_skinny_ok(a, dtype)) {
        if (gemm_splitk_resnorm_ok(a, dtype, residual, norm_weight, normed_out, sum_out)) {
          const int rc = launch_gemm_skinny(a, dtype, s);
          if (rc) return rc;
          return launch_gemm_splitk_resnorm(a, dtype, m, residual, norm_weight, normed_out, sum_out, eps, s);
        }
        planned = true;
      }
    }
    unsplit();
  }
  if (!planned && !gemm_skinny_ok(a, dtype) && k % 64 == 0 && (m > 64 || w_k_stride != 1)) {   // 128-row tiles with their own split (a prefill chunk's o_proj / down_proj)
    int sk128 = 1;
    if (gemm_tile128_use(a, dtype, m, gemm_dense_prefers_tile128(m, k, n, w_n_stride == 1, &sk128))) {
      const int r = tiles128_into_norm(sk128);
      if (r) return r < 0 ? r : MOJO_OK;
    }
  }
  void* p = gemm_out ? gemm_out : static_cast<void*>(prod);
  int rc = mojo_hip_gemm(input, weight, bias, p, m, k, n, lda, n, w_k_stride, w_n_stride, dtype, ws2, ws2_bytes, stream);
  if (rc) return rc;
  return mojo_hip_residual_add_rmsnorm(p, residual, norm_weight, normed_out, sum_out, m, n, dtype, eps, stream);
}
