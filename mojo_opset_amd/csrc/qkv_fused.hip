// Decode step of an attention block in two launches: the fused QKV projection (decode-sized GEMM, K cut into fp32 slabs) and
// ONE finalize that sums the K slices, rounds, adds the bias, rotates q and k (MojoApplyRoPE, rotate-half over the whole
// head), writes q for the attention kernel and stores k / v into the paged caches (MojoStorePagedKVCache, decode mode).
// It replaces: split-K finalize -> apply_rope -> (copy of the strided v) -> store_paged_kv, four launches and three round
// trips of a [B, (Hq + 2 Hkv) D] tensor that exists only to be re-read.
//
// Same bits as the separate calls: the product is rounded to the storage type (+ bias after the rounding) exactly as
// gemm_skinny_finalize_kernel does, the rotation is rope.hip's arithmetic (two separately rounded fp32 products, their sum,
// one rounding), the stores are copies; rows the store would refuse (negative context length, page id out of range, table
// too short) are skipped as store_layout_kernel skips them.
//
// Thread = (token, head, 8 features of the first half + the 8 features of the second half they pair with).
// Algorithmic bytes: slabs sk * B * N * 4 (L2) + cos / sin + q out + 2 cache rows per kv head.
#include "gemm.h"

namespace mojo {

struct QkvFusedArgs {
  const float* slab;         // [sk][rows][N] fp32, or nullptr: `prod` holds the finished product
  const void* prod;          // [rows][N] storage type (ld = N)
  const void* bias;          // [N] or nullptr (slab form only: a finished product already carries it)
  int sk, rows, N;
  int hq, hkv, dim;
  const float* cos;          // [rows][dim] (row stride cs_ld)
  const float* sin;
  int64_t cs_ld;
  void* q_out;               // [rows][hq][dim]
  char* kc; char* vc;        // paged caches [blocks][hkv][page][dim]
  int64_t c_blk, c_head, c_tok;   // bytes
  int64_t num_blocks, page, max_pages, table_stride;
  const int32_t* table;
  const int32_t* ctx_lens;   // [rows]: position the new token of sequence `row` lands at
};

template <typename T>
__global__ __launch_bounds__(256) void qkv_rope_store_kernel(QkvFusedArgs a) {
  typedef typename vec_of<T, 8>::type V8;
  const int half = a.dim / 2, per_head = half / 8;          // threads per head
  const int heads = a.hq + 2 * a.hkv;
  const int64_t total = static_cast<int64_t>(a.rows) * heads * per_head;
  for (int64_t w = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; w < total; w += static_cast<int64_t>(gridDim.x) * 256) {
    const int i0 = static_cast<int>(w % per_head) * 8;
    const int hh = static_cast<int>((w / per_head) % heads);
    const int row = static_cast<int>(w / (static_cast<int64_t>(per_head) * heads));
    const int col = hh * a.dim + i0;                         // first-half column in the product; the partner is col + half
    V8 x1, x2;
    if (a.slab) {
      f32x4 s1l = {0.f, 0.f, 0.f, 0.f}, s1h = s1l, s2l = s1l, s2h = s1l;
      for (int sx0 = 0; sx0 < a.sk; sx0 += 4) {              // slices in index order, four slices' loads in flight
        f32x4 p[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int sx = min(sx0 + i, a.sk - 1);
          const float* p0 = a.slab + (static_cast<int64_t>(sx) * a.rows + row) * a.N + col;
          p[i][0] = *reinterpret_cast<const f32x4*>(p0);
          p[i][1] = *reinterpret_cast<const f32x4*>(p0 + 4);
          p[i][2] = *reinterpret_cast<const f32x4*>(p0 + half);
          p[i][3] = *reinterpret_cast<const f32x4*>(p0 + half + 4);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (sx0 + i < a.sk) { s1l += p[i][0]; s1h += p[i][1]; s2l += p[i][2]; s2h += p[i][3]; }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        x1[j] = static_cast<T>(j < 4 ? s1l[j & 3] : s1h[j & 3]);
        x2[j] = static_cast<T>(j < 4 ? s2l[j & 3] : s2h[j & 3]);
      }
      if (a.bias) {
        const V8 b1 = load_vec<T, 8>(static_cast<const T*>(a.bias) + col), b2 = load_vec<T, 8>(static_cast<const T*>(a.bias) + col + half);
#pragma unroll
        for (int j = 0; j < 8; ++j) {                      // F.linear semantics ([N,K] weights): bias joins the accumulator, one rounding
          x1[j] = round_with_bias<T>(j < 4 ? s1l[j & 3] : s1h[j & 3], b1[j], true);
          x2[j] = round_with_bias<T>(j < 4 ? s2l[j & 3] : s2h[j & 3], b2[j], true);
        }
      }
    } else {
      const T* p0 = static_cast<const T*>(a.prod) + static_cast<int64_t>(row) * a.N + col;
      x1 = load_vec<T, 8>(p0);
      x2 = load_vec<T, 8>(p0 + half);
    }
    const bool is_v = hh >= a.hq + a.hkv;
    if (!is_v) {                                             // rope.hip's arithmetic
      const float* c = a.cos + static_cast<int64_t>(row) * a.cs_ld;
      const float* s = a.sin + static_cast<int64_t>(row) * a.cs_ld;
      float c1[8], c2[8], s1[8], s2[8];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(c + i0 + 4 * q), a2 = *reinterpret_cast<const f32x4*>(c + half + i0 + 4 * q);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(s + i0 + 4 * q), b2 = *reinterpret_cast<const f32x4*>(s + half + i0 + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) { c1[4 * q + e] = a1[e]; c2[4 * q + e] = a2[e]; s1[4 * q + e] = b1[e]; s2[4 * q + e] = b2[e]; }
      }
      V8 o1, o2;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f1 = static_cast<float>(x1[j]), f2 = static_cast<float>(x2[j]);
        o1[j] = static_cast<T>(__fadd_rn(__fmul_rn(f1, c1[j]), __fmul_rn(-f2, s1[j])));
        o2[j] = static_cast<T>(__fadd_rn(__fmul_rn(f2, c2[j]), __fmul_rn(f1, s2[j])));
      }
      x1 = o1; x2 = o2;
    }
    if (hh < a.hq) {
      T* dst = static_cast<T*>(a.q_out) + (static_cast<int64_t>(row) * a.hq + hh) * a.dim + i0;
      store_vec<T, 8>(dst, x1);
      store_vec<T, 8>(dst + half, x2);
      continue;
    }
    // store_layout_kernel, decode mode: sequence = row, one token at position ctx_lens[row]
    const int32_t pos = a.ctx_lens[row];
    if (pos < 0) continue;
    const int64_t lp = pos / a.page;
    if (lp >= a.max_pages) continue;
    const int32_t blk = a.table[static_cast<int64_t>(row) * a.table_stride + lp];
    if (blk < 0 || blk >= a.num_blocks) continue;
    const int kvh = is_v ? hh - a.hq - a.hkv : hh - a.hq;
    char* dst = (is_v ? a.vc : a.kc) + blk * a.c_blk + kvh * a.c_head + (pos - lp * a.page) * a.c_tok + i0 * sizeof(T);
    *reinterpret_cast<V8*>(dst) = x1;
    *reinterpret_cast<V8*>(dst + half * sizeof(T)) = x2;
  }
}

}  // namespace mojo

using namespace mojo;

extern "C" int64_t mojo_hip_gemm_workspace_bytes(int64_t m, int64_t k, int64_t n);
extern "C" int mojo_hip_gemm(const void* input, const void* weight, const void* bias, void* out, int64_t m, int64_t k, int64_t n,
                             int64_t lda, int64_t ldc, int64_t w_k_stride, int64_t w_n_stride, int dtype, void* workspace,
                             int64_t workspace_bytes, mojo_stream_t stream);

extern "C" int64_t mojo_hip_qkv_rope_store_workspace_bytes(int64_t m, int64_t k, int64_t n) {
  return 64 + m * n * 2 + 16 + mojo_hip_gemm_workspace_bytes(m, k, n);
}

extern "C" int mojo_hip_qkv_rope_store(const void* input, const void* weight, const void* bias, const float* cos,
                                       const float* sin, int64_t cos_sin_row_stride, void* q_out, void* key_cache,
                                       void* value_cache, const int32_t* block_table, int64_t block_table_stride,
                                       int64_t max_blocks_per_seq, const int32_t* context_kv_lens, int64_t batch, int64_t k,
                                       int64_t q_heads, int64_t kv_heads, int64_t head_dim, int64_t lda, int64_t w_n_stride,
                                       int64_t num_blocks, int64_t block_size, int64_t cache_block_stride,
                                       int64_t cache_head_stride, int64_t cache_token_stride, int dtype, void* workspace,
                                       int64_t workspace_bytes, mojo_stream_t stream) {
  if (batch == 0) return MOJO_OK;
  MOJO_REQUIRE(batch > 0 && k > 0 && q_heads > 0 && kv_heads > 0 && head_dim > 0, MOJO_EINVAL, "qkv_rope_store: bad shape");
  MOJO_REQUIRE(input && weight && cos && sin && q_out && key_cache && value_cache && block_table && context_kv_lens, MOJO_EINVAL,
               "qkv_rope_store: null pointer");
  MOJO_REQUIRE(dtype == MOJO_F16 || dtype == MOJO_BF16, MOJO_EUNSUPPORTED, "qkv_rope_store: dtype %d (bf16 / fp16 only)", dtype);
  MOJO_REQUIRE(head_dim % 16 == 0 && cos_sin_row_stride % 4 == 0 && aligned_to(cos, 16) && aligned_to(sin, 16) && aligned_to(q_out, 16) &&
                   aligned_to(key_cache, 16) && aligned_to(value_cache, 16) && (cache_block_stride * 2) % 16 == 0 &&
                   (cache_head_stride * 2) % 16 == 0 && (cache_token_stride * 2) % 16 == 0 && (!bias || aligned_to(bias, 16)),
               MOJO_EUNSUPPORTED, "qkv_rope_store: head_dim must be a multiple of 16 and every row 16-byte aligned");
  const int64_t n = (q_heads + 2 * kv_heads) * head_dim;
  MOJO_REQUIRE(batch < (1LL << 31) && k < (1LL << 31) && n < (1LL << 31), MOJO_EUNSUPPORTED, "qkv_rope_store: dimension too large");
  MOJO_REQUIRE(workspace && workspace_bytes >= mojo_hip_qkv_rope_store_workspace_bytes(batch, k, n) && aligned_to(workspace, 16),
               MOJO_EWORKSPACE, "qkv_rope_store: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  char* prod = static_cast<char*>(workspace) + 64;
  char* ws2 = prod + batch * n * 2;
  ws2 += (16 - (reinterpret_cast<uintptr_t>(ws2) & 15)) & 15;
  const int64_t ws2_bytes = workspace_bytes - (ws2 - static_cast<char*>(workspace));
  QkvFusedArgs f;
  f.slab = nullptr; f.prod = prod; f.bias = nullptr; f.sk = 1;
  f.rows = static_cast<int>(batch); f.N = static_cast<int>(n);
  f.hq = static_cast<int>(q_heads); f.hkv = static_cast<int>(kv_heads); f.dim = static_cast<int>(head_dim);
  f.cos = cos; f.sin = sin; f.cs_ld = cos_sin_row_stride; f.q_out = q_out;
  f.kc = static_cast<char*>(key_cache); f.vc = static_cast<char*>(value_cache);
  f.c_blk = cache_block_stride * 2; f.c_head = cache_head_stride * 2; f.c_tok = cache_token_stride * 2;
  f.num_blocks = num_blocks; f.page = block_size; f.max_pages = max_blocks_per_seq; f.table_stride = block_table_stride;
  f.table = block_table; f.ctx_lens = context_kv_lens;
  bool slabs = false, planned = false;
  GemmArgs a;
  a.A = input; a.W = weight; a.bias = nullptr;
  a.lda = lda; a.ldc = n; a.w_group = 0; a.w_k = 1; a.w_n = w_n_stride;
  a.K = static_cast<int>(k); a.N = static_cast<int>(n); a.G = 1;
  a.row_start = nullptr; a.tile_start = nullptr;
  a.uniform_rows = static_cast<int>(batch);
  a.C = prod;                                              // (never written when the finalize is deferred; the alignment checks want a pointer)
  // mojo_hip_gemm's own route, step for step (same slices, same bits as the separate calls): at most 128 rows may take 128-row
  // tiles with their own K split; their slabs feed this kernel like the weight stream's.
  if (int sk128 = 1; gemm_rows128_takes_tile128(a, dtype, batch, k, n, &sk128)) {
    planned = true;
    if (sk128 > 1 && ws2_bytes >= 64 + static_cast<int64_t>(sk128) * batch * n * 4) {
      a.splitk = sk128; a.slab = ws2 + 64; a.slab_rows = static_cast<int>(batch); a.defer_finalize = 1;
      if (gemm_tile128_group_ok(a, dtype)) {
        const int rc = launch_gemm_tile128(a, dtype, batch, s);
        if (rc) return rc;
        f.slab = static_cast<const float*>(a.slab); f.sk = sk128; f.bias = bias;
        slabs = true;
      }
    }
  }
  const int sk = planned ? 1 : gemm_skinny_splitk(batch, k, n, 1);
  if (sk > 1 && ws2_bytes >= 64 + static_cast<int64_t>(sk) * batch * n * 4) {
    a.splitk = sk; a.slab = ws2 + 64; a.slab_rows = static_cast<int>(batch);
    a.defer_finalize = 1;
    if (gemm_skinny_ok(a, dtype)) {
      const int rc = launch_gemm_skinny(a, dtype, s);
      if (rc) return rc;
      f.slab = static_cast<const float*>(a.slab); f.sk = sk; f.bias = bias;
      slabs = true;
    }
  }
  if (!slabs) {
    const int rc = mojo_hip_gemm(input, weight, bias, prod, batch, k, n, lda, n, 1, w_n_stride, dtype, ws2, ws2_bytes, stream);
    if (rc) return rc;
  }
  const int64_t threads = batch * (q_heads + 2 * kv_heads) * (head_dim / 16);
  int64_t blocks = ceil_div(threads, 256);
  if (blocks > 4096) blocks = 4096;
  if (dtype == MOJO_BF16) hipLaunchKernelGGL(qkv_rope_store_kernel<bf16_t>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, f);
  else hipLaunchKernelGGL(qkv_rope_store_kernel<f16_t>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, f);
  MOJO_CHECK_LAUNCH("qkv_rope_store");
  return MOJO_OK;
}
