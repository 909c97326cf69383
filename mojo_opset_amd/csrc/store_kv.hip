// MojoStorePagedKVCache: bit-exact copy of new K/V tokens [T,Hkv,D] (token-major) into the paged
// caches [N,Hkv,page,D] (head-major).  HBM-bound byte mover: 16-byte lanes, one (token, head) row
// of D elements is contiguous on both sides.
//
// Algorithmic bytes per stored token: 2 tensors x Hkv x D x elt (read) + the same (write).
#include "common.h"

namespace mojo {

struct StoreArgs {
  const char* ks;
  const char* vs;
  char* kc;
  char* vc;
  int64_t tokens, heads, row_bytes, num_blocks, page;
  int64_t src_tok, src_head;          // bytes
  int64_t c_blk, c_head, c_tok;       // bytes
  // second tensor's own geometry (the layout kernel only; equal to the first tensor's for K/V, different for the MLA
  // store, where the compressed latent and the positional key have different widths and both go out in ONE launch)
  int64_t row_bytes_v, src_tok_v, src_head_v, c_blk_v, c_head_v, c_tok_v;
  int tensors = 2;                    // 1: only ks -> kc
  int stop_at_hole = 0;               // MLA store: a negative page id ends the sequence's store (and page 0 < 0 skips it)
};

template <int VB /* bytes per lane access: 16, 8, 4, 2 */>
__device__ __forceinline__ void copy_piece(const char* src, char* dst) {
  if constexpr (VB == 16) *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(src);
  else if constexpr (VB == 8) *reinterpret_cast<u32x2*>(dst) = *reinterpret_cast<const u32x2*>(src);
  else if constexpr (VB == 4) *reinterpret_cast<uint32_t*>(dst) = *reinterpret_cast<const uint32_t*>(src);
  else *reinterpret_cast<uint16_t*>(dst) = *reinterpret_cast<const uint16_t*>(src);
}

// One plan row per blockIdx.x; blockIdx.y slices the row's work so a few large chunks still fill
// the chip.  Work item = (tensor in {K,V}, token in chunk, head, piece of the D-row).
template <int VB>
__global__ __launch_bounds__(256) void store_plan_kernel(StoreArgs a, const int32_t* __restrict__ plan,
                                                         int64_t num_chunks) {
  const int pieces = static_cast<int>(a.row_bytes / VB);
  for (int64_t c = blockIdx.x; c < num_chunks; c += gridDim.x) {
    const int32_t src0 = plan[4 * c + 0], blk = plan[4 * c + 1], off = plan[4 * c + 2], len = plan[4 * c + 3];
    // refuse rows that would write outside the pools (the torch golden would raise IndexError)
    if (len <= 0 || blk < 0 || blk >= a.num_blocks || off < 0 || off + len > a.page || src0 < 0 ||
        src0 + len > a.tokens)
      continue;
    const int64_t per_tensor = static_cast<int64_t>(len) * a.heads * pieces;
    const int64_t total = 2 * per_tensor;
    for (int64_t w = static_cast<int64_t>(blockIdx.y) * blockDim.x + threadIdx.x; w < total;
         w += static_cast<int64_t>(gridDim.y) * blockDim.x) {
      const int which = w >= per_tensor;
      int64_t r = which ? w - per_tensor : w;
      const int p = static_cast<int>(r % pieces);
      r /= pieces;
      const int h = static_cast<int>(r % a.heads);
      const int t = static_cast<int>(r / a.heads);
      const char* src = (which ? a.vs : a.ks) + (src0 + t) * a.src_tok + h * a.src_head + p * VB;
      char* dst = (which ? a.vc : a.kc) + blk * a.c_blk + h * a.c_head + (off + t) * a.c_tok + p * VB;
      copy_piece<VB>(src, dst);
    }
  }
}

// Legacy arguments evaluated per token on the device (no host-side plan, no sync).
template <int VB>
__global__ __launch_bounds__(256) void store_layout_kernel(StoreArgs a, const int32_t* __restrict__ table,
                                                           int64_t table_stride, int64_t max_pages,
                                                           const int32_t* __restrict__ cu_q,
                                                           const int32_t* __restrict__ ctx_lens, int64_t batch) {
  const int pieces = static_cast<int>(a.row_bytes / VB);
  const int pieces_v = static_cast<int>(a.row_bytes_v / VB);
  const int64_t per_tensor = a.heads * pieces;
  const int64_t per_token = per_tensor + (a.tensors == 2 ? a.heads * pieces_v : 0);
  const int tok_per_block = blockDim.x / 64;                 // one wave per token
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * tok_per_block + wave; t < a.tokens;
       t += static_cast<int64_t>(gridDim.x) * tok_per_block) {
    // which sequence owns token t, and at which absolute position does it land?
    int64_t seq, pos;
    int64_t first_pos;                                       // where this sequence's new tokens start
    if (cu_q == nullptr) {
      if (t >= batch) continue;
      seq = t;
      const int32_t c = ctx_lens[seq];
      if (c < 0) continue;
      pos = c;
      first_pos = c;
    } else {
      if (t >= cu_q[batch] || t < cu_q[0]) continue;
      int64_t lo = 0, hi = batch;                            // largest seq with cu_q[seq] <= t
      while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (cu_q[mid] <= t) lo = mid; else hi = mid;
      }
      // skip over empty sequences that share the same offset: lo is the LAST index with cu_q<=t,
      // which is the only one with cu_q[lo+1] > t.
      seq = lo;
      const int32_t c = ctx_lens[seq];
      if (c < 0) continue;
      pos = c + (t - cu_q[seq]);
      first_pos = c;
    }
    const int64_t lp = pos / a.page;
    if (lp >= max_pages) continue;
    if (a.stop_at_hole) {
      // experimental/operators/kv_cache.py:80-98: the per-sequence loop skips a sequence whose first table entry is
      // negative and stops at the first negative page it meets: every page from the first written one up to this
      // token's must be valid (64 entries per step, one ballot)
      if (table[seq * table_stride] < 0) continue;
      bool hole = false;
      for (int64_t p0 = first_pos / a.page; p0 <= lp && !hole; p0 += 64) {
        const int64_t idx = p0 + lane;
        const int32_t v = idx <= lp ? table[seq * table_stride + idx] : 0;
        hole = __ballot(v < 0) != 0;
      }
      if (hole) continue;
    }
    const int32_t blk = table[seq * table_stride + lp];
    if (blk < 0 || blk >= a.num_blocks) continue;
    const int64_t slot = pos - lp * a.page;
    for (int64_t w = lane; w < per_token; w += 64) {
      const int which = w >= per_tensor;
      const int64_t r = which ? w - per_tensor : w;
      const int pc = which ? pieces_v : pieces;
      const int p = static_cast<int>(r % pc);
      const int h = static_cast<int>(r / pc);
      const char* src = which ? a.vs + t * a.src_tok_v + h * a.src_head_v + p * VB : a.ks + t * a.src_tok + h * a.src_head + p * VB;
      char* dst = which ? a.vc + blk * a.c_blk_v + h * a.c_head_v + slot * a.c_tok_v + p * VB
                        : a.kc + blk * a.c_blk + h * a.c_head + slot * a.c_tok + p * VB;
      copy_piece<VB>(src, dst);
    }
  }
}

static int pick_vb(const StoreArgs& a) {
  auto ok = [&](int vb) {
    return a.row_bytes % vb == 0 && a.src_tok % vb == 0 && a.src_head % vb == 0 && a.c_blk % vb == 0 &&
           a.c_head % vb == 0 && a.c_tok % vb == 0 && aligned_to(a.ks, vb) && aligned_to(a.kc, vb) &&
           (a.tensors == 1 || (aligned_to(a.vs, vb) && aligned_to(a.vc, vb) && a.row_bytes_v % vb == 0 && a.src_tok_v % vb == 0 &&
                               a.src_head_v % vb == 0 && a.c_blk_v % vb == 0 && a.c_head_v % vb == 0 && a.c_tok_v % vb == 0));
  };
  for (int vb : {16, 8, 4, 2})
    if (ok(vb)) return vb;
  return 0;
}

static int fill_args(StoreArgs& a, const void* ks, const void* vs, void* kc, void* vc, int64_t tokens,
                     int64_t heads, int64_t dim, int64_t num_blocks, int64_t page, int64_t eb, int64_t s_tok,
                     int64_t s_head, int64_t c_blk, int64_t c_head, int64_t c_tok) {
  MOJO_REQUIRE(eb == 1 || eb == 2 || eb == 4, MOJO_EINVAL, "store_paged_kv: elt_bytes must be 1, 2 or 4");
  MOJO_REQUIRE(heads > 0 && dim > 0 && page > 0 && num_blocks >= 0 && tokens >= 0, MOJO_EINVAL,
               "store_paged_kv: bad shape");
  a.ks = static_cast<const char*>(ks);
  a.vs = static_cast<const char*>(vs);
  a.kc = static_cast<char*>(kc);
  a.vc = static_cast<char*>(vc);
  a.tokens = tokens; a.heads = heads; a.row_bytes = dim * eb; a.num_blocks = num_blocks; a.page = page;
  a.src_tok = s_tok * eb; a.src_head = s_head * eb;
  a.c_blk = c_blk * eb; a.c_head = c_head * eb; a.c_tok = c_tok * eb;
  a.row_bytes_v = a.row_bytes; a.src_tok_v = a.src_tok; a.src_head_v = a.src_head;
  a.c_blk_v = a.c_blk; a.c_head_v = a.c_head; a.c_tok_v = a.c_tok;
  return MOJO_OK;
}

}  // namespace mojo

using namespace mojo;

extern "C" int mojo_hip_store_paged_kv_plan(const void* key_states, const void* value_states, void* key_cache,
                                            void* value_cache, const int32_t* plan, int64_t num_chunks,
                                            int64_t num_tokens, int64_t num_kv_heads, int64_t head_dim,
                                            int64_t num_blocks, int64_t block_size, int64_t elt_bytes,
                                            int64_t src_token_stride, int64_t src_head_stride,
                                            int64_t cache_block_stride, int64_t cache_head_stride,
                                            int64_t cache_token_stride, mojo_stream_t stream) {
  if (num_chunks == 0) return MOJO_OK;
  StoreArgs a;
  int rc = fill_args(a, key_states, value_states, key_cache, value_cache, num_tokens, num_kv_heads, head_dim,
                     num_blocks, block_size, elt_bytes, src_token_stride, src_head_stride, cache_block_stride,
                     cache_head_stride, cache_token_stride);
  if (rc) return rc;
  MOJO_REQUIRE(plan != nullptr && num_chunks > 0, MOJO_EINVAL, "store_paged_kv_plan: null plan");
  const int vb = pick_vb(a);
  MOJO_REQUIRE(vb != 0, MOJO_EUNSUPPORTED, "store_paged_kv: rows are not even 2-byte aligned");
  // a chunk holds at most `block_size` tokens: slice it so that each block moves ~16 KiB
  const int64_t max_items = 2 * block_size * num_kv_heads * (a.row_bytes / vb);
  int64_t gy = ceil_div(max_items, 256 * 4);
  if (gy > 64) gy = 64;
  int64_t gx = num_chunks > 65535 ? 65535 : num_chunks;
  dim3 grid(static_cast<unsigned>(gx), static_cast<unsigned>(gy));
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (vb) {
    case 16: hipLaunchKernelGGL(store_plan_kernel<16>, grid, dim3(256), 0, s, a, plan, num_chunks); break;
    case 8: hipLaunchKernelGGL(store_plan_kernel<8>, grid, dim3(256), 0, s, a, plan, num_chunks); break;
    case 4: hipLaunchKernelGGL(store_plan_kernel<4>, grid, dim3(256), 0, s, a, plan, num_chunks); break;
    default: hipLaunchKernelGGL(store_plan_kernel<2>, grid, dim3(256), 0, s, a, plan, num_chunks); break;
  }
  MOJO_CHECK_LAUNCH("store_paged_kv_plan");
  return MOJO_OK;
}

extern "C" int mojo_hip_store_paged_kv_layout(const void* key_states, const void* value_states, void* key_cache,
                                              void* value_cache, const int32_t* block_table,
                                              int64_t block_table_stride, int64_t max_blocks_per_seq,
                                              const int32_t* cu_q_lens, const int32_t* context_kv_lens,
                                              int64_t batch, int64_t num_tokens, int64_t num_kv_heads,
                                              int64_t head_dim, int64_t num_blocks, int64_t block_size,
                                              int64_t elt_bytes, int64_t src_token_stride, int64_t src_head_stride,
                                              int64_t cache_block_stride, int64_t cache_head_stride,
                                              int64_t cache_token_stride, mojo_stream_t stream) {
  if (num_tokens == 0 || batch == 0 || max_blocks_per_seq == 0) return MOJO_OK;
  StoreArgs a;
  int rc = fill_args(a, key_states, value_states, key_cache, value_cache, num_tokens, num_kv_heads, head_dim,
                     num_blocks, block_size, elt_bytes, src_token_stride, src_head_stride, cache_block_stride,
                     cache_head_stride, cache_token_stride);
  if (rc) return rc;
  MOJO_REQUIRE(block_table != nullptr && context_kv_lens != nullptr, MOJO_EINVAL,
               "store_paged_kv_layout: block_table and context_kv_lens are required");
  const int vb = pick_vb(a);
  MOJO_REQUIRE(vb != 0, MOJO_EUNSUPPORTED, "store_paged_kv: rows are not even 2-byte aligned");
  int64_t gx = ceil_div(num_tokens, 4);
  if (gx > 8192) gx = 8192;
  dim3 grid(static_cast<unsigned>(gx));
  hipStream_t s = static_cast<hipStream_t>(stream);
#define LAUNCH(VBV)                                                                                        \
  hipLaunchKernelGGL(store_layout_kernel<VBV>, grid, dim3(256), 0, s, a, block_table, block_table_stride, \
                     max_blocks_per_seq, cu_q_lens, context_kv_lens, batch)
  switch (vb) {
    case 16: LAUNCH(16); break;
    case 8: LAUNCH(8); break;
    case 4: LAUNCH(4); break;
    default: LAUNCH(2); break;
  }
#undef LAUNCH
  MOJO_CHECK_LAUNCH("store_paged_kv_layout");
  return MOJO_OK;
}

// ---- MojoStorePagedMLAKVCache (experimental/operators/kv_cache.py:13-106): the compressed-KV latent and the positional
//      key have different widths; they are the "first" and "second" tensor of ONE launch of the layout kernel (each with its
//      own row width and strides), with the MLA store's own hole semantics (a negative page id ends that sequence's store).
extern "C" int mojo_hip_store_paged_mla_kv(const void* compressed_kv_states, const void* k_pe_states,
                                           void* compressed_kv_cache, void* k_pe_cache, const int32_t* block_table,
                                           int64_t block_table_stride, int64_t max_blocks_per_seq,
                                           const int32_t* cu_q_lens, const int32_t* context_kv_lens, int64_t batch,
                                           int64_t num_tokens, int64_t kv_lora_rank, int64_t rope_dim,
                                           int64_t num_blocks, int64_t block_size, int64_t elt_bytes,
                                           int64_t ckv_src_token_stride, int64_t kpe_src_token_stride,
                                           int64_t ckv_block_stride, int64_t ckv_token_stride,
                                           int64_t kpe_block_stride, int64_t kpe_token_stride, mojo_stream_t stream) {
  if (num_tokens == 0 || batch == 0 || max_blocks_per_seq == 0) return MOJO_OK;
  MOJO_REQUIRE(compressed_kv_states && k_pe_states && compressed_kv_cache && k_pe_cache && block_table && context_kv_lens,
               MOJO_EINVAL, "store_paged_mla_kv: null pointer");
  MOJO_REQUIRE(rope_dim > 0, MOJO_EINVAL, "store_paged_mla_kv: bad shape");
  hipStream_t s = static_cast<hipStream_t>(stream);
  StoreArgs a;
  int rc = fill_args(a, compressed_kv_states, k_pe_states, compressed_kv_cache, k_pe_cache, num_tokens, 1, kv_lora_rank,
                     num_blocks, block_size, elt_bytes, ckv_src_token_stride, 0, ckv_block_stride, 0, ckv_token_stride);
  if (rc) return rc;
  a.row_bytes_v = rope_dim * elt_bytes; a.src_tok_v = kpe_src_token_stride * elt_bytes; a.src_head_v = 0;
  a.c_blk_v = kpe_block_stride * elt_bytes; a.c_head_v = 0; a.c_tok_v = kpe_token_stride * elt_bytes;
  a.tensors = 2;
  a.stop_at_hole = 1;
  const int vb = pick_vb(a);
  MOJO_REQUIRE(vb != 0, MOJO_EUNSUPPORTED, "store_paged_mla_kv: rows are not even 2-byte aligned");
  int64_t gx = ceil_div(num_tokens, 4);
  if (gx > 8192) gx = 8192;
  dim3 grid(static_cast<unsigned>(gx));
#define LAUNCH(VBV)                                                                                        \
  hipLaunchKernelGGL(store_layout_kernel<VBV>, grid, dim3(256), 0, s, a, block_table, block_table_stride, \
                     max_blocks_per_seq, cu_q_lens, context_kv_lens, batch)
  switch (vb) {
    case 16: LAUNCH(16); break;
    case 8: LAUNCH(8); break;
    case 4: LAUNCH(4); break;
    default: LAUNCH(2); break;
  }
#undef LAUNCH
  MOJO_CHECK_LAUNCH("store_paged_mla_kv");
  return MOJO_OK;
}
