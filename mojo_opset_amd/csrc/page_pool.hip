// Device-side block allocator of the paged KV cache.
//
// The reference's `PagedDummyCache.update` (mojo_opset/modeling/qwen3/mojo_qwen3_dense.py:84-123) walks the batch on the
// host: one `.item()` per sequence to read its context length, a Python slice of the free list, a table write.  Here
// the same bookkeeping is one launch with no host round trip, so a decode step that appends to the cache stays
// capturable in a HIP graph:
//
//   need_i  = ceil((ctx_i + new_i) / page) - ceil(ctx_i / page)            blocks sequence i must gain
//   sequence i takes   free[nf - (need_0 + .. + need_i) : nf - (need_0 + .. + need_{i-1})]
//   table[i, ceil(ctx_i / page) + j] = that slice[j]                        -- exactly the reference's pops from the END
//                                                                              of the free list, sequence by sequence
//   nf -= sum(need)
//
// so the resulting block tables are identical, entry for entry, to the reference's for the same free list.  Running out
// of blocks (or of table columns) cannot raise from a kernel: the launch then changes NOTHING and sets a sticky error
// word in the pool state, which the host reads when it next looks (PagedDummyCache.check()).
//
// pool_state: int32[4] = { num_free, error (0 ok, 1 out of blocks, 2 table too narrow), high-water of used blocks, - }.
#include "common.h"

namespace mojo {

__global__ __launch_bounds__(256) void page_pool_extend_kernel(int32_t* table, long long table_stride, int max_blocks,
                                                               const int32_t* seq_lens, const int32_t* new_lens,
                                                               int new_uniform, const int32_t* free_blocks,
                                                               int32_t* state, int32_t* store_ctx, int batch, int page,
                                                               int total_blocks) {
  __shared__ int s_scan[256];
  __shared__ int s_carry;
  __shared__ int s_bad;
  const int tid = threadIdx.x;
  if (tid == 0) { s_carry = 0; s_bad = 0; }
  __syncthreads();
  const int nf = state[0];
  const bool refused = state[1] != 0;                 // sticky: a pool in error stays untouched until the host resets it
  // store_ctx[i] = the context length the KV store must use for row i: its current length, or -1 ("skip this row",
  // core/operators/kv_cache.py:56-74) for rows that append nothing or when the pool is in error
  if (store_ctx && refused)
    for (int i = tid; i < batch; i += 256) store_ctx[i] = -1;
  if (refused) return;
  // pass 1: total demand and the table-width check (nothing is written before both are known to fit)
  for (int base = 0; base < batch; base += 256) {
    const int i = base + tid;
    int need = 0;
    if (i < batch) {
      const int ctx = seq_lens[i];
      const int nw = new_lens ? new_lens[i] : new_uniform;
      if (ctx >= 0 && nw > 0) {
        const int old_nb = (ctx + page - 1) / page, new_nb = (ctx + nw + page - 1) / page;
        need = new_nb - old_nb;
        if (new_nb > max_blocks) atomicOr(&s_bad, 2);
      }
    }
    s_scan[tid] = need;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if (tid < off) s_scan[tid] += s_scan[tid + off];
      __syncthreads();
    }
    if (tid == 0) s_carry += s_scan[0];
    __syncthreads();
  }
  const int total = s_carry;
  if (s_bad || total > nf) {
    if (tid == 0) state[1] = total > nf ? 1 : 2;     // (the reference meets the empty free list first, :78-79)
    if (store_ctx)
      for (int i = tid; i < batch; i += 256) store_ctx[i] = -1;
    return;
  }
  __syncthreads();
  if (tid == 0) s_carry = 0;
  __syncthreads();
  // pass 2: hand the blocks out in sequence order
  for (int base = 0; base < batch; base += 256) {
    const int i = base + tid;
    int need = 0, old_nb = 0;
    if (i < batch) {
      const int ctx = seq_lens[i];
      const int nw = new_lens ? new_lens[i] : new_uniform;
      const bool grows = ctx >= 0 && nw > 0;
      if (grows) {
        old_nb = (ctx + page - 1) / page;
        need = (ctx + nw + page - 1) / page - old_nb;
      }
      if (store_ctx) store_ctx[i] = grows ? ctx : -1;
    }
    s_scan[tid] = need;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {         // inclusive scan
      const int v = tid >= off ? s_scan[tid - off] : 0;
      __syncthreads();
      s_scan[tid] += v;
      __syncthreads();
    }
    const int before = s_carry + s_scan[tid] - need;   // blocks taken by sequences in front of i
    const int lo = nf - before - need;                 // this sequence's slice of the free list: [lo, lo + need)
    for (int j = 0; j < need; ++j) table[static_cast<long long>(i) * table_stride + old_nb + j] = free_blocks[lo + j];
    __syncthreads();
    if (tid == 255) s_carry += s_scan[255];
    __syncthreads();
  }
  if (tid == 0) {
    state[0] = nf - total;
    const int used = total_blocks - (nf - total);
    if (used > state[2]) state[2] = used;
  }
}

__global__ __launch_bounds__(256) void page_pool_advance_kernel(int32_t* seq_lens, const int32_t* new_lens, int new_uniform,
                                                                const int32_t* state, int batch) {
  if (state[1] != 0) return;                          // the extend step refused: lengths stay where they were
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= batch) return;
  const int ctx = seq_lens[i];
  const int nw = new_lens ? new_lens[i] : new_uniform;
  if (ctx >= 0 && nw > 0) seq_lens[i] = ctx + nw;
}

}  // namespace mojo

using namespace mojo;

extern "C" int mojo_hip_page_pool_extend(int32_t* block_table, int64_t block_table_stride, int64_t max_blocks_per_seq,
                                         const int32_t* seq_lens, const int32_t* new_lens, int64_t new_len_uniform,
                                         const int32_t* free_blocks, int32_t* pool_state, int32_t* store_ctx_out,
                                         int64_t batch, int64_t block_size, int64_t total_blocks,
                                         mojo_stream_t stream) {
  if (batch == 0) return MOJO_OK;
  MOJO_REQUIRE(block_table && seq_lens && free_blocks && pool_state, MOJO_EINVAL, "page_pool_extend: null pointer");
  MOJO_REQUIRE(batch > 0 && batch < (1LL << 31) && block_size > 0 && block_size < (1LL << 31) && max_blocks_per_seq >= 0 &&
                   max_blocks_per_seq < (1LL << 31) && total_blocks >= 0 && total_blocks < (1LL << 31) &&
                   new_len_uniform < (1LL << 31),
               MOJO_EINVAL, "page_pool_extend: bad shape");
  hipLaunchKernelGGL(page_pool_extend_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), block_table,
                     static_cast<long long>(block_table_stride), static_cast<int>(max_blocks_per_seq), seq_lens, new_lens,
                     static_cast<int>(new_len_uniform), free_blocks, pool_state, store_ctx_out, static_cast<int>(batch),
                     static_cast<int>(block_size), static_cast<int>(total_blocks));
  MOJO_CHECK_LAUNCH("page_pool_extend");
  return MOJO_OK;
}

extern "C" int mojo_hip_page_pool_advance(int32_t* seq_lens, const int32_t* new_lens, int64_t new_len_uniform,
                                          const int32_t* pool_state, int64_t batch, mojo_stream_t stream) {
  if (batch == 0) return MOJO_OK;
  MOJO_REQUIRE(seq_lens && pool_state && batch > 0 && batch < (1LL << 31), MOJO_EINVAL, "page_pool_advance: bad arguments");
  hipLaunchKernelGGL(page_pool_advance_kernel, dim3(static_cast<unsigned>(ceil_div(batch, 256))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), seq_lens, new_lens, static_cast<int>(new_len_uniform), pool_state,
                     static_cast<int>(batch));
  MOJO_CHECK_LAUNCH("page_pool_advance");
  return MOJO_OK;
}
