/* mojo_hip.h — C ABI of libmojo_hip.so, the MI355X (gfx950) kernel library behind the
 * `HIP<Op>` backend classes of mojo_opset_amd.
 *
 * The reference (XPU-Forces/mojo_opset) has no native code: each accelerated backend class
 * (e.g. `TTXPagedDecodeGQA.forward`, mojo_opset/backends/ttx/operators/attention.py:143-176)
 * calls a Python kernel launcher.  Every entry point below replaces one such launcher call; the
 * comment above it names the reference operator (golden `forward`) whose semantics it implements
 * and the accelerated call site it stands in for.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless the name ends in `_host`;
 *   - shapes/strides are int64 in ELEMENTS; the innermost dimension is always contiguous;
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued, never synchronised;
 *   - no allocation, no host sync, re-entrant; scratch comes from the caller (`workspace`);
 *   - return 0 on success, a negative MOJO_E* code otherwise; `mojo_hip_last_error()` returns a
 *     thread-local message for the last failure on the calling thread.
 */
#ifndef MOJO_HIP_H
#define MOJO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mojo_stream_t;

enum mojo_dtype { MOJO_F32 = 0, MOJO_F16 = 1, MOJO_BF16 = 2, MOJO_I8 = 3, MOJO_F8E4M3 = 4 };

enum mojo_status {
  MOJO_OK = 0,
  MOJO_EINVAL = -1,        /* malformed arguments (-> AssertionError / ValueError in the shim) */
  MOJO_EUNSUPPORTED = -2,  /* legal but not implemented for this shape/dtype (-> NotImplementedError) */
  MOJO_ELAUNCH = -3,       /* HIP launch failure (-> RuntimeError) */
  MOJO_EWORKSPACE = -4     /* workspace too small (-> RuntimeError) */
};

const char* mojo_hip_version(void);
const char* mojo_hip_last_error(void);

/* ---- Run-time switches (MOJO_HIP_* environment variables; the table is INTEGRATION.md section 5).  The reference reads its
 *      switches where it uses them (`MOJO_BACKEND` in `MojoOperator.__new__`, core/operator.py:45-47); a native library
 *      cannot afford getenv() per launch, so every switch is LATCHED the first time a launcher reads it and
 *      `mojo_hip_reload_env()` drops all latched values (the next call re-reads the environment).  Call it after changing a
 *      variable, with no operator call in flight.  `mojo_hip_switches()` writes "NAME=value" (space-separated, "NAME=" when
 *      unset) for every switch read so far into `buf` and returns their count.  `mojo_hip_last_launch()`: a short description
 *      of the kernel form the calling thread's last operator call launched ("decode_mfma:paired:nt", "gemm256:staged:KN",
 *      ...) — a debug / test query with which an A/B test proves that its two legs took different forms;
 *      `mojo_hip_launch_history(clear)`: the same notes of every launch of this thread since the last clear, '|'-separated
 *      (a composite operator — absorb projection, latent attention, output projection — leaves several).               */
void mojo_hip_reload_env(void);
int64_t mojo_hip_switches(char* buf, int64_t capacity);
const char* mojo_hip_last_launch(void);
const char* mojo_hip_launch_history(int clear);

/* ---- MojoStorePagedKVCache (core/operators/kv_cache.py:104-171; replaces store_paged_kv(),
 *      backends/ttx/operators/kv_cache.py:40-46).  Bit-exact copy, token-major -> head-major.
 *      plan rows = (src_token_start, dst_block_id, dst_block_offset, chunk_len) int32.          */
int mojo_hip_store_paged_kv_plan(const void* key_states, const void* value_states,
                                 void* key_cache, void* value_cache,
                                 const int32_t* plan, int64_t num_chunks,
                                 int64_t num_tokens, int64_t num_kv_heads, int64_t head_dim,
                                 int64_t num_blocks, int64_t block_size, int64_t elt_bytes,
                                 int64_t src_token_stride, int64_t src_head_stride,
                                 int64_t cache_block_stride, int64_t cache_head_stride,
                                 int64_t cache_token_stride, mojo_stream_t stream);

/*      Same op, legacy arguments (block_table, cu_q_lens|NULL, context_kv_lens): the plan of
 *      build_paged_kv_chunk_metadata (kv_cache.py:33-101) is evaluated per token on the device, so
 *      there is no host sync.  cu_q_lens == NULL selects decode mode (one token per sequence).  */
int mojo_hip_store_paged_kv_layout(const void* key_states, const void* value_states,
                                   void* key_cache, void* value_cache,
                                   const int32_t* block_table, int64_t block_table_stride,
                                   int64_t max_blocks_per_seq,
                                   const int32_t* cu_q_lens, const int32_t* context_kv_lens,
                                   int64_t batch, int64_t num_tokens, int64_t num_kv_heads,
                                   int64_t head_dim, int64_t num_blocks, int64_t block_size,
                                   int64_t elt_bytes, int64_t src_token_stride,
                                   int64_t src_head_stride, int64_t cache_block_stride,
                                   int64_t cache_head_stride, int64_t cache_token_stride,
                                   mojo_stream_t stream);

/* ---- MojoSwiGLU (core/operators/activation.py:38-66; replaces swiglu_fwd,
 *      backends/ttx/operators/activation.py).  Flat contiguous tensors of n elements.           */
int mojo_hip_swiglu(const void* gate, const void* up, void* out, int64_t n, int dtype,
                    float swiglu_limit, mojo_stream_t stream);
/*      Row-strided form: gate / up / out are [rows, cols] views with row strides in elements (the two halves of a fused
 *      [rows, 2*cols] projection, core/operators/moe.py:441-445).                                              */
int mojo_hip_swiglu_rows(const void* gate, const void* up, void* out, int64_t rows, int64_t cols,
                         int64_t ld_gate, int64_t ld_up, int64_t ld_out, int dtype, float swiglu_limit,
                         mojo_stream_t stream);

/* ---- MojoResidualAddRMSNorm / MojoRMSNorm (core/operators/normalization.py:308-362, :71-111;
 *      replaces fused_add_rmsnorm / rmsnorm launchers, backends/ttx/operators/normalization.py:35-47).
 *      residual == NULL: plain RMSNorm.  sum_out may be NULL (norm_pos="post").
 *      sum = round_dtype(hidden + residual); normed = round_dtype(sum * rsqrt(mean(sum^2)+eps) * w).
 *      normed_out may be the same buffer as hidden: MojoRMSNormInplace(inplace=True)
 *      (experimental/operators/normalization.py:95-140).                                            */
int mojo_hip_residual_add_rmsnorm(const void* hidden, const void* residual, const void* weight,
                                  void* normed_out, void* sum_out, int64_t rows, int64_t dim,
                                  int dtype, float eps, mojo_stream_t stream);

/* ---- MojoApplyRoPE (core/operators/position_embedding.py:98-175; replaces rope_fwd,
 *      backends/ttx/operators/position_embedding.py).  q/k are addressed as [B][T][N][D] through
 *      explicit strides (so head-first and token-first layouts are read in place); cos/sin are
 *      fp32 [.., rope_dim] addressed as [B][T] with strides (cos_b_stride may be 0).             */
int mojo_hip_apply_rope(const void* q, const void* k, void* q_out, void* k_out,
                        const float* cos, const float* sin,
                        int64_t batch, int64_t tokens, int64_t q_heads, int64_t k_heads,
                        int64_t head_dim, int64_t rope_dim,
                        const int64_t q_strides[3], const int64_t k_strides[3],
                        const int64_t qo_strides[3], const int64_t ko_strides[3],
                        int64_t cos_b_stride, int64_t cos_t_stride, int dtype,
                        mojo_stream_t stream);

/* ---- Block allocator of the paged KV cache (replaces the host loop of PagedDummyCache.update,
 *      modeling/qwen3/mojo_qwen3_dense.py:84-123: `.item()` per sequence + Python slices of the free list).
 *      extend: every sequence i with seq_lens[i] >= 0 and new_i > 0 (new_lens[i], or new_len_uniform when new_lens is NULL)
 *      gains ceil((len+new)/page) - ceil(len/page) blocks, taken from the END of free_blocks[0 .. num_free) sequence by
 *      sequence, written to block_table[i, ceil(len/page) ...] — the tables the reference would build, entry for entry.
 *      pool_state int32[4] = {num_free, error, high-water of used blocks, 0}; when the blocks (error 1) or the table
 *      columns (error 2) do not suffice the launch changes nothing and sets `error` (sticky; the host resets it).
 *      store_ctx_out (nullable, int32 [batch]) receives the `context_kv_lens` the KV store must be called with: the
 *      row's length before the append, or -1 (= skip, kv_cache.py:56-74) for rows that append nothing / a refused launch.
 *      advance: seq_lens[i] += new_i for the same rows (no-op while `error` is set).  No host sync: capturable.        */
int mojo_hip_page_pool_extend(int32_t* block_table, int64_t block_table_stride, int64_t max_blocks_per_seq,
                              const int32_t* seq_lens, const int32_t* new_lens, int64_t new_len_uniform,
                              const int32_t* free_blocks, int32_t* pool_state, int32_t* store_ctx_out,
                              int64_t batch, int64_t block_size, int64_t total_blocks,
                              mojo_stream_t stream);
int mojo_hip_page_pool_advance(int32_t* seq_lens, const int32_t* new_lens, int64_t new_len_uniform,
                               const int32_t* pool_state, int64_t batch, mojo_stream_t stream);

/* ---- MojoRotaryEmbedding (core/operators/position_embedding.py:9-95; replaces rot_pos_embed,
 *      backends/ttx/kernels/ilu/rope.py:634-675 — a host loop with .item() per sequence).
 *      mode 0: positions = position_ids[i];  mode 1: positions = i (padded prefill);
 *      mode 2: positions from cu_q_lens (+ total_seq_lens or NULL), tokens outside every
 *              sequence get position -1 (python-style: last table row / angle -inv_freq).
 *      cos_table/sin_table == NULL: compute cos/sin(pos * inv_freq) * scaling on the fly.        */
int mojo_hip_rotary_embedding(float* cos_out, float* sin_out, int64_t n_pos, int64_t rope_dim,
                              int mode, const int32_t* position_ids, const int32_t* cu_q_lens,
                              const int32_t* total_seq_lens, int64_t batch,
                              const float* cos_table, const float* sin_table, int64_t table_len,
                              const float* inv_freq, float attention_scaling,
                              mojo_stream_t stream);

/* ---- MojoPagedDecodeGQA (core/operators/attention.py:113-232; replaces paged_attention_decode,
 *      backends/ttx/operators/attention.py:166-174).  Split-KV flash decoding; the q-heads of one
 *      kv-head share every K/V load.  layout_abab: 0 = "AABB", 1 = "ABAB".
 *      max_seq_len_hint <= 0: derive the split count from max_blocks_per_seq * block_size.  Lengths above
 *      min(hint, block_size * max_blocks_per_seq) are truncated to that capacity (nothing is indexed past what
 *      the launch was sized for).  leave_empty_rows: 0 = rows with total_seq_lens <= 0 are written as zeros (the
 *      eager golden, attention.py:184-185); 1 = such rows of `out` are left untouched — the graph-replay contract
 *      of padded rows (tests/accuracy/operators/test_attention.py:318-353; the reference's ILU kernel switches on
 *      `not is_current_stream_capturing()`, backends/ttx/kernels/ilu/flash_attention.py:816,425-428).             */
int64_t mojo_hip_paged_decode_gqa_workspace_bytes(int64_t batch, int64_t q_heads, int64_t kv_heads,
                                                  int64_t head_dim, int64_t block_size,
                                                  int64_t max_blocks_per_seq,
                                                  int64_t max_seq_len_hint);
int mojo_hip_paged_decode_gqa(const void* query, const void* key_cache, const void* value_cache,
                              const int32_t* total_seq_lens, const int32_t* block_tables,
                              void* out, void* workspace, int64_t workspace_bytes,
                              int64_t batch, int64_t q_heads, int64_t kv_heads, int64_t head_dim,
                              int64_t block_size, int64_t max_blocks_per_seq,
                              int64_t block_table_stride, int64_t cache_block_stride,
                              int64_t cache_head_stride, int64_t cache_token_stride,
                              int64_t max_seq_len_hint, float softmax_scale, int layout_abab,
                              int leave_empty_rows, int dtype, mojo_stream_t stream);

/* ---- MojoGroupGemm (core/operators/gemm.py:59-124; replaces m_grouped_matmul,
 *      backends/ttx/operators/gemm.py:51-92).  out[rows of g] = input[rows of g] @ W[g];
 *      group_list = per-group ROW COUNTS on the device (int32, or int64 when group_list_is_i64);
 *      offsets are prefix-summed on the device (no host sync, graph-capturable).
 *      trans_weight = 0: weight [G,K,N];  1: weight [G,N,K].  bf16/fp16 with K % 64 == 0 run on the
 *      MFMA kernel, everything else (fp32, odd K/N) on a generic kernel.                            */
int64_t mojo_hip_group_gemm_workspace_bytes(int64_t num_groups);
int mojo_hip_group_gemm(const void* input, const void* weight, void* out, const void* group_list,
                        int group_list_is_i64, int64_t m_total, int64_t k, int64_t n,
                        int64_t num_groups, int trans_weight, int dtype, void* workspace,
                        int64_t workspace_bytes, mojo_stream_t stream);

/*      MojoExperts' first projection with the SwiGLU fused into the epilogue (mojo_opset/core/operators/moe.py:402-449:
 *      GroupGemm -> SwiGLU -> GroupGemm): weight [G, 2*inter, K] (trans_weight) or [G, K, 2*inter], columns [gate | up];
 *      out [m_total, inter] = round(round(silu(round(gate))) * round(up)) — the roundings of the two-op golden path — so
 *      the [m_total, 2*inter] product never goes to HBM.  bf16 / fp16, inter % 128 == 0 and the 256x256 MFMA kernel's
 *      layout preconditions; MOJO_EUNSUPPORTED otherwise — and where the library's time model prefers the unfused route (the
 *      product on 128-row tiles: few small experts, groups of about a hundred rows) — callers fall back to group_gemm +
 *      swiglu_rows (same bits).                                                                                         */
int mojo_hip_group_gemm_swiglu(const void* input, const void* weight, void* out, const void* group_list,
                               int group_list_is_i64, int64_t m_total, int64_t k, int64_t inter,
                               int64_t num_groups, int trans_weight, int dtype, void* workspace,
                               int64_t workspace_bytes, mojo_stream_t stream);

/*      Same kernel with explicit strides (elements): input row stride lda, output row stride ldc, weight
 *      element (g,k,n) at weight + g*w_group_stride + k*w_k_stride + n*w_n_stride, and optional row maps
 *      {rc, ml, off, mul} (NULL = identity): logical row m reads input row (m/rc)*ml + off + (m%rc)*mul and
 *      writes the output row given by c_map.  Used by the MLA ops to multiply by the two halves of kv_b_proj in
 *      place, one group per head, reading/writing token-major [T,H,*] tensors directly (rc=T, ml=1, mul=H).
 *      group_list == NULL means num_groups equal groups of m_total / num_groups rows (no prefix pass).      */
int mojo_hip_group_gemm_strided(const void* input, const void* weight, void* out, const void* group_list,
                                int group_list_is_i64, int64_t m_total, int64_t k, int64_t n,
                                int64_t num_groups, int64_t lda, int64_t ldc, int64_t w_group_stride,
                                int64_t w_k_stride, int64_t w_n_stride, const int64_t a_map[4],
                                const int64_t c_map[4], int dtype, void* workspace,
                                int64_t workspace_bytes, mojo_stream_t stream);

/* ---- MojoPagedDecodeMLA / MojoPagedPrefillMLA (experimental/operators/attention.py:131-227, :325-447; the
 *      reference has no accelerated kernel for either).  Attention over the COMPRESSED cache in the
 *      weight-absorbed form: q_lat [Tq,H,r(+rope)] = [q_nope @ W_kn (| q_rope)], o_lat [Tq,H,r] = sum_s p c_kv[s].
 *      q_lat rows are q_lat_stride elements apart; the rope part is read from q_rope (row stride q_rope_stride)
 *      when q_rope != NULL, else from q_lat columns r.. .
 *      decode : total_seq_lens != NULL, cu_q_lens == NULL, Tq == batch (one token per sequence)
 *      prefill: cu_q_lens != NULL (cu_total_seq_lens optional), token t sees keys 0 .. kv_len-q_len+t
 *      attn_sink: optional fp32 [H] extra softmax logit (probability mass only).                           */
int64_t mojo_hip_mla_latent_attn_workspace_bytes(int64_t q_tokens, int64_t heads, int64_t kv_lora_rank,
                                                 int64_t max_kv_len);
int mojo_hip_mla_latent_attn(const void* q_lat, int64_t q_lat_stride, const void* q_rope, int64_t q_rope_stride,
                             const void* ckv_cache, const void* kpe_cache,
                             const int32_t* total_seq_lens, const int32_t* cu_q_lens,
                             const int32_t* cu_total_seq_lens, const int32_t* block_tables,
                             const float* attn_sink, void* o_lat, void* workspace,
                             int64_t workspace_bytes, int64_t q_tokens, int64_t batch, int64_t heads,
                             int64_t kv_lora_rank, int64_t rope_dim, int64_t block_size,
                             int64_t max_blocks_per_seq, int64_t block_table_stride,
                             int64_t ckv_block_stride, int64_t ckv_token_stride,
                             int64_t kpe_block_stride, int64_t kpe_token_stride, int64_t max_kv_len,
                             float softmax_scale, int dtype, mojo_stream_t stream);

/* ---- dense GEMM used by the GEMM+collective operators (core/operators/compute_with_comm.py:12-24,
 *      `_gemm`): out[M,N] = input[M,K] @ W (+ bias).  W element (k,n) at weight + k*w_k_stride +
 *      n*w_n_stride (one of the two strides is 1).  bias (optional, [N]) is added after the product
 *      has been rounded to the storage type, as the golden's two separate ops do.  Decode-sized M with
 *      K-major weights streams the weight (gemm_skinny.hip), cut along K into fp32 slabs in the workspace
 *      when there are few column tiles (slabs summed in fixed order).                                    */
int64_t mojo_hip_gemm_workspace_bytes(int64_t m, int64_t k, int64_t n);
int mojo_hip_gemm(const void* input, const void* weight, const void* bias, void* out, int64_t m,
                  int64_t k, int64_t n, int64_t lda, int64_t ldc, int64_t w_k_stride,
                  int64_t w_n_stride, int dtype, void* workspace, int64_t workspace_bytes,
                  mojo_stream_t stream);

/*      Decode-sized fusions of a decoder layer's dense chain (the op sequence of core/operators/moe.py:402-449 and of
 *      modeling/*: linear -> MojoSwiGLU, linear -> MojoResidualAddRMSNorm).  Results are bit-identical to the separate
 *      calls (mojo_hip_gemm, mojo_hip_swiglu_rows, mojo_hip_residual_add_rmsnorm): the fused kernels round where
 *      those round (gemm_swiglu never cuts K; where the separate projection would, the fp32 sums differ in order).
 *      gemm_swiglu: weight = [gate | up] rows, [2*inter, k] K-major (row stride w_n_stride); out [m, inter] =
 *      silu(x @ gate^T) * (x @ up^T).  m <= 64: ONE launch, wave units dealt evenly over the CUs; otherwise the
 *      product goes through the workspace.                                                                        */
int64_t mojo_hip_gemm_swiglu_workspace_bytes(int64_t m, int64_t k, int64_t inter);
int mojo_hip_gemm_swiglu(const void* input, const void* weight, void* out, int64_t m, int64_t k,
                         int64_t inter, int64_t lda, int64_t ldc, int64_t w_n_stride, int dtype,
                         void* workspace, int64_t workspace_bytes, mojo_stream_t stream);
/*      gemm_residual_rmsnorm: normed_out = RMSNorm(x @ W (+ bias) + residual) * norm_weight, sum_out (may be NULL) =
 *      the sum, gemm_out (may be NULL) = the product.  With a K split (decode-sized m, few column tiles) the
 *      fp32 K-slice slabs are summed by the norm kernel itself: the product makes no round trip through HBM.      */
int64_t mojo_hip_gemm_residual_rmsnorm_workspace_bytes(int64_t m, int64_t k, int64_t n);
int mojo_hip_gemm_residual_rmsnorm(const void* input, const void* weight, const void* bias,
                                   const void* residual, const void* norm_weight, void* normed_out,
                                   void* sum_out, void* gemm_out, int64_t m, int64_t k, int64_t n,
                                   int64_t lda, int64_t w_k_stride, int64_t w_n_stride, int dtype,
                                   float eps, void* workspace, int64_t workspace_bytes,
                                   mojo_stream_t stream);

/*      qkv_rope_store: one decode step's fused QKV projection (weight [(Hq + 2 Hkv) * D, k] K-major, heads in q | k | v
 *      order) -> MojoApplyRoPE on q and k (rotate-half over the whole head; cos / sin fp32 [batch, D], one row per
 *      sequence: core/operators/position_embedding.py) -> q_out [batch, Hq, D] -> MojoStorePagedKVCache in decode mode
 *      (one token per sequence at position context_kv_lens[b]; kv_cache.py:33-101).  With a K split the slice sums are
 *      taken by the kernel that rotates and stores: two launches for the chain's five.  Same bits as the separate calls. */
int64_t mojo_hip_qkv_rope_store_workspace_bytes(int64_t m, int64_t k, int64_t n);
int mojo_hip_qkv_rope_store(const void* input, const void* weight, const void* bias, const float* cos,
                            const float* sin, int64_t cos_sin_row_stride, void* q_out, void* key_cache,
                            void* value_cache, const int32_t* block_table, int64_t block_table_stride,
                            int64_t max_blocks_per_seq, const int32_t* context_kv_lens, int64_t batch,
                            int64_t k, int64_t q_heads, int64_t kv_heads, int64_t head_dim, int64_t lda,
                            int64_t w_n_stride, int64_t num_blocks, int64_t block_size,
                            int64_t cache_block_stride, int64_t cache_head_stride,
                            int64_t cache_token_stride, int dtype, void* workspace,
                            int64_t workspace_bytes, mojo_stream_t stream);

/*      Same with row maps {rc, ml, off} (NULL or rc == 0: identity): logical row m reads A row
 *      (m / rc) * ml + off + m % rc, and likewise for the C row it writes.  One launch can thus consume or
 *      produce the "c-th sub-chunk of every rank" view of the chunked reduce-scatter / all-gather pipelines. */
int mojo_hip_gemm_rowmap(const void* input, const void* weight, const void* bias, void* out, int64_t m,
                         int64_t k, int64_t n, int64_t lda, int64_t ldc, int64_t w_k_stride,
                         int64_t w_n_stride, const int64_t a_map[3], const int64_t c_map[3], int dtype,
                         void* workspace, int64_t workspace_bytes, mojo_stream_t stream);

/* ---- MojoQuantGemm (core/operators/gemm.py:127-231; the reference's accelerated int8 kernel is
 *      backends/ttx/kernels/ilu/int8_gemm.py:18-66).  out = (A_q @ W_q) * input_scale[m] * weight_scale[n]
 *      input [M,K] and weight ([K,N], or [N,K] when trans_weight) are int8 (MOJO_I8) or OCP fp8-e4m3
 *      (MOJO_F8E4M3, an extension: no reference implementation, parity unpinned); input_scale fp32 [M];
 *      weight_scale bf16 [N]; out_dtype in {f32, f16, bf16}.  Decode-sized M is cut along K into slices
 *      whose raw accumulators go to the workspace and are summed in a fixed order (deterministic).          */
int64_t mojo_hip_quant_gemm_workspace_bytes(int64_t m, int64_t k, int64_t n);
int mojo_hip_quant_gemm(const void* input, const void* weight, const float* input_scale,
                        const void* weight_scale, void* out, int64_t m, int64_t k, int64_t n,
                        int trans_weight, int quant_dtype, int out_dtype, void* workspace,
                        int64_t workspace_bytes, mojo_stream_t stream);

/* ---- MojoPagedPrefillGQA (core/operators/attention.py:315-451; replaces paged_attention_prefill,
 *      backends/ttx/operators/attention.py).  Packed var-len queries [T,Hq,D], causal with offset
 *      kv_len - q_len; cu_total_seq_lens may be NULL (kv_len = q_len).  `out` is fully written: rows of
 *      empty sequences / padding tokens are zeroed.  max_q_len_hint <= 0: use total_tokens as the bound.
 *      Launches of few, long blocks (a chunked prefill against a long cache) are cut along the keys: every (query block,
 *      kv head, sequence) workgroup becomes several that walk a slice of its key tiles and leave fp32 partials in the
 *      workspace, combined by a merge kernel in slice order.  workspace_bytes() == 0: the launch is not split (workspace may
 *      be NULL); a workspace that is too small runs the launch unsplit.  max_kv_len_hint <= 0: block_size * max_blocks.   */
int64_t mojo_hip_paged_prefill_gqa_workspace_bytes(int64_t total_tokens, int64_t batch, int64_t q_heads,
                                                   int64_t kv_heads, int64_t head_dim, int64_t block_size,
                                                   int64_t max_blocks_per_seq, int64_t max_q_len_hint,
                                                   int64_t max_kv_len_hint);
int mojo_hip_paged_prefill_gqa(const void* query, const void* key_cache, const void* value_cache,
                               const int32_t* cu_q_lens, const int32_t* cu_total_seq_lens,
                               const int32_t* block_tables, void* out, int64_t total_tokens,
                               int64_t batch, int64_t q_heads, int64_t kv_heads, int64_t head_dim,
                               int64_t block_size, int64_t max_blocks_per_seq,
                               int64_t block_table_stride, int64_t cache_block_stride,
                               int64_t cache_head_stride, int64_t cache_token_stride,
                               int64_t max_q_len_hint, int64_t max_kv_len_hint, float softmax_scale,
                               int layout_abab, int dtype, void* workspace, int64_t workspace_bytes,
                               mojo_stream_t stream);

/* ---- MoE routing either side of the grouped GEMM (core/operators/moe.py).
 *      gating (:299-316): softmax(hidden.float() @ gate_weight [hidden, E] fp32) over all experts, top-k in descending
 *      order (ties: lowest expert id), gates renormalised to sum 1.  top_k <= min(E, 64), E <= 1024.  With many
 *      experts and tokens (16-bit activations) the logits run on MFMA as x @ w_hi + x @ w_lo, w = w_hi + w_lo split
 *      into the activation dtype (error ~2^-16 relative); that route takes its scratch from the workspace.
 *      dispatch (:344-400): stable counting sort of the tokens*top_k routing slots by expert id (the reference leaves
 *      the order inside a bucket undefined; this one keeps flat-slot order, so the op is deterministic).  Outputs:
 *      sorted_hidden [slots, H], tokens_per_expert int32 [E], sorted_gates fp32 [slots] (viewed [slots,1]),
 *      token_indices int32 [slots].  Ids outside [0, E) are dropped.
 *      combine (:687-716): out[t] = sum of expert_outputs[j] (* sorted_gates[j]; NULL = no gates) over the rows j with
 *      token_indices[j] == t, fp32 from zero in ascending j, product and sum rounded separately — bit-identical to the
 *      reference's fp32 scatter-add.  Tokens nobody routes to are zero.                                         */
int64_t mojo_hip_moe_gating_workspace_bytes(int64_t tokens, int64_t hidden_size, int64_t num_experts, int dtype);
int mojo_hip_moe_gating(const void* hidden, const float* gate_weight, int32_t* top_k_indices, float* top_k_gates,
                        int64_t tokens, int64_t hidden_size, int64_t num_experts, int64_t top_k, int dtype,
                        void* workspace, int64_t workspace_bytes, mojo_stream_t stream);
int64_t mojo_hip_moe_dispatch_workspace_bytes(int64_t slots, int64_t num_experts);
int mojo_hip_moe_dispatch(const void* hidden, const float* top_k_gates, const int32_t* top_k_indices,
                          void* sorted_hidden, int32_t* tokens_per_expert, float* sorted_gates,
                          int32_t* token_indices, int64_t tokens, int64_t hidden_size, int64_t top_k,
                          int64_t num_experts, int dtype, void* workspace, int64_t workspace_bytes,
                          mojo_stream_t stream);
int64_t mojo_hip_moe_combine_workspace_bytes(int64_t tokens, int64_t rows);
int mojo_hip_moe_combine(const void* expert_outputs, const float* sorted_gates, const int32_t* token_indices,
                         void* out, int64_t tokens, int64_t rows, int64_t hidden_size, int dtype,
                         void* workspace, int64_t workspace_bytes, mojo_stream_t stream);

/* ---- Per-token activation quantisers feeding MojoQuantGemm.
 *      dynamic_quant (core/operators/quantize.py:153-169): y = x.float() [* inv_smooth_scale[K] fp32, nullable];
 *      scale = max(amax|y|, 1e-12) / 127, 1.0 where that is < 1e-6; q = clamp(round_half_even(y / scale), -128, 127).
 *      residual_add_rmsnorm_quant (core/operators/normalization.py:493-526): s = hidden + residual (rounded to the
 *      input dtype; residual nullable), y = rms_norm(s.float(), weight fp32, eps) [* smooth_scale fp32, nullable] kept
 *      in fp32, scale = max(amax|y|, 1e-12) / q_max (127 or 448), q = clamp(round_half_even(y / scale), q_min, q_max)
 *      as int8 or as float8_e4m3fn of that integer value.  out_sum (T, nullable) receives s (norm_pos = "pre"),
 *      out_normed (fp32, nullable) the normed tensor before smoothing (norm_pos = "post").  out_scale is fp32 [rows]. */
int mojo_hip_dynamic_quant(const void* input, const float* inv_smooth_scale, void* out_q, float* out_scale,
                           int64_t rows, int64_t dim, int dtype, mojo_stream_t stream);
int mojo_hip_residual_add_rmsnorm_quant(const void* hidden, const void* residual, const float* weight,
                                        const float* smooth_scale, void* out_q, void* out_sum, float* out_normed,
                                        float* out_scale, int64_t rows, int64_t dim, int dtype, int quant_dtype,
                                        float q_min, float eps, mojo_stream_t stream);

/* ---- MojoStorePagedMLAKVCache (experimental/operators/kv_cache.py:13-106): bit-exact copy of the new latent
 *      tokens compressed_kv_states [T, r] and k_pe_states [T, rope] into compressed_kv_cache [N,1,page,r] and
 *      k_pe_cache [N,1,page,rope] at positions context_kv_lens[b].. (cu_q_lens == NULL: one token per sequence).
 *      Per sequence the reference stops at the first negative page id and skips sequences whose first table entry or
 *      context length is negative; evaluated per token on the device, no host sync.  Strides in elements.        */
int mojo_hip_store_paged_mla_kv(const void* compressed_kv_states, const void* k_pe_states,
                                void* compressed_kv_cache, void* k_pe_cache, const int32_t* block_table,
                                int64_t block_table_stride, int64_t max_blocks_per_seq,
                                const int32_t* cu_q_lens, const int32_t* context_kv_lens, int64_t batch,
                                int64_t num_tokens, int64_t kv_lora_rank, int64_t rope_dim, int64_t num_blocks,
                                int64_t block_size, int64_t elt_bytes, int64_t ckv_src_token_stride,
                                int64_t kpe_src_token_stride, int64_t ckv_block_stride, int64_t ckv_token_stride,
                                int64_t kpe_block_stride, int64_t kpe_token_stride, mojo_stream_t stream);

/* ---- MojoPagedPrefillMLA in the golden's own formulation (experimental/operators/attention.py:405-447):
 *      mla_unpage: flat row cu[b] + t <- cache[table[b, t / page], 0, t % page, :] for t < kv_len_b (cu = cu_total_seq_lens, or
 *      cu_q_lens when that is NULL), for the compressed latent [.., kv_lora_rank] and the positional key [.., rope_dim];
 *      strides in elements; max_tokens_per_seq only sizes the grid; flat rows are relative to cu[0] (the pointers may address a
 *      slice of the batch) and total_keys_out (nullable, int32 [1]) receives min(cu[batch] - cu[0], capacity_rows), the
 *      decompression GEMM's device-side row count.  capacity_rows = rows the flat buffers hold: the lengths live on the device
 *      and the host sizes the buffers without a sync, so a row at or past the capacity is neither written by mla_unpage nor
 *      read by mla_prefill_attn (a sequence longer than max_tokens_per_seq is cut at that bound, per sequence, by both calls).  The decompression kv = c_kv @ kv_b_proj^T is
 *      mojo_hip_group_gemm with one group whose row count is the device-side total.
 *      mla_prefill_attn: causal flash attention per head over K = [kv[t, h, :nope] | k_pe[t, :]] and V = kv[t, h, nope:],
 *      kv [T_kv, heads, nope + v_dim] and k_pe [T_kv, rope] contiguous, sequence b's keys at rows cu[b] ..; optional fp32
 *      sink logit per head in the softmax denominator; rows of empty sequences read as zeros, and so do the padding rows
 *      behind cu_q_lens[batch] when zero_padding_rows is set (the last slice of a batch processed in slices).
 *      round_scaled_scores: 0 = scores rounded to the storage type, scaled in fp32 (the prefill golden, :425); 1 = the scaled
 *      scores are rounded to the storage type once more (the DECODE golden multiplies storage-type tensors, :215) — used by
 *      the golden-rounding decode route of MojoPagedDecodeMLA (one query token per sequence).  mla_prefill_supported: 1 when (nope, rope, v_dim, dtype) has an instantiation.                  */
int mojo_hip_mla_prefill_supported(int64_t nope, int64_t rope, int64_t v_dim, int dtype);
int mojo_hip_mla_unpage(const void* compressed_kv_cache, const void* k_pe_cache, void* ckv_out, void* kpe_out,
                        const int32_t* cu_q_lens, const int32_t* cu_total_seq_lens, const int32_t* block_tables,
                        int64_t block_table_stride, int64_t max_blocks_per_seq, int64_t batch,
                        int64_t kv_lora_rank, int64_t rope_dim, int64_t block_size, int64_t elt_bytes,
                        int64_t ckv_block_stride, int64_t ckv_token_stride, int64_t kpe_block_stride,
                        int64_t kpe_token_stride, int64_t max_tokens_per_seq, int64_t capacity_rows,
                        int32_t* total_keys_out, mojo_stream_t stream);
int mojo_hip_mla_prefill_attn(const void* query, const void* kv_decompressed, const void* k_pe_flat,
                              const float* attn_sink, void* out, const int32_t* cu_q_lens,
                              const int32_t* cu_total_seq_lens, int64_t total_tokens, int64_t batch,
                              int64_t heads, int64_t head_begin, int64_t head_count, int64_t nope, int64_t rope,
                              int64_t v_dim, int64_t max_q_len,
                              int64_t max_tokens_per_seq, int64_t capacity_rows, float softmax_scale, int round_scaled_scores,
                              int zero_padding_rows, int dtype, mojo_stream_t stream);

/* ---- Direct reduce-scatter / all-gather over HIP-IPC peer buffers: the exchange step of MojoGemmAllReduce /
 *      MojoGemmReduceScatter without a ring (core/operators/compute_with_comm.py:57-116, :264-340; role of
 *      runtime/comm_context.py:107-153 `allocate_peer_mem` + the pull-and-add loops of
 *      backends/ttx/kernels/npu/a2/gemm_allreduce.py:85-145 and gemm_reduce_scatter.py:108-156).
 *      Set-up (the only entry points of this library that allocate or synchronise): each rank allocates ONE buffer
 *      [2 x capacity data | 4 KiB | ctrl words], exports a 64-byte handle, opens every peer's handle.
 *      peer_data[r] / peer_flags[r] = this process's pointer to rank r's data area / control words (own entry = local).
 *      Exchange steps only enqueue kernels.  epoch grows by one per operator call (never reset); flag kind 0 = "partial
 *      product of (rank, chunk) is in memory", kind 1 = "reduced share of (rank, chunk) is in memory".
 *      reduce: dst[rows, n] = sum over ranks of the contiguous [rows, n] block at src_offset_bytes of every rank's data
 *      area (fp32 sum in rank order, one rounding), after waiting for every rank's kind-0 flag of `chunk`; write_back = 1
 *      also stores the result over the own block and raises this rank's kind-1 flag at every peer when the launch is done.
 *      gather: for every other rank p, after its kind-1 flag: rows [rows*p/ws, rows*(p+1)/ws) of the [rows, n] chunk at
 *      chunk_offset_bytes of rank p's data area -> the same rows of dst.
 *      Every wait is bounded (MOJO_HIP_PEER_TIMEOUT_MS, default 20 s; mojo_hip_peer_set_timeout_ms overrides it for the
 *      steps enqueued afterwards and returns the previous override, 0 = none): on expiry the sticky error word is set, the
 *      affected output is filled with NaN and the grid drains; mojo_hip_peer_error reads (and clears) the word.        */
int64_t mojo_hip_peer_ctrl_bytes(void);
int64_t mojo_hip_peer_max_ranks(void);
int64_t mojo_hip_peer_max_chunks(void);
int64_t mojo_hip_peer_handle_bytes(void);
int64_t mojo_hip_peer_set_timeout_ms(int64_t milliseconds);
int mojo_hip_peer_alloc(void** ptr_out, int64_t bytes, int uncached);
int mojo_hip_peer_free(void* ptr);
int mojo_hip_peer_export(void* ptr, void* handle_out);
int mojo_hip_peer_open(const void* handle, void** ptr_out);
int mojo_hip_peer_close(void* ptr);
int mojo_hip_peer_error(void* local_flags, int clear, int32_t* error_out);
/*      Captured mode (HIP-graph replay).  Every exchange step below takes the call's epoch; epoch 0 means "the epoch
 *      word of this rank's control area", which mojo_hip_peer_begin advances once per call after waiting (bounded) until
 *      every peer has raised flag kind 2 ("finished reading this rank's data") for the previous call.  A captured call is
 *      begin -> steps with epoch 0 on ONE data area (no parity halves) -> mojo_hip_peer_signal(kind 2, chunk 0, epoch 0).  */
int mojo_hip_peer_begin(void* const* peer_data, void* const* peer_flags, int64_t world, int64_t rank,
                        mojo_stream_t stream);
/* set-up check: `bytes` of a peer's buffer (opened mapping) copied to host memory by the runtime; synchronises */
int mojo_hip_peer_peek(const void* peer_ptr, void* host_out, int64_t bytes);
int mojo_hip_peer_signal(void* const* peer_data, void* const* peer_flags, int64_t world, int64_t rank,
                         int kind, int64_t chunk, uint32_t epoch, mojo_stream_t stream);
int mojo_hip_peer_reduce(void* const* peer_data, void* const* peer_flags, int64_t world, int64_t rank,
                         int64_t chunk, uint32_t epoch, int64_t src_offset_bytes, int64_t rows, int64_t n,
                         void* dst, int64_t ld_dst, int write_back, int dtype, mojo_stream_t stream);
int mojo_hip_peer_gather(void* const* peer_data, void* const* peer_flags, int64_t world, int64_t rank,
                         int64_t chunk, uint32_t epoch, int64_t chunk_offset_bytes, int64_t rows, int64_t n,
                         void* dst, int64_t ld_dst, int dtype, mojo_stream_t stream);
/*      pull (the all-gather of MojoAllGatherGemm, compute_with_comm.py:119-184): slot p of dst (dst + p * dst_stride_bytes)
 *      <- `bytes` at src_offset_bytes of rank p's data area, for every rank (include_self) or every other rank, each after
 *      rank p's flag (kind, flag_chunk) reached `epoch` (the own block needs no wait).                                   */
int mojo_hip_peer_pull(void* const* peer_data, void* const* peer_flags, int64_t world, int64_t rank, int kind,
                       int64_t flag_chunk, uint32_t epoch, int64_t src_offset_bytes, int64_t bytes, void* dst,
                       int64_t dst_stride_bytes, int include_self, mojo_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MOJO_HIP_H */
