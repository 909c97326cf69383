// MojoPagedPrefillMLA in the golden's OWN formulation (experimental/operators/attention.py:405-447) for gfx950:
//   1. un-page the compressed latent and the positional key of every sequence into contiguous rows        (this file)
//   2. decompress: kv[t, h, :] = c_kv[t, :] @ kv_b_proj[h]^T, rounded to the storage type, as the golden does  (grouped GEMM)
//   3. flash attention per head over K = [k_nope | k_pe] (D_qk = nope + rope) and V (D_v)                   (this file)
// The weight-absorbed form (csrc/mla_attn.hip: attention over the 576-wide latent, every query token against the whole
// compressed cache) spends 2 * (2r + rope) = 2176 FLOPs per (query, key, head) pair at DeepSeek-V3 dimensions; this form
// spends 2 * (192 + 128) = 640 plus one decompression GEMM over the keys — 3.4x fewer FLOPs once a key is visible to many
// queries, and it shares the golden's rounding points (decompressed K/V in bf16, probabilities in bf16), so the parity
// band against the golden is the GQA kernel's 2e-2 instead of the golden's own distance from its definition.
//
// Attention kernel: one 256-thread workgroup = one head x 128 consecutive query positions of one sequence, 4 waves x 32
// rows (two 16-row MFMA tiles); keys advance in tiles of 64, double-buffered in LDS by LDS-DMA from the contiguous buffers:
//   K_nope image  [64 keys][256 B]  chunk c of key r at c ^ (r & 15)              (as csrc/paged_prefill_gqa.hip)
//   K_pe   image  [64 keys][128 B]  chunk c of key r at c ^ ((r >> 1) & 7)        (rows of 128 B: parity of r picks the
//                                                                                   bank half, the XOR spreads the rest)
//   V      image  [64 keys][256 B]  chunk c of key r at c ^ ((r & 7) << 1)         (transposed reads, ds_read_b64_tr_b16)
// Transposed formulation S^T = K Q^T, O^T += V^T P^T (v_mfma_f32_16x16x32): statistics lane-local, P feeds the second
// product from registers; lazy reference maximum.  fp32 scores / statistics / accumulation, probabilities rounded to the
// storage type.  Optional per-head sink logit joins the denominator only (attention.py:20-42).
//
// FLOPs: decompression 2 * T_kv * r * H * (nope + v) + attention sum_b 2 * H * (D_qk + D_v) * (q_b * kv_b - q_b (q_b - 1) / 2).
#include <math.h>

#include <type_traits>

#include "common.h"

namespace mojo {

typedef __attribute__((address_space(3))) char lds_m;

// Row of sequence b in the flat (un-paged) buffers when every sequence is cut at `per_seq` keys: sum over i < b of
// min(len_i, per_seq).  The host sizes the buffers from per_seq without reading a length; a sequence longer than the bound
// loses ITS OWN tail and nothing else (a positional cut at the capacity would keep an over-long early sequence whole and
// starve well-formed later ones).  Wave-uniform result; every lane of a wave calls it.  cu is relative (cu[0] = first sequence).
__device__ __forceinline__ int mla_clamped_start(const int32_t* cu, int b, int per_seq) {
  const int lane = threadIdx.x & 63;
  int v = 0;
  for (int i = lane; i < b; i += 64) v += min(cu[i + 1] - cu[i], per_seq);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return __builtin_amdgcn_readfirstlane(v);
}

// ---- 1. un-page --------------------------------------------------------------------------------------------------
// flat row cu_kv[b] + t  <-  cache[table[b, t / page], 0, t % page, :]   for t < kv_len_b.  One wave per token row.
struct UnpageArgs {
  const char* ckv; const char* kpe;
  char* ckv_out; char* kpe_out;
  const int32_t* cu_q; const int32_t* cu_kv; const int32_t* tables;
  int32_t* count_out;                                          // [1]: total keys of these sequences (the GEMM's row count)
  int64_t table_stride, ckv_blk, ckv_tok, kpe_blk, kpe_tok;     // bytes
  int ckv_row_bytes, kpe_row_bytes, page, max_pages, batch;
  int capacity_rows;                                           // rows the flat buffers hold: nothing is written at or past it
  int per_seq;                                                 // keys kept per sequence (the host's bound of one sequence's length)
};

__global__ __launch_bounds__(256) void mla_unpage_kernel(UnpageArgs a) {
  const int b = blockIdx.y;
  // (flat rows are relative to the first sequence handed in: a caller may pass a slice of the batch)
  const int32_t* cu = a.cu_kv ? a.cu_kv : a.cu_q;
  // The host sizes the flat buffers from what it knows without a sync (table width, a caller's hint): a sequence longer than
  // that bound is cut at the bound, here and — by the same rule — in the attention kernel, PER SEQUENCE as the paged GQA ops
  // cut theirs; the capacity stays the hard limit of what is written.
  const int start = mla_clamped_start(cu, b, a.per_seq);
  int len = min(cu[b + 1] - cu[b], a.per_seq);
  if (len > a.capacity_rows - start) len = a.capacity_rows - start;
  if (a.count_out && b == 0 && blockIdx.x == 0 && threadIdx.x < 64) {
    const int total = mla_clamped_start(cu, a.batch, a.per_seq);
    if (threadIdx.x == 0) a.count_out[0] = min(total, a.capacity_rows);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int32_t* table = a.tables + static_cast<int64_t>(b) * a.table_stride;
  for (int t = blockIdx.x * 4 + wave; t < len; t += gridDim.x * 4) {
    const int lp = t / a.page;
    int phys = lp < a.max_pages ? table[lp] : -1;
    if (phys < 0) phys = 0;                            // (the golden cannot run with a hole inside kv_len: :355-369 then fails)
    const int off = t - lp * a.page;
    const char* s1 = a.ckv + phys * a.ckv_blk + off * a.ckv_tok;
    char* d1 = a.ckv_out + static_cast<int64_t>(start + t) * a.ckv_row_bytes;
    for (int i = lane * 16; i < a.ckv_row_bytes; i += 64 * 16) *reinterpret_cast<u32x4*>(d1 + i) = *reinterpret_cast<const u32x4*>(s1 + i);
    const char* s2 = a.kpe + phys * a.kpe_blk + off * a.kpe_tok;
    char* d2 = a.kpe_out + static_cast<int64_t>(start + t) * a.kpe_row_bytes;
    for (int i = lane * 16; i < a.kpe_row_bytes; i += 64 * 16) *reinterpret_cast<u32x4*>(d2 + i) = *reinterpret_cast<const u32x4*>(s2 + i);
  }
}

// ---- 3. attention --------------------------------------------------------------------------------------------------
struct MlaPfArgs {
  const void* q;             // [T, H, nope + rope]
  const void* kv;            // [T_kv, H, nope + vd]   decompressed
  const void* kpe;           // [T_kv, rope]
  const float* sink;         // [H] or null
  void* out;                 // [T, H, vd]
  const int32_t* cu_q;
  const int32_t* cu_kv;      // may be null: kv_len = q_len
  int heads, batch, n_qb;    // heads = heads of the tensors (row strides)
  int head0, heads_here;     // this launch covers heads [head0, head0 + heads_here): the decompression of the next head group
                             // can run beside it on another stream (operators/mla.py)
  int capacity_rows;         // rows of `kv` / `kpe`: keys at or past it do not exist (mla_unpage wrote none)
  int per_seq;               // keys kept per sequence (mla_unpage's rule: rows of sequence b start at sum_{i<b} min(len_i, per_seq))
  int64_t total_tokens;      // rows of `out`; rows behind cu_q[batch] are zeroed when zero_tail is set
  float scale_log2;          // softmax_scale * log2(e); log2(e) alone when the scores are scaled before the exponent (below)
  float pre_scale;           // decode golden: scores = round(round(q k) * softmax_scale) in the storage type (attention.py:215)
  int round_scaled;          // 1 = apply pre_scale and round once more; 0 = the prefill golden (fp32 scaling, attention.py:425)
  int zero_tail;
  int n_slots;               // dispatch slots per unit: n_qb, made odd (see the kernel)
};

template <typename T> struct mpf_mfma;
template <> struct mpf_mfma<bf16_t> {
  typedef bf16x8 frag;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct mpf_mfma<f16_t> {
  typedef f16x8 frag;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

constexpr float MPF_LAZY_LOG2 = 8.f;
constexpr int MPF_KEYS = 64;
constexpr int MPF_QPB = 128;                          // query positions per 4-wave workgroup (8 waves: 256)
constexpr int MPF_KN_BYTES = MPF_KEYS * 256;          // K_nope image
constexpr int MPF_KP_BYTES = MPF_KEYS * 128;          // K_pe image
constexpr int MPF_V_BYTES = MPF_KEYS * 256;           // V image
constexpr int MPF_BUF_BYTES = MPF_KN_BYTES + MPF_KP_BYTES + MPF_V_BYTES;   // 40 KiB
constexpr int MPF_LDS = 2 * MPF_BUF_BYTES;            // 80 KiB: two workgroups per CU
constexpr int MPF_ZERO_TOKENS = 32;

// NW = waves per workgroup, 32 rows each (QPB = 32 NW query positions).  Waves skip the key tiles none of their rows sees and
// take the masked form only on the tiles their OWN rows need it (the causal band of a block is QPB / 64 tiles wide).
// Round 4: NW = 8 (256 rows, one workgroup per CU, half the K/V bytes through the LDS-DMA — every head has its own K/V, so a
// 40 KiB tile feeds only one workgroup's rows) was built on the hypothesis that the launch is bound by that stream (2.9 GB
// for 4 x 512 tokens against 2048 cached ones).  Parity-green and 10-20 % SLOWER (4 x 512: 203 -> 222 us, + 2048 cached: 930 ->
// 1120 us, profiles/r4_mla_prefill_waves_ab.json): the stream is served by L2 (the query blocks of a (sequence, head) are
// resident together on one XCD), and eight waves behind one barrier align their LDS reads as the phase-alternating GQA kernel
// did (DESIGN Appendix A 10a).  Only NW = 4 is instantiated.
template <typename T, int DKN /* nope / 32 */, int DKR /* rope / 32 */, int DVV /* vd / 32 */, int NW = 4>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void mla_prefill_kernel(MlaPfArgs a) {
  constexpr int QPB = 32 * NW, NT = 64 * NW;
  typedef typename mpf_mfma<T>::frag frag;
  constexpr int NOPE = DKN * 32, ROPE = DKR * 32, VD = DVV * 32;
  constexpr int QK = NOPE + ROPE, KVW = NOPE + VD;     // query row width, decompressed row width per head
  constexpr int DT = DVV * 2;                          // 16-wide d tiles of the output
  constexpr int DK = DKN + DKR;                        // 32-wide k steps of Q K^T
  extern __shared__ __attribute__((aligned(1024))) char smem_generic[];
  lds_m* smem = (lds_m*)smem_generic;

  // Block -> (sequence, head, query block).  The query blocks of one (sequence, head) read the SAME keys (block j the first
  // offset + 128 (j + 1) of them), and with 128 heads nothing else does: they must meet in one L2.  Consecutive block ids go
  // to consecutive XCDs (each with its own 4 MiB L2), ids 8 apart share an XCD in dispatch order; so a unit = (sequence, head)
  // is pinned to XCD (unit % 8) and its query blocks take consecutive slots there, longest first — they are resident together
  // and the later ones find the tiles of the first in L2.  (Query block as the slow coordinate, as in the GQA kernel, put 64
  // other units' streams — ~80 MB through a 4 MiB L2 — between two readers of the same keys: every block read HBM.)
  const int inner = a.heads_here * a.batch;
  const int units_per_xcd = (inner + 7) / 8;
  const int n_attn = 8 * units_per_xcd * a.n_slots;
  if (static_cast<int>(blockIdx.x) >= n_attn) {          // trailing workgroups: rows no sequence owns read as zeros
    const int64_t z = static_cast<int64_t>(blockIdx.x) - n_attn;
    const int64_t t0 = max(static_cast<int64_t>(a.cu_q[a.batch]), z * MPF_ZERO_TOKENS);
    const int64_t t1 = min(a.total_tokens, (z + 1) * MPF_ZERO_TOKENS);
    const int64_t row_elems = static_cast<int64_t>(a.heads) * VD;
    typedef typename vec_of<T, 8>::type V8;
    V8 zv;
#pragma unroll
    for (int e = 0; e < 8; ++e) zv[e] = static_cast<T>(0.f);
    for (int64_t i = t0 * row_elems + threadIdx.x * 8; i < t1 * row_elems; i += NT * 8)
      *reinterpret_cast<V8*>(static_cast<T*>(a.out) + i) = zv;
    return;
  }
  const int xcd = static_cast<int>(blockIdx.x) & 7, slot = static_cast<int>(blockIdx.x) >> 3;
  const int unit = (slot / a.n_slots) * 8 + xcd;
  if (unit >= inner) return;
  // Inside an XCD the blocks go to its four shader engines in strict rotation and in order, and a block waits for ITS
  // engine (scripts/probes/dispatch_rate.hip: work that repeats every 4 blocks of an XCD lands on one engine, 3.7x the
  // balanced time).  With every unit's blocks listed longest first and an even count per unit, an engine would get the same
  // position of every unit — engine 0 all the longest blocks.  The count of slots per unit is therefore odd (one empty
  // slot when n_qb is even), so consecutive units start on consecutive engines and the order inside a unit stays.
  if (slot % a.n_slots >= a.n_qb) return;
  const int qb = a.n_qb - 1 - slot % a.n_slots;
  const int head = a.head0 + unit % a.heads_here, b = unit / a.heads_here;
  const int q_start = a.cu_q[b];
  const int q_len = a.cu_q[b + 1] - q_start;
  const int kv_start = mla_clamped_start(a.cu_kv ? a.cu_kv : a.cu_q, b, a.per_seq);   // row in the (slice-relative) flat buffers
  int kv_len = min(a.cu_kv ? a.cu_kv[b + 1] - a.cu_kv[b] : q_len, a.per_seq);
  if (kv_len > a.capacity_rows - kv_start) kv_len = a.capacity_rows - kv_start;
  if (qb * QPB >= q_len) return;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = lane >> 4, l15 = lane & 15;
  typedef typename vec_of<T, 8>::type V8;

  if (kv_len <= 0) {                                     // a sequence without keys: its rows read as zeros (:396-397)
    V8 zv;
#pragma unroll
    for (int e = 0; e < 8; ++e) zv[e] = static_cast<T>(0.f);
    const int n_pos = min(q_len - qb * QPB, QPB);
    for (int i = threadIdx.x; i < n_pos * (VD / 8); i += NT) {
      const int c = i % (VD / 8), pos = qb * QPB + i / (VD / 8);
      *reinterpret_cast<V8*>(static_cast<T*>(a.out) + (static_cast<int64_t>(q_start + pos) * a.heads + head) * VD + c * 8) = zv;
    }
    return;
  }
  const int offset = kv_len - q_len;                     // query i sees keys 0 .. offset + i
  const int pos_hi = min(q_len, (qb + 1) * QPB) - 1;
  int kv_hi = min(kv_len, offset + pos_hi + 1);          // keys [0, kv_hi) are visible to some row of this block
  if (kv_hi < 1) kv_hi = 1;
  const int n_kb = (kv_hi + MPF_KEYS - 1) / MPF_KEYS;

  // ---- this wave's rows: two 16-row tiles -------------------------------------------------------------------------
  int row_pos[2];
  frag qf[2][DK];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    int pos = qb * QPB + wave * 32 + qt * 16 + l15;
    row_pos[qt] = pos;
    if (pos >= q_len) pos = q_len - 1;                   // clamp: computed, never stored
    const T* qp = static_cast<const T*>(a.q) + (static_cast<int64_t>(q_start + pos) * a.heads + head) * QK;
#pragma unroll
    for (int ks = 0; ks < DK; ++ks) qf[qt][ks] = *reinterpret_cast<const frag*>(qp + ks * 32 + grp * 8);
  }

  // ---- staging: wave w fills keys [KPW w, KPW w + KPW) of a tile, KPW = 64 / NW ------------------------------------------
  //   K_nope / V: 4 LDS-DMA instructions each (4 keys x 256 B; lane l: key l / 16, LDS chunk position l % 16)
  //   K_pe      : 2 instructions (8 keys x 128 B; lane l: key l / 8, position l % 8)
  const char* kv_base = reinterpret_cast<const char*>(static_cast<const T*>(a.kv) + (static_cast<int64_t>(kv_start) * a.heads + head) * KVW);
  const char* pe_base = reinterpret_cast<const char*>(static_cast<const T*>(a.kpe) + static_cast<int64_t>(kv_start) * ROPE);
  const int64_t kv_row_bytes = static_cast<int64_t>(a.heads) * KVW * sizeof(T);
  constexpr int KPW = MPF_KEYS / NW;
  auto stage = [&](int kb, int buf) {
    lds_m* base = smem + buf * MPF_BUF_BYTES;
#pragma unroll
    for (int i = 0; i < KPW / 4; ++i) {
      const int kl = wave * KPW + i * 4 + (lane >> 4);
      int key = kb * MPF_KEYS + kl;
      if (key >= kv_hi) key = kv_hi - 1;                 // rows past the end: re-read a valid key, masked later
      const int cp = lane & 15;
      int ck = cp ^ (kl & 15);
      int cv = cp ^ ((kl & 7) << 1);
      if (ck >= NOPE / 8) ck = NOPE / 8 - 1;
      if (cv >= VD / 8) cv = VD / 8 - 1;
      const char* row = kv_base + static_cast<int64_t>(key) * kv_row_bytes;
      lds_m* dk = base + (wave * KPW + i * 4) * 256;
      lds_m* dv = base + MPF_KN_BYTES + MPF_KP_BYTES + (wave * KPW + i * 4) * 256;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(row + ck * 16),
                                       (__attribute__((address_space(3))) void*)dk, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(row + NOPE * sizeof(T) + cv * 16),
                                       (__attribute__((address_space(3))) void*)dv, 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < KPW / 8; ++i) {
      const int kl = wave * KPW + i * 8 + (lane >> 3);
      int key = kb * MPF_KEYS + kl;
      if (key >= kv_hi) key = kv_hi - 1;
      int cr = (lane & 7) ^ ((kl >> 1) & 7);
      if (cr >= ROPE / 8) cr = ROPE / 8 - 1;
      lds_m* dp = base + MPF_KN_BYTES + (wave * KPW + i * 8) * 128;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pe_base + static_cast<int64_t>(key) * (ROPE * sizeof(T)) + cr * 16),
                                       (__attribute__((address_space(3))) void*)dp, 16, 0, 0);
    }
  };

  // ---- state --------------------------------------------------------------------------------------------------------
  f32x4 o[2][DT];
  float m[2], lsum[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    m[qt] = -INFINITY;
    lsum[qt] = 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float lazy_raw = MPF_LAZY_LOG2 / a.scale_log2;
  const unsigned smem_u32 = static_cast<unsigned>(reinterpret_cast<size_t>(smem));
  const int tq = l15 >> 2, tp = l15 & 3;
  // per wave: key blocks every row of the wave sees completely, and one past the last block any of its rows sees
  // (rows past the sequence's end are clamped to its last position: computed, never stored)
  const int wpos_lo = min(qb * QPB + wave * 32, q_len - 1), wpos_hi = min(qb * QPB + wave * 32 + 31, q_len - 1);
  const int n_full = min(kv_len, offset + wpos_lo + 1) / MPF_KEYS;
  const int n_mine = min(n_kb, (max(min(kv_len, offset + wpos_hi + 1), 1) + MPF_KEYS - 1) / MPF_KEYS);

  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  auto key_block = [&](auto masked_tag, int kb) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    const int buf = kb & 1;
    const lds_m* kn = smem + buf * MPF_BUF_BYTES;
    const lds_m* kp = kn + MPF_KN_BYTES;
    const unsigned vt = smem_u32 + buf * MPF_BUF_BYTES + MPF_KN_BYTES + MPF_KP_BYTES;

    // ---- S^T = K Q^T : 4 key tiles x 2 q tiles, DK k-steps (nope from the K_nope image, rope from the K_pe image) ----
    // fragments of two key tiles are in flight at a time (DK = 6 would need 96 registers for all four)
    f32x4 s[2][4];
    frag kf[2][DK];
    auto read_k = [&](frag (&dst)[DK], int t) {
      const int key_row = t * 16 + l15;
#pragma unroll
      for (int ks = 0; ks < DKN; ++ks)
        dst[ks] = *reinterpret_cast<const __attribute__((address_space(3))) frag*>(kn + key_row * 256 + (((ks * 4 + grp) ^ (key_row & 15)) * 16));
#pragma unroll
      for (int ks = 0; ks < DKR; ++ks)
        dst[DKN + ks] = *reinterpret_cast<const __attribute__((address_space(3))) frag*>(kp + key_row * 128 + (((ks * 4 + grp) ^ ((key_row >> 1) & 7)) * 16));
    };
    read_k(kf[0], 0);
    read_k(kf[1], 1);
    if (kb + 1 < n_kb) stage(kb + 1, buf ^ 1);           // behind the reads: its issue time covers their latency
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      s[0][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      s[1][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < DK; ++ks) {
        s[0][t] = mpf_mfma<T>::run(kf[t & 1][ks], qf[0][ks], s[0][t]);
        s[1][t] = mpf_mfma<T>::run(kf[t & 1][ks], qf[1][ks], s[1][t]);
      }
      if (t + 2 < 4) read_k(kf[t & 1], t + 2);
    }

    // ---- V^T fragments: transposed reads, 4 per d tile, issued in two batches of DT / 2 d tiles -------------------------
    constexpr int HB = DT / 2 * 4;                       // reads per batch (8 or 16)
    auto issue_v = [&](s16x4 (&dst)[16], int dt0) {
      unsigned ad[4];
#pragma unroll
      for (int i = 0; i < DT / 2; ++i) {
        const int dt = dt0 + i;
        const int row = 4 * grp + tq;
        ad[i] = vt + row * 256 + (((2 * dt + (tp >> 1)) ^ ((row & 7) << 1)) * 16) + (tp & 1) * 8;
      }
      if constexpr (DT / 2 == 4) {
        asm volatile(
            "ds_read_b64_tr_b16 %0, %16\n\tds_read_b64_tr_b16 %1, %16 offset:4096\n\tds_read_b64_tr_b16 %2, %16 offset:8192\n\tds_read_b64_tr_b16 %3, %16 offset:12288\n\t"
            "ds_read_b64_tr_b16 %4, %17\n\tds_read_b64_tr_b16 %5, %17 offset:4096\n\tds_read_b64_tr_b16 %6, %17 offset:8192\n\tds_read_b64_tr_b16 %7, %17 offset:12288\n\t"
            "ds_read_b64_tr_b16 %8, %18\n\tds_read_b64_tr_b16 %9, %18 offset:4096\n\tds_read_b64_tr_b16 %10, %18 offset:8192\n\tds_read_b64_tr_b16 %11, %18 offset:12288\n\t"
            "ds_read_b64_tr_b16 %12, %19\n\tds_read_b64_tr_b16 %13, %19 offset:4096\n\tds_read_b64_tr_b16 %14, %19 offset:8192\n\tds_read_b64_tr_b16 %15, %19 offset:12288"
            : "=&v"(dst[0]), "=&v"(dst[1]), "=&v"(dst[2]), "=&v"(dst[3]), "=&v"(dst[4]), "=&v"(dst[5]), "=&v"(dst[6]), "=&v"(dst[7]),
              "=&v"(dst[8]), "=&v"(dst[9]), "=&v"(dst[10]), "=&v"(dst[11]), "=&v"(dst[12]), "=&v"(dst[13]), "=&v"(dst[14]), "=&v"(dst[15])
            : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3])
            : "memory");
      } else {
        asm volatile(
            "ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:4096\n\tds_read_b64_tr_b16 %2, %8 offset:8192\n\tds_read_b64_tr_b16 %3, %8 offset:12288\n\t"
            "ds_read_b64_tr_b16 %4, %9\n\tds_read_b64_tr_b16 %5, %9 offset:4096\n\tds_read_b64_tr_b16 %6, %9 offset:8192\n\tds_read_b64_tr_b16 %7, %9 offset:12288"
            : "=&v"(dst[0]), "=&v"(dst[1]), "=&v"(dst[2]), "=&v"(dst[3]), "=&v"(dst[4]), "=&v"(dst[5]), "=&v"(dst[6]), "=&v"(dst[7])
            : "v"(ad[0]), "v"(ad[1])
            : "memory");
      }
    };
    auto retire_v = [&](s16x4 (&dst)[16]) {
      if constexpr (HB == 16) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(dst[0]), "+v"(dst[1]), "+v"(dst[2]), "+v"(dst[3]), "+v"(dst[4]), "+v"(dst[5]), "+v"(dst[6]), "+v"(dst[7]),
                       "+v"(dst[8]), "+v"(dst[9]), "+v"(dst[10]), "+v"(dst[11]), "+v"(dst[12]), "+v"(dst[13]), "+v"(dst[14]), "+v"(dst[15])
                     : : "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(dst[0]), "+v"(dst[1]), "+v"(dst[2]), "+v"(dst[3]), "+v"(dst[4]), "+v"(dst[5]), "+v"(dst[6]), "+v"(dst[7])
                     : : "memory");
      }
    };
    // lane holds, for query column l15 of each q tile, keys  kb*64 + 16t + 4*grp + r
    const int key0 = kb * MPF_KEYS + 4 * grp;
    frag pf[2][2];
    auto softmax_tile = [&](int qt) {
      f32x4 (&sc)[4] = s[qt];
      // the golden forms the scores with an einsum of storage-type tensors, i.e. ROUNDED to the storage type, before it
      // upcasts and scales them (attention.py:425): the same rounding point here keeps the two within an output ulp of each
      // other even where the scores are large (the reference's test draws kv_b_proj from randn: |score| ~ 50)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[t][r] = static_cast<float>(static_cast<T>(sc[t][r]));
      if (a.round_scaled) {                                // wave-uniform: the decode golden multiplies in the storage type
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) sc[t][r] = static_cast<float>(static_cast<T>(sc[t][r] * a.pre_scale));
      }
      if constexpr (MASKED) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = key0 + 16 * t + r;
            if (key > offset + row_pos[qt] || key >= kv_len) sc[t][r] = -INFINITY;
          }
      }
      float mx = fmaxf(fmaxf(sc[0][0], sc[0][1]), fmaxf(sc[0][2], sc[0][3]));
#pragma unroll
      for (int t = 1; t < 4; ++t) mx = fmaxf(mx, fmaxf(fmaxf(sc[t][0], sc[t][1]), fmaxf(sc[t][2], sc[t][3])));
      if (__any(mx > m[qt] + lazy_raw)) {
        mx = xor_max_16_32(mx);
        mx = fmaxf(mx, m[qt]);
        const float ms_new = (mx == -INFINITY ? 0.f : mx) * a.scale_log2;
        const float alpha = fast_exp2(m[qt] * a.scale_log2 - ms_new);
        m[qt] = mx;
        lsum[qt] *= alpha;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[qt][dt] *= alpha;
      }
      const float ms = (m[qt] == -INFINITY ? 0.f : m[qt]) * a.scale_log2;
      float ps = 0.f;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        frag f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p0 = fast_exp2(fmaf(sc[2 * kk][r], a.scale_log2, -ms));
          const float p1 = fast_exp2(fmaf(sc[2 * kk + 1][r], a.scale_log2, -ms));
          ps += p0 + p1;
          f[r] = static_cast<T>(p0);
          f[4 + r] = static_cast<T>(p1);
        }
        pf[qt][kk] = f;
      }
      lsum[qt] += ps;
    };
    s16x4 vb0[16], vb1[16];
    issue_v(vb0, 0);
    softmax_tile(0);
    softmax_tile(1);
    auto pv_batch = [&](const s16x4 (&src)[16], int dt0) {
#pragma unroll
      for (int i = 0; i < DT / 2; ++i)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const s16x4 lo = src[i * 4 + kk * 2], hi = src[i * 4 + kk * 2 + 1];
          const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          const frag vf = __builtin_bit_cast(frag, both);
          o[0][dt0 + i] = mpf_mfma<T>::run(vf, pf[0][kk], o[0][dt0 + i]);
          o[1][dt0 + i] = mpf_mfma<T>::run(vf, pf[1][kk], o[1][dt0 + i]);
        }
    };
    retire_v(vb0);
    issue_v(vb1, DT / 2);
    pv_batch(vb0, 0);
    retire_v(vb1);
    pv_batch(vb1, DT / 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // next tile landed
    __builtin_amdgcn_s_barrier();                         // ... and everyone is done reading this one
  };
  int kb_i = 0;
  for (; kb_i < n_full; ++kb_i) key_block(std::false_type{}, kb_i);
  for (; kb_i < n_mine; ++kb_i) key_block(std::true_type{}, kb_i);
  for (; kb_i < n_kb; ++kb_i) {                           // tiles only later waves need: keep staging and the barrier cadence
    if (kb_i + 1 < n_kb) stage(kb_i + 1, (kb_i & 1) ^ 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  // ---- finish: row sums over the 4 lane groups (+ the sink's share of the denominator), normalise, store whole rows ----
  constexpr int OROW = 272;                       // (68 dwords: the 16 rows of a write land 4 banks apart; 288 left rows l and l + 8 on one bank pair)
  lds_m* stage_o = smem + wave * (32 * OROW);
  typedef typename vec_of<T, 4>::type V4;
  const float sink_l2 = a.sink ? a.sink[head] * 1.4426950408889634f : 0.f;
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    float den = xor_sum_16_32(lsum[qt]);
    const float mref = xor_max_16_32(m[qt]);              // (the lane groups share one reference: m is wave-row uniform)
    if (a.sink) den += fast_exp2(sink_l2 - (mref == -INFINITY ? 0.f : mref) * a.scale_log2);
    const float inv = den > 0.f ? 1.0f / den : 0.f;       // a row that sees no key: zeros (nan_to_num, :27)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      V4 ov;
#pragma unroll
      for (int r = 0; r < 4; ++r) ov[r] = static_cast<T>(o[qt][dt][r] * inv);
      *reinterpret_cast<__attribute__((address_space(3))) V4*>(stage_o + (qt * 16 + l15) * OROW + (dt * 16 + grp * 4) * 2) = ov;
    }
  }
  {
    constexpr int CPR = DT * 2;                          // 16-byte chunks per output row (vd / 8)
    constexpr int RPI = 64 / CPR;                        // rows per store instruction
    const int sub = lane / CPR, ch = lane % CPR;
#pragma unroll
    for (int i = 0; i < (32 + RPI - 1) / RPI; ++i) {
      const int row = i * RPI + sub;
      if (sub >= RPI || row >= 32) continue;
      const int pos = qb * QPB + wave * 32 + row;
      if (pos >= q_len) continue;
      const V8 v = *reinterpret_cast<const __attribute__((address_space(3))) V8*>(stage_o + row * OROW + ch * 16);
      *reinterpret_cast<V8*>(static_cast<T*>(a.out) + (static_cast<int64_t>(q_start + pos) * a.heads + head) * VD + ch * 8) = v;
    }
  }
}

template <typename T, int DKN, int DKR, int DVV, int NW>
static int launch_mla_pf(const MlaPfArgs& a, hipStream_t s) {
  auto* fn = mla_prefill_kernel<T, DKN, DKR, DVV, NW>;
  static std::atomic<uint64_t> attr_set{0};
  if (first_call_on_device(attr_set)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, MPF_LDS);
  const int64_t n_zero = a.zero_tail ? ceil_div(a.total_tokens, static_cast<int64_t>(MPF_ZERO_TOKENS)) : 0;
  const int64_t blocks = static_cast<int64_t>(a.n_slots) * 8 * ceil_div(static_cast<int64_t>(a.heads_here) * a.batch, 8) + n_zero;
  MOJO_REQUIRE(blocks < (int64_t{1} << 31), MOJO_EUNSUPPORTED, "mla_prefill: grid limit");
  hipLaunchKernelGGL(fn, dim3(static_cast<unsigned>(blocks)), dim3(NW * 64), MPF_LDS, s, a);
  MOJO_CHECK_LAUNCH("mla_prefill");
  note_launch("mla_prefill_attn");
  return MOJO_OK;
}

}  // namespace mojo

using namespace mojo;

// 1 when (nope, rope, v) has an instantiation of the non-absorbed attention kernel
extern "C" int mojo_hip_mla_prefill_supported(int64_t nope, int64_t rope, int64_t v_dim, int dtype) {
  if (dtype != MOJO_BF16 && dtype != MOJO_F16) return 0;
  return (nope == 128 && rope == 64 && v_dim == 128) || (nope == 64 && rope == 32 && v_dim == 64) ||
         (nope == 96 && rope == 32 && v_dim == 128);
}

extern "C" int mojo_hip_mla_unpage(const void* compressed_kv_cache, const void* k_pe_cache, void* ckv_out, void* kpe_out,
                                   const int32_t* cu_q_lens, const int32_t* cu_total_seq_lens, const int32_t* block_tables,
                                   int64_t block_table_stride, int64_t max_blocks_per_seq, int64_t batch,
                                   int64_t kv_lora_rank, int64_t rope_dim, int64_t block_size, int64_t elt_bytes,
                                   int64_t ckv_block_stride, int64_t ckv_token_stride, int64_t kpe_block_stride,
                                   int64_t kpe_token_stride, int64_t max_tokens_per_seq, int64_t capacity_rows,
                                   int32_t* total_keys_out, mojo_stream_t stream) {
  if (batch == 0 || max_tokens_per_seq <= 0) return MOJO_OK;
  MOJO_REQUIRE(capacity_rows > 0 && capacity_rows < (int64_t{1} << 31), MOJO_EINVAL, "mla_unpage: capacity_rows %lld",
               (long long)capacity_rows);
  MOJO_REQUIRE(compressed_kv_cache && k_pe_cache && ckv_out && kpe_out && cu_q_lens && block_tables, MOJO_EINVAL,
               "mla_unpage: null pointer");
  MOJO_REQUIRE((kv_lora_rank * elt_bytes) % 16 == 0 && (rope_dim * elt_bytes) % 16 == 0 && aligned_to(compressed_kv_cache, 16) &&
                   aligned_to(k_pe_cache, 16) && aligned_to(ckv_out, 16) && aligned_to(kpe_out, 16) &&
                   (ckv_token_stride * elt_bytes) % 16 == 0 && (kpe_token_stride * elt_bytes) % 16 == 0 &&
                   (ckv_block_stride * elt_bytes) % 16 == 0 && (kpe_block_stride * elt_bytes) % 16 == 0,
               MOJO_EUNSUPPORTED, "mla_unpage: rows must be whole 16-byte vectors");
  MOJO_REQUIRE(batch <= 65535, MOJO_EUNSUPPORTED, "mla_unpage: batch %lld exceeds the grid limit", (long long)batch);
  UnpageArgs a;
  a.ckv = static_cast<const char*>(compressed_kv_cache); a.kpe = static_cast<const char*>(k_pe_cache);
  a.ckv_out = static_cast<char*>(ckv_out); a.kpe_out = static_cast<char*>(kpe_out);
  a.cu_q = cu_q_lens; a.cu_kv = cu_total_seq_lens; a.tables = block_tables; a.table_stride = block_table_stride;
  a.count_out = total_keys_out;
  a.ckv_blk = ckv_block_stride * elt_bytes; a.ckv_tok = ckv_token_stride * elt_bytes;
  a.kpe_blk = kpe_block_stride * elt_bytes; a.kpe_tok = kpe_token_stride * elt_bytes;
  a.ckv_row_bytes = static_cast<int>(kv_lora_rank * elt_bytes); a.kpe_row_bytes = static_cast<int>(rope_dim * elt_bytes);
  a.page = static_cast<int>(block_size); a.max_pages = static_cast<int>(max_blocks_per_seq); a.batch = static_cast<int>(batch);
  a.capacity_rows = static_cast<int>(capacity_rows);
  a.per_seq = static_cast<int>(max_tokens_per_seq < (int64_t{1} << 30) ? max_tokens_per_seq : (int64_t{1} << 30));
  int64_t gx = ceil_div(max_tokens_per_seq, 4);
  if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(mla_unpage_kernel, dim3(static_cast<unsigned>(gx), static_cast<unsigned>(batch)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a);
  MOJO_CHECK_LAUNCH("mla_unpage");
  return MOJO_OK;
}

extern "C" int mojo_hip_mla_prefill_attn(const void* query, const void* kv_decompressed, const void* k_pe_flat,
                                         const float* attn_sink, void* out, const int32_t* cu_q_lens,
                                         const int32_t* cu_total_seq_lens, int64_t total_tokens, int64_t batch,
                                         int64_t heads, int64_t head_begin, int64_t head_count, int64_t nope, int64_t rope,
                                         int64_t v_dim, int64_t max_q_len,
                                         int64_t max_tokens_per_seq, int64_t capacity_rows, float softmax_scale, int round_scaled_scores,
                                         int zero_padding_rows, int dtype, mojo_stream_t stream) {
  if (total_tokens == 0) return MOJO_OK;
  MOJO_REQUIRE(capacity_rows > 0 && capacity_rows < (int64_t{1} << 31), MOJO_EINVAL, "mla_prefill_attn: capacity_rows %lld",
               (long long)capacity_rows);
  MOJO_REQUIRE(query && kv_decompressed && k_pe_flat && out && cu_q_lens, MOJO_EINVAL, "mla_prefill_attn: null pointer");
  MOJO_REQUIRE(mojo_hip_mla_prefill_supported(nope, rope, v_dim, dtype), MOJO_EUNSUPPORTED,
               "mla_prefill_attn: (nope, rope, v) = (%lld, %lld, %lld) has no instantiation", (long long)nope, (long long)rope, (long long)v_dim);
  MOJO_REQUIRE(head_begin >= 0 && head_count > 0 && head_begin + head_count <= heads, MOJO_EINVAL,
               "mla_prefill_attn: head range [%lld, +%lld) of %lld", (long long)head_begin, (long long)head_count, (long long)heads);
  MOJO_REQUIRE(heads > 0 && batch >= 0 && aligned_to(query, 16) && aligned_to(kv_decompressed, 16) && aligned_to(k_pe_flat, 16) &&
                   aligned_to(out, 16), MOJO_EINVAL, "mla_prefill_attn: bad arguments");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (batch == 0) {
    hipError_t e = hipMemsetAsync(out, 0, static_cast<size_t>(total_tokens * heads * v_dim * 2), s);
    MOJO_REQUIRE(e == hipSuccess, MOJO_ELAUNCH, "mla_prefill_attn: memset failed");
    return MOJO_OK;
  }
  MlaPfArgs a;
  a.q = query; a.kv = kv_decompressed; a.kpe = k_pe_flat; a.sink = attn_sink; a.out = out;
  a.cu_q = cu_q_lens; a.cu_kv = cu_total_seq_lens;
  a.heads = static_cast<int>(heads); a.batch = static_cast<int>(batch);
  a.head0 = static_cast<int>(head_begin); a.heads_here = static_cast<int>(head_count);
  a.capacity_rows = static_cast<int>(capacity_rows);
  a.per_seq = static_cast<int>(max_tokens_per_seq > 0 && max_tokens_per_seq < (int64_t{1} << 30) ? max_tokens_per_seq : (int64_t{1} << 30));
  const int64_t mq = (max_q_len > 0 && max_q_len < total_tokens) ? max_q_len : total_tokens;
  constexpr int nw = 4;                                // (8 waves = 256 rows per workgroup: measured 10-20 % slower, see the kernel's header)
  a.n_qb = static_cast<int>(ceil_div(mq, static_cast<int64_t>(32 * nw)));
  a.n_slots = a.n_qb | 1;
#ifdef MOJO_HIP_BUILD_EXPERIMENTS           // placement A/B of the odd slot count
  if (MOJO_SWITCH("MOJO_HIP_MLA_PREFILL_ODD_SLOTS", 1) == 0) a.n_slots = a.n_qb;
#endif
  a.total_tokens = total_tokens;
  a.round_scaled = round_scaled_scores ? 1 : 0;
  a.pre_scale = softmax_scale;
  a.scale_log2 = (round_scaled_scores ? 1.0f : softmax_scale) * 1.4426950408889634f;
  a.zero_tail = zero_padding_rows ? 1 : 0;
#define MPF_LAUNCH(T, NW_)                                                             \
  (nope == 128 ? launch_mla_pf<T, 4, 2, 4, NW_>(a, s) : nope == 64 ? launch_mla_pf<T, 2, 1, 2, NW_>(a, s) : launch_mla_pf<T, 3, 1, 4, NW_>(a, s))
  return dtype == MOJO_BF16 ? MPF_LAUNCH(bf16_t, nw) : MPF_LAUNCH(f16_t, nw);
#undef MPF_LAUNCH
}
