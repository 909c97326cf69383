// MojoPagedDecodeMLA / MojoPagedPrefillMLA — attention over the compressed (latent) KV cache, gfx950.
//
// Weight-absorbed formulation.  With W_kn[h] (nope x r) and W_v[h] (v x r) the two
// halves of kv_b_proj for head h:
//     score[h][s] = (q_nope[h] W_kn[h]) . c_kv[s] + q_rope[h] . k_pe[s]     (q_lat = [q_nope W_kn | q_rope])
//     out[h]      = W_v[h] ( sum_s p[h][s] c_kv[s] )                         (o_lat = sum_s p c_kv)
// so the cache is read ONCE in its compressed form (r + rope values per token) instead of being
// decompressed to H x (nope + v) values per token as the golden does.  The two projections run on the
// grouped GEMM (one group per head); this file is the latent attention in between.
//
// One workgroup = one "row tile" = all H (<= 128) heads of ONE query token (decode: the sequence's single
// token; prefill: token t, which sees keys 0 .. offset + t), optionally one split of the key range.
// Wave w owns heads [16w, 16w+16).  Keys advance in tiles of 64, double-buffered in LDS, filled by
// LDS-DMA from the c_kv and k_pe pages; an LDS row is [c_kv (r) | k_pe (rope)] with 16-byte chunk c of
// key s stored at c ^ (s & 7) and a row stride = 128 (mod 256) bytes, conflict-free for both the row
// reads (QK^T) and the transposed reads (PV).
//   S^T[key][h] = K_lat Q_lat^T       (A = K_lat fragment via ds_read_b128,      B = Q_lat fragment in VGPRs)
//   O^T[d][h]  += C_kv^T P^T          (A = c_kv^T fragment via ds_read_b64_tr_b16, B = P^T from the S^T accumulators)
//
// Algorithmic bytes (decode): sum_b len_b * (r + rope) * 2 + q/o;  FLOPs: 2 * H * len * (2r + rope).
#include <math.h>

#include "common.h"

namespace mojo {

typedef __attribute__((address_space(3))) char lds_m;

struct MlaArgs {
  const void* q_lat;        // [Tq, H, r (+ rope)], rows q_stride elements apart
  const void* q_rope;       // rope part of the query, rows q_rope_stride apart (may point into q_lat)
  int64_t q_stride, q_rope_stride;
  const void* ckv;          // [N, 1, page, r]
  const void* kpe;          // [N, 1, page, rope]
  void* o_lat;              // [Tq, H, r]  storage dtype (final) ...
  float* part_o;            // ... or [Tq, splits, H, r] fp32 un-normalised partials
  float* part_ml;           // [Tq, splits, H, 2]
  const int32_t* seq_lens;  // decode: [B] total lengths;   prefill: nullptr
  const int32_t* cu_q;      // prefill: [B+1]
  const int32_t* cu_kv;     // prefill: [B+1] or nullptr
  const int32_t* tables;
  const float* sink;        // [H] or nullptr (natural-log units)
  int64_t table_stride, ckv_blk, ckv_tok, kpe_blk, kpe_tok;
  int heads, page, page_shift, max_pages, batch, n_tiles, n_splits, split_keys;
  float scale_log2;
  int ps_debug;             // mla512_ps ablation bits (MOJO_HIP_MLA_PS_DEBUG; timing only: 1 no DMA after the first slots, 2 no QK^T, 4 no softmax, 8 no PV)
  int ps_issuers;           // mla512_ps: waves that issue the LDS-DMA: 2 (the loaders; default) or 6 (loaders + consumers; MOJO_HIP_MLA_PS_ISSUERS=6: measured equal)
  int ps_prefetch;          // mla512_ps: L2 prefetch ahead of the LDS-DMA (MOJO_HIP_MLA_PS_PREFETCH=1: by the loaders, 2: by a consumer wave per group; default off)
};

template <typename T> struct mla_mfma;
template <> struct mla_mfma<bf16_t> {
  typedef bf16x8 frag;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct mla_mfma<f16_t> {
  typedef f16x8 frag;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

constexpr int MLA_KEYS = 64;
constexpr int MLA512_DEFAULT_KERNEL = 3;       // 0 oct, 1 pair, 2 ping-pong, 3 specialised waves (dispatch_mla)

template <int R, int ROPE> struct mla_geom {
  static constexpr int CH = (R + ROPE) / 8;                               // 16-byte chunks per latent row
  static constexpr int CH8 = (CH + 7) / 8 * 8;
  static constexpr int CHS = ((CH8 / 8) % 2 == 0) ? CH8 + 8 : CH8;       // row stride: odd multiple of 128 B
  static constexpr int ROW_BYTES = CHS * 16;
  static constexpr int TILE_BYTES = MLA_KEYS * ROW_BYTES;
  static constexpr int TABLE_ENTRIES = 4096;                              // block-table slice cached in LDS
  static constexpr int LDS_BYTES = 2 * TILE_BYTES + TABLE_ENTRIES * 4;
  static constexpr int NK = (R + ROPE) / 32;                              // k-steps of QK^T
  static constexpr int ND = R / 16;                                        // 16-wide d tiles of O
};

// NQ = 16-head column tiles per wave: 1 -> 8 waves x 16 heads (<= 256 VGPRs each);  2 -> 4 waves x 32 heads with
// the whole 512-register file per wave (needed for r = 512: 128 accumulator + 72 query registers per tile).
template <typename T, int R, int ROPE, int NQ>
__global__ __launch_bounds__(512 / NQ, NQ == 1 ? 2 : 1) void mla_latent_kernel(MlaArgs a) {
  constexpr int NTHREADS = 512 / NQ;
  typedef typename mla_mfma<T>::frag frag;
  typedef mla_geom<R, ROPE> GE;
  extern __shared__ __attribute__((aligned(1024))) char smem_generic[];
  lds_m* smem = (lds_m*)smem_generic;

  const int tile = blockIdx.x, split = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = lane >> 4, l15 = lane & 15;

  // ---- which sequence / how many visible keys -----------------------------------------------------------
  int b, n_vis;
  if (a.cu_q == nullptr) {                                  // decode: tile == sequence
    b = tile;
    n_vis = a.seq_lens[b];
  } else {                                                  // prefill: tile == query token
    if (tile < a.cu_q[0] || tile >= a.cu_q[a.batch]) return;   // padding token: stays zero
    int lo = 0, hi = a.batch;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (a.cu_q[mid] <= tile) lo = mid; else hi = mid;
    }
    b = lo;
    const int q_len = a.cu_q[b + 1] - a.cu_q[b];
    const int kv_len = a.cu_kv ? a.cu_kv[b + 1] - a.cu_kv[b] : q_len;
    n_vis = min(kv_len, kv_len - q_len + (tile - a.cu_q[b]) + 1);
  }
  const int32_t* table = a.tables + static_cast<int64_t>(b) * a.table_stride;
  if (n_vis > 0) {                                          // golden: pages behind the first negative id are dropped
    int p1 = (n_vis + a.page - 1) / a.page;
    int fn = -1;
    if (p1 > a.max_pages) { fn = a.max_pages; p1 = a.max_pages; }
    for (int base = 0; base < p1; base += 64) {
      const int idx = base + lane;
      const int v = idx < p1 ? table[idx] : 0;
      const unsigned long long neg = __ballot(v < 0);
      if (neg) { fn = base + __builtin_ctzll(neg); break; }
    }
    if (fn >= 0) n_vis = min(n_vis, fn * a.page);
  }
  const int k_begin = split * a.split_keys;
  const int k_end = min(n_vis, k_begin + a.split_keys);
  // A window of TABLE_ENTRIES page ids lives in LDS (refilled when the key loop walks past it): no dependent
  // global load — and no FLAT load, which hipcc guards with vmcnt(0)/lgkmcnt(0) — sits in front of the LDS-DMA.
  int* s_table = reinterpret_cast<int*>(smem_generic + 2 * GE::TILE_BYTES);
  int win_base = 0;
  auto fill_window = [&](int p0) {
    for (int i = threadIdx.x; i < GE::TABLE_ENTRIES; i += NTHREADS) s_table[i] = (p0 + i < a.max_pages) ? table[p0 + i] : -1;
    win_base = p0;
    __syncthreads();
  };
  fill_window(a.page_shift >= 0 ? (k_begin >> a.page_shift) : k_begin / a.page);
  const bool active = wave * 16 * NQ < a.heads;            // waves beyond the head count only help staging
  int head[NQ];
#pragma unroll
  for (int c = 0; c < NQ; ++c) head[c] = min((wave * NQ + c) * 16 + l15, a.heads - 1);
  const int n_kt = k_end > k_begin ? (k_end - k_begin + MLA_KEYS - 1) / MLA_KEYS : 0;

  // ---- Q_lat fragments of this wave's 16 heads -------------------------------------------------------------
  frag qf[NQ][GE::NK];
#pragma unroll
  for (int c = 0; c < NQ; ++c) {
    const int64_t qrow = static_cast<int64_t>(tile) * a.heads + head[c];
    const T* qp = static_cast<const T*>(a.q_lat) + qrow * a.q_stride;
    const T* qr = static_cast<const T*>(a.q_rope) + qrow * a.q_rope_stride;
#pragma unroll
    for (int ks = 0; ks < GE::NK; ++ks) {
      const int e = ks * 32 + grp * 8;
      qf[c][ks] = *reinterpret_cast<const frag*>(e < R ? qp + e : qr + (e - R));
    }
  }

  // ---- staging: the tile is GE::CHS * 64 chunks, laid out linearly; 512 lanes take 512 chunks per round ------
  auto stage = [&](int kt, int buf) {
    {  // wave-uniform for the whole workgroup
      const int k_first = k_begin + kt * MLA_KEYS;
      const int k_last = min(k_first + MLA_KEYS - 1, k_end - 1);
      const int p_last = a.page_shift >= 0 ? (k_last >> a.page_shift) : k_last / a.page;
      if (p_last >= win_base + GE::TABLE_ENTRIES) {
        __syncthreads();
        fill_window(a.page_shift >= 0 ? (k_first >> a.page_shift) : k_first / a.page);
      }
    }
    constexpr int TOTAL = MLA_KEYS * GE::CHS;
    constexpr int ROUNDS = (TOTAL + NTHREADS - 1) / NTHREADS;
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i) {
      const int base_chunk = i * NTHREADS + wave * 64;     // wave-uniform
      if (base_chunk >= TOTAL) break;                      // TOTAL is a multiple of 64: whole waves drop out
      const int L = base_chunk + lane;
      const int kl = L / GE::CHS, cp = L - kl * GE::CHS;
      int cs = (cp & ~7) | ((cp & 7) ^ (kl & 7));
      if (cs >= GE::CH) cs = GE::CH - 1;                   // pad chunks: never read
      int key = k_begin + kt * MLA_KEYS + kl;
      if (key >= k_end) key = k_end - 1;
      const int lp = a.page_shift >= 0 ? (key >> a.page_shift) : key / a.page;
      int phys = s_table[lp - win_base];
      if (phys < 0) phys = 0;
      const int slot = key - lp * a.page;
      const T* src = cs < R / 8
                         ? static_cast<const T*>(a.ckv) + static_cast<int64_t>(phys) * a.ckv_blk + static_cast<int64_t>(slot) * a.ckv_tok + cs * 8
                         : static_cast<const T*>(a.kpe) + static_cast<int64_t>(phys) * a.kpe_blk + static_cast<int64_t>(slot) * a.kpe_tok + (cs - R / 8) * 8;
      lds_m* dst = smem + buf * GE::TILE_BYTES + base_chunk * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };

  f32x4 o[NQ][GE::ND];
  float m[NQ], lsum[NQ];
#pragma unroll
  for (int c = 0; c < NQ; ++c) {
    m[c] = -INFINITY;
    lsum[c] = 0.f;
#pragma unroll
    for (int dt = 0; dt < GE::ND; ++dt) o[c][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const unsigned smem_u32 = static_cast<unsigned>(reinterpret_cast<size_t>(smem));
  const int tq = l15 >> 2, tp = l15 & 3;
  int kf_lane[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) kf_lane[par] = l15 * GE::ROW_BYTES + (((4 * par + grp) ^ (l15 & 7)) * 16);
  // transposed V reads: chunk (2dt + (tp>>1)) ^ (row & 7), row = 32kk + 16hf + 4grp + tq  ->  row & 7 = (4grp + tq) & 7
  // depends on dt only through dt & 3: 4 x 4 per-lane bases + the compile-time offset (dt >> 2) * 128
  // (rows 16 apart share row & 7, so the four reads of a d tile differ by compile-time offsets only)
  unsigned tr_lane[4];
#pragma unroll
  for (int dl = 0; dl < 4; ++dl) {
    const int row = 4 * grp + tq;
    tr_lane[dl] = row * GE::ROW_BYTES + (((2 * dl + (tp >> 1)) ^ (row & 7)) * 16) + (tp & 1) * 8;
  }

  if (n_kt > 0) {
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  for (int kt = 0; kt < n_kt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < n_kt) stage(kt + 1, buf ^ 1);
    if (active) {
      const lds_m* ktile = smem + buf * GE::TILE_BYTES;
      const unsigned vt = smem_u32 + buf * GE::TILE_BYTES;
      f32x4 s[NQ][4];
      // chunk (4ks + grp) ^ (row & 7): the XOR touches the low 3 bits only and row & 7 == l15 & 7 for every
      // key tile, so two per-lane bases (k-step parity) + compile-time offsets address every fragment
      const lds_m* kbase[2] = {ktile + kf_lane[0], ktile + kf_lane[1]};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int c = 0; c < NQ; ++c) s[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < GE::NK; ++ks) {
          const frag kf = *reinterpret_cast<const __attribute__((address_space(3))) frag*>(kbase[ks & 1] + t * 16 * GE::ROW_BYTES + (ks >> 1) * 128);
#pragma unroll
          for (int c = 0; c < NQ; ++c) s[c][t] = mla_mfma<T>::run(kf, qf[c][ks], s[c][t]);
        }
      }
      // lane: head column l15, keys k_begin + kt*64 + 16t + 4*grp + r
      const int key0 = k_begin + kt * MLA_KEYS + 4 * grp;
      const bool tail = k_begin + (kt + 1) * MLA_KEYS > k_end;
      frag pf[NQ][2];
#pragma unroll
      for (int c = 0; c < NQ; ++c) {
        f32x4 (&sc)[4] = s[c];
        float mx = m[c];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = s[c][t][r] * a.scale_log2;
            if (tail && key0 + 16 * t + r >= k_end) v = -INFINITY;
            sc[t][r] = v;
            mx = fmaxf(mx, v);
          }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float ms = mx == -INFINITY ? 0.f : mx;
        const float alpha = exp2f(m[c] - ms);
        m[c] = mx;
        float ps = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          frag f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p0 = exp2f(sc[2 * kk][r] - ms), p1 = exp2f(sc[2 * kk + 1][r] - ms);
            ps += p0 + p1;
            f[r] = static_cast<T>(p0);
            f[4 + r] = static_cast<T>(p1);
          }
          pf[c][kk] = f;
        }
        lsum[c] = lsum[c] * alpha + ps;
        if (!__all(alpha == 1.0f)) {                       // the running max moved for some row of this wave
#pragma unroll
          for (int dt = 0; dt < GE::ND; ++dt) o[c][dt] *= alpha;
        }
      }
      // O^T += C_kv^T P^T: per d tile four transposed reads (both 32-key steps) and their MFMAs.  Reads and their
      // wait live in ONE asm statement: scalar loads share lgkmcnt and return out of order, so a counted wait
      // across statements is not safe in compiler-scheduled code; latency is covered by the partner wave.
#pragma unroll
      for (int dt = 0; dt < GE::ND; ++dt) {
        s16x4 v4[4];
        asm volatile(
            "ds_read_b64_tr_b16 %0, %4 offset:%5\n\t"
            "ds_read_b64_tr_b16 %1, %4 offset:%6\n\t"
            "ds_read_b64_tr_b16 %2, %4 offset:%7\n\t"
            "ds_read_b64_tr_b16 %3, %4 offset:%8\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(v4[0]), "=&v"(v4[1]), "=&v"(v4[2]), "=&v"(v4[3])
            : "v"(vt + tr_lane[dt & 3]), "i"((dt >> 2) * 128), "i"((dt >> 2) * 128 + 16 * GE::ROW_BYTES),
              "i"((dt >> 2) * 128 + 32 * GE::ROW_BYTES), "i"((dt >> 2) * 128 + 48 * GE::ROW_BYTES)
            : "memory");
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const s16x8 both = {v4[kk * 2][0], v4[kk * 2][1], v4[kk * 2][2], v4[kk * 2][3],
                              v4[kk * 2 + 1][0], v4[kk * 2 + 1][1], v4[kk * 2 + 1][2], v4[kk * 2 + 1][3]};
#pragma unroll
          for (int c = 0; c < NQ; ++c) o[c][dt] = mla_mfma<T>::run(__builtin_bit_cast(frag, both), pf[c][kk], o[c][dt]);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  if (!active) return;
#pragma unroll
  for (int c = 0; c < NQ; ++c) {
    float ls = lsum[c];
    ls += __shfl_xor(ls, 16);
    ls += __shfl_xor(ls, 32);
    if ((wave * NQ + c) * 16 + l15 >= a.heads) continue;
    const int hd = head[c];
    if (a.n_splits == 1) {
      // finish here: apply the optional sink logit, normalise; rows without keys become zeros (nan_to_num)
      float den = ls;
      float w = 1.f;
      if (a.sink) {
        const float sk = a.sink[hd] * 1.4426950408889634f;
        const float M = fmaxf(m[c], sk);
        w = (m[c] == -INFINITY) ? 0.f : exp2f(m[c] - M);
        den = ls * w + exp2f(sk - M);
      }
      const float inv = den > 0.f ? w / den : 0.f;
      T* dst = static_cast<T*>(a.o_lat) + (static_cast<int64_t>(tile) * a.heads + hd) * R;
      typedef typename vec_of<T, 4>::type V4;
#pragma unroll
      for (int dt = 0; dt < GE::ND; ++dt) {
        V4 ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) ov[r] = static_cast<T>(o[c][dt][r] * inv);
        *reinterpret_cast<V4*>(dst + dt * 16 + grp * 4) = ov;
      }
    } else {
      const int64_t slot = (static_cast<int64_t>(tile) * a.n_splits + split) * a.heads + hd;
      float* po = a.part_o + slot * R;
#pragma unroll
      for (int dt = 0; dt < GE::ND; ++dt) *reinterpret_cast<f32x4*>(po + dt * 16 + grp * 4) = o[c][dt];
      if (grp == 0) {
        a.part_ml[slot * 2] = m[c];
        a.part_ml[slot * 2 + 1] = ls;
      }
    }
  }
}

}  // namespace mojo
#include "mla512_oct.h"
#ifdef MOJO_HIP_BUILD_EXPERIMENTS        // measured-slower kernels of rounds 2-3 (DESIGN Appendix A #12): opt-in build only
#include "experiments/mla512_pair.h"
#include "experiments/mla512_pp.h"
#endif
#include "mla512_ps.h"
namespace mojo {

// merge the splits of one (token, head): grid = (Tq, H), a workgroup of NL split lanes x 128 threads (4 latent elements each),
// NL = blockDim.x / 128 = 1 (up to four splits: the headline shapes, where three idle lanes per row would cost more in wave
// launches than they save) or 4.  Split lane j takes the splits j, j + NL, ...; the partial sums meet in LDS and are added in
// lane order.  (One thread
// per four elements used to walk all splits through dependent loads: a single sequence of 32K tokens — 128 splits — spent
// 31 us here, next to 44 us in the attention kernel.)
template <typename T>
__global__ __launch_bounds__(512) void mla_merge_kernel(MlaArgs a, int R) {
  __shared__ float s_m[4], s_den[4];
  __shared__ f32x4 s_num[4][128];
  const int tile = blockIdx.x, head = blockIdx.y;
  const int sl = threadIdx.x >> 7, dt = threadIdx.x & 127, nl = static_cast<int>(blockDim.x >> 7);
  const int d0 = dt * 4;
  const bool live = d0 < R;
  // prefill: tokens no sequence owns were skipped by the attention launch (no partials written); their rows stay zero
  if (a.cu_q && (tile < a.cu_q[0] || tile >= a.cu_q[a.batch])) return;          // (workgroup-uniform)
  if (nl == 1 && a.n_splits <= 4) {
    // few splits (the decode batch: two): every load of the item is requested before the first is used — the general loop below
    // is three dependent round trips (maxima, then per split its statistics and, behind a branch, its partial).  Same
    // arithmetic in the same order.
    float ms[4], ls[4];
    f32x4 po[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t slot = (static_cast<int64_t>(tile) * a.n_splits + min(i, a.n_splits - 1)) * a.heads + head;
      ms[i] = a.part_ml[slot * 2];
      ls[i] = a.part_ml[slot * 2 + 1];
      po[i] = live ? *reinterpret_cast<const f32x4*>(a.part_o + slot * R + d0) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float M1 = -INFINITY;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < a.n_splits) M1 = fmaxf(M1, ms[i]);
    float sk1 = 0.f;
    if (a.sink) {
      sk1 = a.sink[head] * 1.4426950408889634f;
      M1 = fmaxf(M1, sk1);
    }
    f32x4 num1 = {0.f, 0.f, 0.f, 0.f};
    float den1 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i >= a.n_splits || ms[i] == -INFINITY) continue;
      const float w = exp2f(ms[i] - M1);
      den1 = fmaf(w, ls[i], den1);
      if (live) num1 += po[i] * w;
    }
    if (!live) return;
    den1 = (a.sink ? exp2f(sk1 - M1) : 0.f) + den1;
    const float inv1 = den1 > 0.f ? 1.0f / den1 : 0.f;
    typedef typename vec_of<T, 4>::type V4;
    V4 ov;
#pragma unroll
    for (int e = 0; e < 4; ++e) ov[e] = static_cast<T>(num1[e] * inv1);
    *reinterpret_cast<V4*>(static_cast<T*>(a.o_lat) + (static_cast<int64_t>(tile) * a.heads + head) * R + d0) = ov;
    return;
  }
  float M = -INFINITY;
  for (int sp = sl; sp < a.n_splits; sp += nl) M = fmaxf(M, a.part_ml[((static_cast<int64_t>(tile) * a.n_splits + sp) * a.heads + head) * 2]);
  if (dt == 0) s_m[sl] = M;
  __syncthreads();
  M = s_m[0];
  for (int j = 1; j < nl; ++j) M = fmaxf(M, s_m[j]);
  float sk = 0.f;
  if (a.sink) {
    sk = a.sink[head] * 1.4426950408889634f;
    M = fmaxf(M, sk);
  }
  f32x4 num = {0.f, 0.f, 0.f, 0.f};
  float den = 0.f;
  for (int sp = sl; sp < a.n_splits; sp += nl) {
    const int64_t slot = (static_cast<int64_t>(tile) * a.n_splits + sp) * a.heads + head;
    const float ms = a.part_ml[slot * 2];
    if (ms == -INFINITY) continue;
    const float w = exp2f(ms - M);
    den = fmaf(w, a.part_ml[slot * 2 + 1], den);
    if (live) num += *reinterpret_cast<const f32x4*>(a.part_o + slot * R + d0) * w;
  }
  s_num[sl][dt] = num;
  if (dt == 0) s_den[sl] = den;
  __syncthreads();
  if (sl != 0 || !live) return;
  den = a.sink ? exp2f(sk - M) : 0.f;
  for (int j = 0; j < nl; ++j) { den += s_den[j]; if (j) num += s_num[j][dt]; }
  const float inv = den > 0.f ? 1.0f / den : 0.f;
  typedef typename vec_of<T, 4>::type V4;
  V4 ov;
#pragma unroll
  for (int e = 0; e < 4; ++e) ov[e] = static_cast<T>(num[e] * inv);
  *reinterpret_cast<V4*>(static_cast<T*>(a.o_lat) + (static_cast<int64_t>(tile) * a.heads + head) * R + d0) = ov;
}

template <typename T, int R, int ROPE>
static int launch_mla(const MlaArgs& a, hipStream_t s) {
  typedef mla_geom<R, ROPE> GE;
  constexpr int NQ = 1;
  auto* fn = mla_latent_kernel<T, R, ROPE, NQ>;
  static std::atomic<uint64_t> attr_set{0};
  if (first_call_on_device(attr_set)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, GE::LDS_BYTES);
  hipLaunchKernelGGL(fn, dim3(a.n_tiles, a.n_splits), dim3(512 / NQ), GE::LDS_BYTES, s, a);
  MOJO_CHECK_LAUNCH("mla_latent");
  note_launch("mla_latent:r%d:splits%d", R, a.n_splits);
  if (a.n_splits > 1) {
    hipLaunchKernelGGL(mla_merge_kernel<T>, dim3(a.n_tiles, a.heads), dim3(a.n_splits <= 4 ? 128 : 512), 0, s, a, R);
    MOJO_CHECK_LAUNCH("mla_merge");
  }
  return MOJO_OK;
}

template <typename T>
static int dispatch_mla(const MlaArgs& a, int r, int rope, hipStream_t s) {
  if (r == 512 && rope == 64 && a.page_shift >= 0) {
    const int head_blocks = (a.heads + 63) / 64;
    // MOJO_HIP_MLA_KERNEL: "ps" specialised waves (default), "oct" two waves per SIMD in lock-step on 64-key tiles (the
    // kernel for pages below 16 tokens); an experiments build adds "pp" (ping-pong) and "pair" (one wave per SIMD)
    int which = [] {
      const long long e = MOJO_SWITCH("MOJO_HIP_MLA_KERNEL", -1);
      if (e < 0) return MLA512_DEFAULT_KERNEL;
      if (e == switch_word("ps")) return 3;
#ifdef MOJO_HIP_BUILD_EXPERIMENTS
      if (e == switch_word("pair")) return 1;
      if (e == switch_word("pp")) return 2;
#endif
      return 0;                                          // "oct"
    }();
    if (which == 3 && a.page_shift < 4) which = 0;       // a loader's 16 rows must share one page id: pages of >= 16 tokens
    if (which == 3) {
      void (*fn)(MlaArgs) = mla512_ps_kernel<T>;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, MLA512_PS_LDS);
      hipLaunchKernelGGL(fn, dim3(a.n_tiles * head_blocks, a.n_splits), dim3(512), MLA512_PS_LDS, s, a);
#ifdef MOJO_HIP_BUILD_EXPERIMENTS
    } else if (which == 1) {
      void (*fn)(MlaArgs) = mla512_pair_kernel<T>;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, MLA512_PAIR_LDS);
      hipLaunchKernelGGL(fn, dim3(a.n_tiles * head_blocks, a.n_splits), dim3(256), MLA512_PAIR_LDS, s, a);
    } else if (which == 2) {
      void (*fn)(MlaArgs) = mla512_pp_kernel<T>;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, MLA512_PP_LDS);
      hipLaunchKernelGGL(fn, dim3(a.n_tiles * head_blocks, a.n_splits), dim3(512), MLA512_PP_LDS, s, a);
#endif
    } else {
      void (*fn)(MlaArgs) = mla512_oct_kernel<T>;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, MLA512_OCT_LDS);
      hipLaunchKernelGGL(fn, dim3(a.n_tiles * head_blocks, a.n_splits), dim3(512), MLA512_OCT_LDS, s, a);
    }
    MOJO_CHECK_LAUNCH("mla512");
    note_launch("mla512:%s:splits%d", which == 3 ? "ps" : which == 1 ? "pair" : which == 2 ? "pp" : "oct", a.n_splits);
    if (a.n_splits > 1) {
      hipLaunchKernelGGL(mla_merge_kernel<T>, dim3(a.n_tiles, a.heads), dim3(a.n_splits <= 4 ? 128 : 512), 0, s, a, 512);
      MOJO_CHECK_LAUNCH("mla_merge");
    }
    return MOJO_OK;
  }
  if (r == 64 && rope == 32) return launch_mla<T, 64, 32>(a, s);
  if (r == 32 && rope == 32) return launch_mla<T, 32, 32>(a, s);
  if (r == 256 && rope == 64) return launch_mla<T, 256, 64>(a, s);
  if (r == 128 && rope == 64) return launch_mla<T, 128, 64>(a, s);
  MOJO_REQUIRE(false, MOJO_EUNSUPPORTED, "mla: (kv_lora_rank, rope) = (%d, %d) not instantiated", r, rope);
}

static int mla_splits(int64_t tiles, int64_t max_len) {
  if (const int v = static_cast<int>(MOJO_SWITCH("MOJO_HIP_MLA_SPLITS", 0)); v >= 1) return v;
  int64_t sp = 256 / (tiles > 0 ? tiles : 1);
  const int64_t cap = ceil_div(max_len > 0 ? max_len : 1, 256);
  if (sp > cap) sp = cap;
  if (sp < 1) sp = 1;
  if (sp > 64) sp = 64;
  return static_cast<int>(sp);
}

}  // namespace mojo

using namespace mojo;

extern "C" int64_t mojo_hip_mla_latent_attn_workspace_bytes(int64_t q_tokens, int64_t heads, int64_t kv_lora_rank,
                                                            int64_t max_kv_len) {
  const int sp = mla_splits(q_tokens * (kv_lora_rank == 512 ? (heads + 63) / 64 : 1), max_kv_len);
  if (sp == 1) return 64;
  return q_tokens * sp * heads * (kv_lora_rank + 2) * static_cast<int64_t>(sizeof(float)) + 64;
}

extern "C" int mojo_hip_mla_latent_attn(const void* q_lat, int64_t q_lat_stride, const void* q_rope,
                                        int64_t q_rope_stride, const void* ckv_cache, const void* kpe_cache,
                                        const int32_t* total_seq_lens, const int32_t* cu_q_lens,
                                        const int32_t* cu_total_seq_lens, const int32_t* block_tables,
                                        const float* attn_sink, void* o_lat, void* workspace, int64_t workspace_bytes,
                                        int64_t q_tokens, int64_t batch, int64_t heads, int64_t kv_lora_rank,
                                        int64_t rope_dim, int64_t block_size, int64_t max_blocks_per_seq,
                                        int64_t block_table_stride, int64_t ckv_block_stride, int64_t ckv_token_stride,
                                        int64_t kpe_block_stride, int64_t kpe_token_stride, int64_t max_kv_len,
                                        float softmax_scale, int dtype, mojo_stream_t stream) {
  if (q_tokens == 0) return MOJO_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  MOJO_REQUIRE(q_lat && ckv_cache && kpe_cache && block_tables && o_lat && (total_seq_lens || cu_q_lens), MOJO_EINVAL,
               "mla_latent_attn: null pointer");
  MOJO_REQUIRE(dtype == MOJO_BF16 || dtype == MOJO_F16, MOJO_EUNSUPPORTED, "mla_latent_attn: dtype %d (bf16/fp16 only)", dtype);
  MOJO_REQUIRE(heads >= 1 && heads <= 128, MOJO_EUNSUPPORTED, "mla_latent_attn: heads %lld (1..128)", (long long)heads);
  MOJO_REQUIRE(ckv_token_stride % 8 == 0 && kpe_token_stride % 8 == 0 && ckv_block_stride % 8 == 0 && kpe_block_stride % 8 == 0 &&
                   aligned_to(ckv_cache, 16) && aligned_to(kpe_cache, 16) && aligned_to(q_lat, 16) && aligned_to(o_lat, 8) && q_lat_stride % 8 == 0 &&
                   (!q_rope || (aligned_to(q_rope, 16) && q_rope_stride % 8 == 0)),
               MOJO_EUNSUPPORTED, "mla_latent_attn: tensors must be 16-byte aligned with 16-byte row strides");
  MOJO_REQUIRE(q_tokens < (1 << 30), MOJO_EUNSUPPORTED, "mla_latent_attn: too many query tokens");
  const int64_t eb = 2;
  // prefill: query tokens outside [cu_q[0], cu_q[batch]) are never visited by a workgroup — zero them up front.
  // Decode visits every token (an empty sequence writes zeros itself), so the fill would only cost a launch.
  if (cu_q_lens && hipMemsetAsync(o_lat, 0, static_cast<size_t>(q_tokens * heads * kv_lora_rank * eb), s) != hipSuccess) {
    set_error("mla_latent_attn: memset failed");
    return MOJO_ELAUNCH;
  }
  MlaArgs a;
  a.q_lat = q_lat; a.q_stride = q_lat_stride;
  if (q_rope) { a.q_rope = q_rope; a.q_rope_stride = q_rope_stride; }
  else { a.q_rope = static_cast<const char*>(q_lat) + kv_lora_rank * 2; a.q_rope_stride = q_lat_stride; }
  a.ckv = ckv_cache; a.kpe = kpe_cache; a.o_lat = o_lat;
  a.seq_lens = total_seq_lens; a.cu_q = cu_q_lens; a.cu_kv = cu_total_seq_lens; a.tables = block_tables; a.sink = attn_sink;
  a.table_stride = block_table_stride; a.ckv_blk = ckv_block_stride; a.ckv_tok = ckv_token_stride;
  a.kpe_blk = kpe_block_stride; a.kpe_tok = kpe_token_stride;
  a.heads = static_cast<int>(heads); a.page = static_cast<int>(block_size);
  a.page_shift = (block_size & (block_size - 1)) == 0 ? __builtin_ctzll(block_size) : -1;
  a.max_pages = static_cast<int>(max_blocks_per_seq); a.batch = static_cast<int>(batch);
  a.n_tiles = static_cast<int>(q_tokens);
  const int64_t cap = block_size * max_blocks_per_seq;
  const int64_t max_len = (max_kv_len > 0 && max_kv_len < cap) ? max_kv_len : cap;
  // the r = 512 kernel runs one workgroup per 64 heads: count those when sizing the split
  a.n_splits = mla_splits(q_tokens * ((kv_lora_rank == 512 && rope_dim == 64) ? (heads + 63) / 64 : 1), max_len);
  a.split_keys = static_cast<int>(ceil_div(ceil_div(max_len > 0 ? max_len : 1, a.n_splits), MLA_KEYS) * MLA_KEYS);
  a.scale_log2 = softmax_scale * 1.4426950408889634f;
  a.ps_debug = 0; a.ps_issuers = 2; a.ps_prefetch = 0;
#ifdef MOJO_HIP_BUILD_EXPERIMENTS           // ablation switches of the `ps` kernel
  a.ps_debug = static_cast<int>(MOJO_SWITCH("MOJO_HIP_MLA_PS_DEBUG", 0));
  a.ps_issuers = MOJO_SWITCH("MOJO_HIP_MLA_PS_ISSUERS", 2) == 6 ? 6 : 2;
  { const long long e = MOJO_SWITCH("MOJO_HIP_MLA_PS_PREFETCH", 0); a.ps_prefetch = e == 1 ? 1 : e == 2 ? 2 : 0; }
#endif
  a.part_o = nullptr; a.part_ml = nullptr;
  if (a.n_splits > 1) {
    const int64_t slots = q_tokens * a.n_splits * heads;
    const int64_t need = slots * (kv_lora_rank + 2) * static_cast<int64_t>(sizeof(float));
    MOJO_REQUIRE(workspace && workspace_bytes >= need && aligned_to(workspace, 16), MOJO_EWORKSPACE,
                 "mla_latent_attn: workspace %lld B < required %lld B", (long long)workspace_bytes, (long long)need);
    a.part_o = static_cast<float*>(workspace);
    a.part_ml = a.part_o + slots * kv_lora_rank;
  }
  const int r = static_cast<int>(kv_lora_rank), rope = static_cast<int>(rope_dim);
  return dtype == MOJO_BF16 ? dispatch_mla<bf16_t>(a, r, rope, s) : dispatch_mla<f16_t>(a, r, rope, s);
}


#ifdef MLA_STAMPS
extern "C" int mojo_hip_debug_mla_stamps(unsigned* host_out, int64_t count) {
  if (hipDeviceSynchronize() != hipSuccess) return MOJO_ELAUNCH;
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mojo::g_mla_stamps), static_cast<size_t>(count) * 4) == hipSuccess ? MOJO_OK : MOJO_ELAUNCH;
}
#endif
