// MojoPagedDecodeGQA — split-KV flash decoding for gfx950.
//
// Work decomposition
//   grid = (chunks, B * Hkv); one 64-lane wave per (sequence, kv-head, chunk of the KV range).
//   The G = Hq/Hkv query heads that share a kv-head are processed together, so every K/V byte is
//   read from HBM exactly once (the reference's Triton kernel launches (B, Hq) and re-reads the
//   cache G times: backends/ttx/kernels/ilu/flash_attention.py:818).
//
// Data layout in the wave
//   A (page, head) slab of the cache is [page tokens][D] contiguous.  LPT = 16 lanes cover one token
//   row with 16 B (8 elements) each, so one wave-wide load instruction moves 64/LPT = 4 consecutive
//   tokens = up to 1 KiB fully coalesced.  A tile is 4 such loads of K and 4 of V (16 tokens);
//   two tiles are kept in flight (register double buffer) — K/V go straight to VGPRs, there is no
//   reuse to stage through LDS.
//   lane = r * LPT + j :  r = token slot inside a load (0..3), j = 8-element slice of the head dim.
//   Each lane row r runs its own online softmax over tokens {4i + r}; the 4 partial states are merged
//   once at the end.  Scores: v_dot2c_f32_bf16 partial dots + a 4-step DPP butterfly inside the row.
//
// Numerics: fp32 scores/softmax/accumulators; the golden rounds scores and probabilities to the
// storage dtype, so parity is by tolerance (atol = rtol = 2e-2 in the reference test).
//
// Algorithmic bytes per launch: sum_b len_b * Hkv * D * 2(K,V) * elt  +  2 * B * Hq * D * elt
//                               + 4 * B * (max_blocks + 1).
#include <math.h>
#include <stdlib.h>

#include "common.h"

namespace mojo {

constexpr int DEC_LPT = 16;               // lanes per token row
constexpr int DEC_TPL = 64 / DEC_LPT;     // tokens per wave-wide load
constexpr int DEC_LOADS = 4;              // loads per tile
constexpr int DEC_TILE = DEC_TPL * DEC_LOADS;   // 16 tokens

template <typename T> struct pack8;
template <> struct pack8<bf16_t> {
  typedef bf16x8 vec;
  typedef bf16x2 pair;
  static __device__ __forceinline__ float dot2(pair a, pair b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(a, b, c, false);
  }
};
template <> struct pack8<f16_t> {
  typedef f16x8 vec;
  typedef f16x2 pair;
  static __device__ __forceinline__ float dot2(pair a, pair b, float c) { return __builtin_amdgcn_fdot2(a, b, c, false); }
};

struct DecodeArgs {
  const void* q;
  const void* kc;
  const void* vc;
  const int32_t* seq_lens;
  const int32_t* tables;
  void* out;
  float* ws_acc;     // [B*Hkv][chunks][G][D] fp32, un-normalised
  float* ws_ml;      // [B*Hkv][chunks][G][2]  (running max in log2 units, running sum)
  int hq, hkv, dim, page, page_shift, max_pages, batch;   // hkv: kv heads the GRID sees (= real kv heads << hshift)
  int hshift;                     // 1: groups of 8 query heads run as two 4-head halves of one kv head (see launch), else 0
  int64_t table_stride, c_blk, c_head, c_tok;
  int chunk_tokens, n_chunks;     // launch-wide bound: no sequence is cut into more than n_chunks pieces of <= chunk_tokens
  float scale_log2;
  int abab;
  int leave_empty;                // rows with seq_len <= 0: 1 = leave the output row untouched (graph replay contract), 0 = zeros
  int fuse_group = 0;             // > 0: GROUPED form — workgroups of this many waves each own that many consecutive chunks of a row,
                                  // merge them in LDS and leave ONE partial per workgroup (slot = workgroup index along x) for the
                                  // merge launch; a row whose chunks fit one workgroup is finished there
};

// Chunking is PER SEQUENCE: a sequence of `len` tokens is cut into n_chunks equal pieces (whole tiles, at least 128 tokens,
// never more than the launch-wide chunk_tokens), so the waves of a short row of a ragged batch all work (and finish early)
// instead of some of them idling while the others walk max_len / n_chunks tokens.  Lengths beyond what the launch was sized
// for (the caller's max_total_seq_len hint, or page * max_blocks) are truncated to that capacity: the grid, the LDS image and
// the workspace hold n_chunks slots per row and nothing may index past them.
__device__ __forceinline__ int decode_seq_len(const DecodeArgs& a, int b) {
  return min(a.seq_lens[b], a.n_chunks * a.chunk_tokens);
}
__device__ __forceinline__ int decode_seq_chunk(const DecodeArgs& a, int seq_len) {
  int c = (seq_len + a.n_chunks - 1) / a.n_chunks;
  c = max(c, 128);
  c = ((c + DEC_TILE - 1) / DEC_TILE) * DEC_TILE;
  return min(c, a.chunk_tokens);
}

// MODE 1 (fused): one workgroup = all chunks of one (sequence, kv-head), one wave per chunk (n_chunks <= 8); the partial
// states meet in LDS and the workgroup writes the final output itself — no partials in HBM, no merge launch.
// MODE 2 (fused, paired): an 8-wave workgroup owns TWO sequences of one kv-head — the one of rank p and the one of rank
// B-1-p when the batch is ordered by length — and deals its waves between them in proportion to their lengths.  Every wave
// of the launch is resident at once, so a launch lasts as long as its busiest wave: with four waves per sequence that is
// max_len / 4 tokens; a long row paired with a short one leaves every wave ~(len_long + len_short) / 8, the batch mean / 4.
// A uniform batch gets 4 + 4 waves per pair, i.e. exactly the MODE 1 work per wave.
constexpr int DEC_SPLIT = 0, DEC_FUSED = 1, DEC_PAIRED = 2;

// query head of (grid kv head kvh, head g of the kernel's G).  With hshift = 1 the grid's kv heads are the halves of the real
// ones: real kv head kvh >> 1, heads (kvh & 1) * G + g of its 2 G.
__device__ __forceinline__ int decode_head(const DecodeArgs& a, int kvh, int g, int G) {
  if (!a.abab) return kvh * G + g;                        // AABB: (real kv head, half, g) is already kvh * G + g
  const int real = kvh >> a.hshift, half = kvh & ((1 << a.hshift) - 1);
  return (half * G + g) * (a.hkv >> a.hshift) + real;
}

template <typename T, int G, bool NT, int MODE>
__global__ __launch_bounds__(MODE != DEC_SPLIT ? 512 : 64) void decode_split_kernel(DecodeArgs a) {
  constexpr bool FUSED = MODE != DEC_SPLIT;
  constexpr bool PAIRED = MODE == DEC_PAIRED;
  typedef typename pack8<T>::vec V8;
  typedef typename pack8<T>::pair V2;
  const int lane = threadIdx.x & 63;
  const int r = lane / DEC_LPT;
  const int j = lane % DEC_LPT;
  // (readfirstlane: the wave index must be a scalar, or every page-table lookup below turns into a vector load)
  const int wave_id = FUSED ? __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6)) : 0;
  int chunk = FUSED ? wave_id : static_cast<int>(blockIdx.x);
  int b = blockIdx.y / a.hkv;
  const int kvh = blockIdx.y % a.hkv;

  int seq_len, chunk_tokens;
  // paired mode: the two sequences of this workgroup, their (clamped) lengths and the waves dealt to the first
  int pb[2] = {0, -1}, plen[2] = {0, 0}, pchunk[2] = {DEC_TILE, DEC_TILE}, n_first = 8;
  if constexpr (PAIRED) {
    const int cap = a.n_chunks * a.chunk_tokens;
    int len = -1;                                      // lanes past the batch rank behind every sequence
    if (lane < a.batch) len = a.max_pages > 0 ? max(min(a.seq_lens[lane], cap), 0) : 0;
    int rank = lane;                                   // position of sequence `lane` when ordered by length, longest first
    if (__ballot(lane < a.batch && len != __builtin_amdgcn_readfirstlane(len)) != 0) {   // (a uniform batch keeps its order)
      rank = 0;
      for (int o = 0; o < a.batch; ++o) {
        const int lo = __builtin_amdgcn_readlane(len, o);
        rank += (lo > len || (lo == len && o < lane)) ? 1 : 0;
      }
    }
    const int p = b;                                   // blockIdx.y / hkv is the pair index here
    const unsigned long long first = __ballot(lane < a.batch && rank == p);
    const unsigned long long second = __ballot(lane < a.batch && rank == a.batch - 1 - p && a.batch - 1 - p > p);
    pb[0] = __builtin_ctzll(first);
    plen[0] = __builtin_amdgcn_readlane(len, pb[0]);
    if (second) {
      pb[1] = __builtin_ctzll(second);
      plen[1] = __builtin_amdgcn_readlane(len, pb[1]);
    }
    const int sum = plen[0] + plen[1];
    n_first = plen[1] <= 0 ? 8 : min(max((8 * plen[0] + sum / 2) / sum, 1), 7);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int n_u = u ? 8 - n_first : n_first;
      int c = n_u > 0 ? (plen[u] + n_u - 1) / n_u : DEC_TILE;
      c = max(c, 128);
      pchunk[u] = ((c + DEC_TILE - 1) / DEC_TILE) * DEC_TILE;
    }
    const int u = wave_id < n_first ? 0 : 1;
    b = pb[u];
    chunk = u ? wave_id - n_first : wave_id;
    seq_len = b >= 0 ? plen[u] : 0;
    chunk_tokens = pchunk[u];
    if (b < 0) b = pb[0];                              // a wave without a sequence: valid addresses, no work
  } else {
    seq_len = a.max_pages > 0 ? decode_seq_len(a, b) : 0;      // (no table columns: nothing to attend over)
    chunk_tokens = decode_seq_chunk(a, seq_len);
  }
  const int tok_begin = chunk * chunk_tokens;
  const bool has_work = seq_len > 0 && tok_begin < seq_len;
  if (!FUSED && !has_work) return;
  const int tok_end = has_work ? min(seq_len, tok_begin + chunk_tokens) : tok_begin + 1;
  const bool dim_ok = j * 8 < a.dim;
  const int jd = dim_ok ? j * 8 : a.dim - 8;          // lanes past a short head re-read its last slice

  // query slices for the G heads of this kv-head (AABB: h = kvh*G + g, ABAB: h = g*Hkv + kvh).  The 8-head instance of the
  // workgroup forms parks them in LDS (its own 16 lanes x 16 B per head, read back four heads at a time): with them in
  // registers it needs ~10 more than a wave has at two waves per SIMD.
  constexpr bool Q_LDS = FUSED && G >= 8;
  V8 qv[Q_LDS ? 1 : G];
  extern __shared__ float s_part[];                      // [waves][G][dim + 2] partials (+ [waves][G][16 lanes] query slices)
  V8* const q_lds = reinterpret_cast<V8*>(s_part + (blockDim.x >> 6) * G * (a.dim + 2)) + (wave_id * G) * DEC_LPT + j;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int h = decode_head(a, kvh, g, G);
    V8 z = {};
    const V8 x = *reinterpret_cast<const V8*>(static_cast<const T*>(a.q) + (static_cast<int64_t>(b) * a.hq + h) * a.dim + jd);
    if constexpr (Q_LDS) { if (r == 0) q_lds[g * DEC_LPT] = dim_ok ? x : z; }
    else qv[g] = dim_ok ? x : z;
  }

  float m[G], l[G], acc[G][8];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    m[g] = -INFINITY;
    l[g] = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[g][e] = 0.f;
  }

  const int32_t* table = a.tables + static_cast<int64_t>(b) * a.table_stride;
  // The golden walks the pages in order and stops at the first negative id, leaving every later row zero
  // (core/operators/attention.py:195-198).  `first_neg` = that page index among the pages up to the end of this chunk.
  // The scan (64 table entries per load, one ballot each) does NOT gate the K/V loads: a load only needs its own table
  // entry (a negative id is clamped to page 0, an address that certainly exists) and remembers its logical page; whether
  // it must read as zero (logical page >= first_neg) is decided when the tile is consumed.  So the prologue is one
  // round trip — scan batch, query and the first two tiles in flight together — instead of up to four dependent ones
  // in front of the first K/V byte (measured fixed cost per call before: ~18 us).
  int p1 = (tok_end + a.page - 1) / a.page;
  int first_neg = 0x7fffffff;
  if (p1 > a.max_pages) { first_neg = a.max_pages; p1 = a.max_pages; }
  constexpr int SCAN = 4;                               // table loads in flight per scan step (256 pages)
  int scan_v[SCAN];
  auto scan_issue = [&](int base) {
#pragma unroll
    for (int u = 0; u < SCAN; ++u) {
      const int idx = base + u * 64 + lane;
      scan_v[u] = idx < p1 ? table[idx] : 0;
    }
  };
  auto scan_reduce = [&](int base) {
#pragma unroll
    for (int u = 0; u < SCAN; ++u) {
      const unsigned long long neg = __ballot(scan_v[u] < 0);
      if (neg && first_neg == 0x7fffffff) first_neg = base + u * 64 + __builtin_ctzll(neg);
    }
  };
  if (has_work) scan_issue(0);

  const T* kbase = static_cast<const T*>(a.kc) + (kvh >> a.hshift) * a.c_head + jd;
  const T* vbase = static_cast<const T*>(a.vc) + (kvh >> a.hshift) * a.c_head + jd;
  const int last_load = ((tok_end - 1) / DEC_TPL) * DEC_TPL;   // first token of the last non-empty load
  const int last_page = a.max_pages - 1;

  struct Tile { V8 k[DEC_LOADS]; V8 v[DEC_LOADS]; int lp[DEC_LOADS]; };

  auto ld = [&](const T* p) -> V8 {
    if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const V8*>(p));
    else return *reinterpret_cast<const V8*>(p);
  };

  // Branch-free: every load is issued (clamped to an address that certainly exists); its logical page is kept so that
  // `process` can tell which loads must read as zero (pages at or behind the first negative id).
  auto load_tile = [&](Tile& t, int t0) {
#pragma unroll
    for (int u = 0; u < DEC_LOADS; ++u) {
      const int tu = min(t0 + u * DEC_TPL, last_load);    // wave-uniform; TPL | page
      const int lp = a.page_shift >= 0 ? (tu >> a.page_shift) : tu / a.page;
      t.lp[u] = lp;
      const int phys = max(table[min(lp, last_page)], 0);
      const int64_t off = static_cast<int64_t>(phys) * a.c_blk + static_cast<int64_t>(tu - lp * a.page + r) * a.c_tok;
      t.k[u] = ld(kbase + off);
      t.v[u] = ld(vbase + off);
    }
  };

  auto process = [&](Tile& t, int t0) {
    if (t.lp[DEC_LOADS - 1] >= first_neg) {              // rare: pages behind a hole read as zeros (lp grows with u)
#pragma unroll
      for (int u = 0; u < DEC_LOADS; ++u)
        if (t.lp[u] >= first_neg) { V8 z = {}; t.k[u] = z; t.v[u] = z; }
    }
    const bool full = t0 + DEC_TILE <= tok_end;          // wave-uniform
    if (!full) {                                         // last tile of the chunk: slots past the length may hold NaN/Inf
#pragma unroll
      for (int u = 0; u < DEC_LOADS; ++u) {
        V8 z = {};
        if (!((t0 + u * DEC_TPL + r) < tok_end)) t.v[u] = z;
      }
    }
    // Heads in blocks of GB = 4: eight heads at once need 32 score registers on top of 64 accumulators, 32 query registers
    // and the tile ring — more than a wave has at two waves per SIMD (the 8-head instance spilled: B 64, ctx 4096, 64 / 8
    // heads 876 us).  Scores and probabilities of one block are dead before the next block's are formed.
    constexpr int GB = G < 4 ? G : 4;
#pragma unroll
    for (int g0 = 0; g0 < G; g0 += GB) {
      float s[DEC_LOADS][GB];
      V8 qb[GB];
#pragma unroll
      for (int gg = 0; gg < GB; ++gg) {
        if constexpr (Q_LDS) qb[gg] = q_lds[(g0 + gg) * DEC_LPT];   // (written by this wave's own lanes 0-15: no barrier)
        else qb[gg] = qv[g0 + gg];
      }
#pragma unroll
      for (int u = 0; u < DEC_LOADS; ++u) {
#pragma unroll
        for (int gg = 0; gg < GB; ++gg) {
          float d = 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            V2 qa = {qb[gg][2 * e], qb[gg][2 * e + 1]};
            V2 ka = {t.k[u][2 * e], t.k[u][2 * e + 1]};
            d = pack8<T>::dot2(qa, ka, d);
          }
          s[u][gg] = row16_sum(d) * a.scale_log2;
        }
      }
      if (!full) {                                       // mask the tail
#pragma unroll
        for (int u = 0; u < DEC_LOADS; ++u) {
          const bool valid = (t0 + u * DEC_TPL + r) < tok_end;
#pragma unroll
          for (int gg = 0; gg < GB; ++gg) s[u][gg] = valid ? s[u][gg] : -INFINITY;
        }
      }
#pragma unroll
      for (int gg = 0; gg < GB; ++gg) {
        const int g = g0 + gg;
        float mx = m[g];
#pragma unroll
        for (int u = 0; u < DEC_LOADS; ++u) mx = fmaxf(mx, s[u][gg]);
        const float ms = mx == -INFINITY ? 0.f : mx;     // row r may not have seen a valid token yet
        const float alpha = exp2f(m[g] - ms);
        m[g] = mx;
        float p[DEC_LOADS];
        float ps = 0.f;
#pragma unroll
        for (int u = 0; u < DEC_LOADS; ++u) {
          p[u] = exp2f(s[u][gg] - ms);
          ps += p[u];
        }
        l[g] = l[g] * alpha + ps;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float x = acc[g][e] * alpha;
#pragma unroll
          for (int u = 0; u < DEC_LOADS; ++u) x = fmaf(p[u], static_cast<float>(t.v[u][e]), x);
          acc[g][e] = x;
        }
      }
      if constexpr (GB < G) asm volatile("" ::: "memory");   // keep the blocks apart (the scheduler would interleave them again)
    }
  };

  // register ring of tiles: three (prefetch distance two tiles, 16 KiB in flight per wave) — two for the 8-head instance, whose
  // accumulators take the third tile's registers
  constexpr int RING = G >= 8 ? 2 : 3;
  Tile ta, tb, tc;
  if (has_work) {
  load_tile(ta, tok_begin);
  if (tok_begin + DEC_TILE < tok_end) load_tile(tb, tok_begin + DEC_TILE);
  scan_reduce(0);
  for (int base = 64 * SCAN; base < p1 && first_neg == 0x7fffffff; base += 64 * SCAN) {   // contexts past 256 pages
    scan_issue(base);
    scan_reduce(base);
  }
  if constexpr (RING == 3) {
    for (int t0 = tok_begin; t0 < tok_end; t0 += 3 * DEC_TILE) {
      if (t0 + 2 * DEC_TILE < tok_end) load_tile(tc, t0 + 2 * DEC_TILE);
      process(ta, t0);
      if (t0 + DEC_TILE >= tok_end) break;
      if (t0 + 3 * DEC_TILE < tok_end) load_tile(ta, t0 + 3 * DEC_TILE);
      process(tb, t0 + DEC_TILE);
      if (t0 + 2 * DEC_TILE >= tok_end) break;
      if (t0 + 4 * DEC_TILE < tok_end) load_tile(tb, t0 + 4 * DEC_TILE);
      process(tc, t0 + 2 * DEC_TILE);
    }
  } else {
    for (int t0 = tok_begin; t0 < tok_end; t0 += 2 * DEC_TILE) {
      process(ta, t0);
      if (t0 + DEC_TILE >= tok_end) break;
      if (t0 + 2 * DEC_TILE < tok_end) load_tile(ta, t0 + 2 * DEC_TILE);
      process(tb, t0 + DEC_TILE);
      if (t0 + 3 * DEC_TILE < tok_end) load_tile(tb, t0 + 3 * DEC_TILE);
    }
  }
  }

  // merge the DEC_TPL lane rows (lanes j, j+16, j+32, j+48 hold the same head-dim slice)
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float mx = m[g];
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float ms = mx == -INFINITY ? 0.f : mx;
    const float w = exp2f(m[g] - ms);
    float lw = l[g] * w;
    lw += __shfl_xor(lw, 16);
    lw += __shfl_xor(lw, 32);
    l[g] = lw;
    m[g] = mx;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float x = acc[g][e] * w;
      x += __shfl_xor(x, 16);
      x += __shfl_xor(x, 32);
      acc[g][e] = x;
    }
  }

  if constexpr (FUSED) {
    const int stride = a.dim + 2;
    if (r == 0 && dim_ok) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        float* dst = s_part + (wave_id * G + g) * stride;     // (wave_id == chunk in the unpaired form)
#pragma unroll
        for (int e = 0; e < 8; ++e) dst[j * 8 + e] = acc[g][e];
        if (j == 0) { dst[a.dim] = m[g]; dst[a.dim + 1] = l[g]; }
      }
    }
    __syncthreads();
    // every thread of the workgroup takes (sequence of the pair, head g, 4 output elements) items
    const int per_head = a.dim / 4;
    typedef typename vec_of<T, 4>::type V4;
    constexpr int UNITS = PAIRED ? 2 : 1;
    for (int item = threadIdx.x; item < UNITS * G * per_head; item += blockDim.x) {
      const int u = item / (G * per_head);
      const int rest = item - u * (G * per_head);
      const int g = rest / per_head, d0 = (rest - g * per_head) * 4;
      int ub, ulen, uchunk, slot0, uwaves;
      if constexpr (PAIRED) {
        ub = pb[u]; ulen = plen[u]; uchunk = pchunk[u];
        slot0 = u ? n_first : 0;
        uwaves = u ? 8 - n_first : n_first;
        if (ub < 0) continue;                            // odd batch: the middle sequence has no partner
      } else {
        ub = b; ulen = seq_len; uchunk = chunk_tokens; slot0 = 0; uwaves = static_cast<int>(blockDim.x >> 6);
      }
      const int n_chunks_seq = ulen <= 0 ? 0 : min((ulen + uchunk - 1) / uchunk, uwaves);
      if (n_chunks_seq == 0 && a.leave_empty) continue;
      const int h = decode_head(a, kvh, g, G);
      float mx = -INFINITY;
      for (int c = 0; c < n_chunks_seq; ++c) mx = fmaxf(mx, s_part[((slot0 + c) * G + g) * stride + a.dim]);
      f32x4 num = {0.f, 0.f, 0.f, 0.f};
      float den = 0.f;
      for (int c = 0; c < n_chunks_seq; ++c) {
        const float* src = s_part + ((slot0 + c) * G + g) * stride;
        const float w = exp2f(src[a.dim] - mx);
        den = fmaf(w, src[a.dim + 1], den);
        num += f32x4{src[d0], src[d0 + 1], src[d0 + 2], src[d0 + 3]} * w;
      }
      const float inv = n_chunks_seq > 0 ? 1.0f / den : 0.f;      // empty sequence: zeros (golden semantics)
      V4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = static_cast<T>(num[e] * inv);
      *reinterpret_cast<V4*>(static_cast<T*>(a.out) + (static_cast<int64_t>(ub) * a.hq + h) * a.dim + d0) = o;
    }
    return;
  }
  if (r != 0 || !dim_ok) return;
  const int n_chunks_seq = (seq_len + chunk_tokens - 1) / chunk_tokens;
  if (n_chunks_seq == 1) {
    // single chunk: finish here, the merge kernel skips this row
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int h = decode_head(a, kvh, g, G);
      const float inv = 1.0f / l[g];
      V8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = static_cast<T>(acc[g][e] * inv);
      *reinterpret_cast<V8*>(static_cast<T*>(a.out) + (static_cast<int64_t>(b) * a.hq + h) * a.dim + j * 8) = o;
    }
    return;
  }
  const int64_t slot = (static_cast<int64_t>(blockIdx.y) * a.n_chunks + chunk) * G;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float* dst = a.ws_acc + (slot + g) * a.dim + j * 8;
    *reinterpret_cast<f32x4*>(dst) = f32x4{acc[g][0], acc[g][1], acc[g][2], acc[g][3]};
    *reinterpret_cast<f32x4*>(dst + 4) = f32x4{acc[g][4], acc[g][5], acc[g][6], acc[g][7]};
    if (j == 0) {
      a.ws_ml[(slot + g) * 2 + 0] = m[g];
      a.ws_ml[(slot + g) * 2 + 1] = l[g];
    }
  }
}

}  // namespace mojo
#include "paged_decode_mfma.h"
namespace mojo {

// Merge the chunk partials: grid = (B*Hkv, G), a workgroup of 8 chunk lanes x 32 threads (4 output elements each); rows with
// seq_len <= 0 become zeros (golden semantics), single-chunk rows were finished by the split kernel.  Chunk lane j takes
// the chunks j, j + 8, ...; the eight partial sums meet in LDS and are added in lane order (fixed, so the bits do not depend
// on timing).  (One thread used to walk all chunks of a row through dependent loads: a single long sequence — B = 1,
// 16K tokens, 128 chunks — spent ~38 of its 54 us here.)
template <typename T>
__global__ __launch_bounds__(256) void decode_merge_kernel(DecodeArgs a, int G) {
  __shared__ float s_mx[8], s_den[8];
  __shared__ f32x4 s_num[8][32];
  const int b = blockIdx.x / a.hkv;
  const int kvh = blockIdx.x % a.hkv;
  const int g = blockIdx.y;
  const int seq_len = decode_seq_len(a, b);
  const int chunk_tokens = decode_seq_chunk(a, seq_len);
  int n_chunks_seq = seq_len <= 0 ? 0 : (seq_len + chunk_tokens - 1) / chunk_tokens;   // <= a.n_chunks by construction
  if (a.fuse_group > 0) n_chunks_seq = (n_chunks_seq + a.fuse_group - 1) / a.fuse_group;   // grouped form: one partial per workgroup
  if (n_chunks_seq == 1) return;                         // (workgroup-uniform)
  const int h = decode_head(a, kvh, g, G);
  const int cl = threadIdx.x >> 5, dt = threadIdx.x & 31;
  const int d0 = dt * 4;
  const bool live = d0 < a.dim;
  T* dst = static_cast<T*>(a.out) + (static_cast<int64_t>(b) * a.hq + h) * a.dim + d0;
  typedef typename vec_of<T, 4>::type V4;
  if (n_chunks_seq == 0) {
    if (a.leave_empty || cl != 0 || !live) return;
    V4 z = {};
    *reinterpret_cast<V4*>(dst) = z;
    return;
  }
  const int64_t base = static_cast<int64_t>(blockIdx.x) * a.n_chunks * G + g;
  float mx = -INFINITY;
  for (int c = cl; c < n_chunks_seq; c += 8) mx = fmaxf(mx, a.ws_ml[(base + static_cast<int64_t>(c) * G) * 2]);
  if (dt == 0) s_mx[cl] = mx;
  __syncthreads();
  mx = s_mx[0];
#pragma unroll
  for (int j = 1; j < 8; ++j) mx = fmaxf(mx, s_mx[j]);
  f32x4 num = {0.f, 0.f, 0.f, 0.f};
  float den = 0.f;
#pragma unroll 4
  for (int c = cl; c < n_chunks_seq; c += 8) {
    const int64_t slot = base + static_cast<int64_t>(c) * G;
    const float w = exp2f(a.ws_ml[slot * 2] - mx);
    den = fmaf(w, a.ws_ml[slot * 2 + 1], den);
    if (live) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(a.ws_acc + slot * a.dim + d0);
      num += x * w;
    }
  }
  s_num[cl][dt] = num;
  if (dt == 0) s_den[cl] = den;
  __syncthreads();
  if (cl != 0 || !live) return;
#pragma unroll
  for (int j = 1; j < 8; ++j) { num += s_num[j][dt]; den += s_den[j]; }
  const float inv = 1.0f / den;
  V4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = static_cast<T>(num[e] * inv);
  *reinterpret_cast<V4*>(dst) = o;
}

// The grouped form (DecodeArgs::fuse_group; matrix-core kernel only: the same conditions as decode_use_mfma, which the
// workspace sizing cannot call) — MOJO_HIP_DECODE_GROUPED=0 restores the split + merge form.
static bool decode_grouped(int64_t G, int64_t head_dim, int64_t page) {
  // Measured (scripts/probes/decode_grouped_ab.py, profiles/r4_decode_grouped_ab.txt; split + merge form -> grouped form, graph
  // replay): 8 q / 1 kv heads B 64 ctx 4096 (Llama-3-70B under TP 8) 33.6 -> 28.1 us, B 16 ctx 16384 38.2 -> 29.8 us; 32 / 8 heads
  // B 8 ctx 4096 31.3 -> 27.5 us; 64 / 8 heads B 8 ctx 8192 56.1 -> 49.8 us; head_dim 64 B 8 ctx 8192 33.7 -> 28.6 us.
  // (the grouped launch exists in the fused form only: without MOJO_HIP_DECODE_FUSE the chunks are sized, written and merged
  // one by one — ADVICE r4: the merge kernel divided the chunk count by the group size although every chunk had its partial)
  if (MOJO_SWITCH("MOJO_HIP_DECODE_GROUPED", 1) == 0 || MOJO_SWITCH("MOJO_HIP_DECODE_FUSE", 1) == 0) return false;
  const bool pow2 = page > 0 && (page & (page - 1)) == 0;
  if (!pow2 || page < 16 || (head_dim != 64 && head_dim != 128) || G > 16) return false;
  const long long m = MOJO_SWITCH("MOJO_HIP_DECODE_MFMA", -1);
  if (m == 0) return false;
  if (m == 1) return true;
  return G >= 4 || head_dim == 64 || (G != 1 && G != 2);
}

static int decode_chunk_tokens(int64_t batch, int64_t kv_heads, int64_t max_len, bool grouped = false) {
  {                                                            // tuning override
    const int v = static_cast<int>(MOJO_SWITCH("MOJO_HIP_DECODE_CHUNK", 0));
    if (v >= DEC_TILE) return (v / DEC_TILE) * DEC_TILE;
  }
  // The split kernel holds 2 waves per SIMD (register-ring bound), i.e. 2048 resident waves on 256
  // CUs.  Cut each sequence into as few chunks as still fill those slots once: long-lived waves
  // amortise their ramp-up and leave few partials to merge.  Never below 128 tokens per chunk.
  const int64_t units = batch * kv_heads > 0 ? batch * kv_heads : 1;
  int64_t splits = (256 * 8) / units;
  if (splits < 1) splits = 1;
  const int64_t len = max_len > 0 ? max_len : 1;
  int64_t chunk = ceil_div(len, splits);
  if (chunk < 128) chunk = 128;
  chunk = ceil_div(chunk, DEC_TILE) * DEC_TILE;
  // More than 8 chunks per (sequence, kv-head) take the split + merge form: single-wave workgroups and a merge launch whose
  // cost grows with the chunk count.  There 1 024 waves measured best, not 2 048 (round 3, scripts/probes/decode_tp8_chunk_ab.py,
  // matrix-core kernel: 8 q / 1 kv heads B 64 ctx 4096 39.8 -> 34.3 us, B 16 ctx 16384 52.1 -> 38.9 us, 64 / 8 heads B 8 ctx 8192
  // 64.2 -> 57.4 us, 32 / 8 heads B 8 ctx 4096 36.8 -> 32.6 us; 512 waves: 1.5-2 x slower) - but never fewer than 8 chunks.
  if (ceil_div(len, chunk) > 8) {
    // (grouped form: the partials are merged in LDS eight at a time, so the 2 048 waves that fill the chip cost no more partials
    // than 256 single-wave workgroups per row would)
    int64_t s2 = (grouped ? 2048 : 1024) / units;
    if (s2 < 8) s2 = 8;
    chunk = ceil_div(len, s2);
    if (chunk < 128) chunk = 128;
    chunk = ceil_div(chunk, DEC_TILE) * DEC_TILE;
  }
  return static_cast<int>(chunk);
}

static int64_t decode_max_len(int64_t page, int64_t max_pages, int64_t hint) {
  const int64_t cap = page * max_pages;
  return (hint > 0 && hint < cap) ? hint : cap;
}

// The matrix-core kernel (paged_decode_mfma.h): pages of a multiple of 16 tokens, head_dim 64 / 128, groups of <= 16 heads.
// MOJO_HIP_DECODE_MFMA: unset = where it measured faster (B 64, ctx 4096, page 16, graph replay, vector-unit -> matrix-core):
// groups of 8 heads (Llama-3-70B 64 / 8: 356 -> 170 us; 8 / 1: 79 -> 39 us), head_dim 64 (149 -> 99 us), groups of 4 at
// head_dim 128 (headline 179 -> 169 us, ctx 1024 51 -> 47 us, ragged 150 -> 139 us; B 8: 34.5 -> 32.6 us with the chunk rule
// of decode_chunk_tokens), and the group sizes the vector-unit kernel has no instance for.  Groups of 1 and 2 heads at head_dim 128
// stay on the vector-unit kernel (163 vs 168 us, 167 vs 169 us).  1 = wherever it applies, 0 = never.
static bool decode_use_mfma(const DecodeArgs& a, int G) {
  if (a.page_shift < 4 || (a.dim != 64 && a.dim != 128) || G > 16) return false;
  const long long e = MOJO_SWITCH("MOJO_HIP_DECODE_MFMA", -1);
  if (e == 0) return false;
  if (e == 1) return true;
  return G >= 4 || a.dim == 64 || (G != 1 && G != 2);
}

template <typename T, bool NT, int MODE>
static void launch_decode_mfma(const DecodeArgs& a, dim3 grid, dim3 block, int G, hipStream_t s) {
  const size_t waves = block.x / 64;
  const size_t lds = (MODE != DEC_SPLIT ? waves * G * (a.dim + 2) * sizeof(float) : 0) + waves * 4096;   // + one 4 KiB V image (set) per wave
  static std::atomic<uint64_t> attr_set{0};             // (per instantiation: dynamic LDS beyond 64 KiB needs the attribute)
  if (first_call_on_device(attr_set)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_mfma_kernel<T, 4, NT, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_mfma_kernel<T, 2, NT, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  if (a.dim == 128) hipLaunchKernelGGL((decode_mfma_kernel<T, 4, NT, MODE>), grid, block, lds, s, a, G);
  else hipLaunchKernelGGL((decode_mfma_kernel<T, 2, NT, MODE>), grid, block, lds, s, a, G);
}

template <typename T, bool NT>
static int launch_decode_nt(DecodeArgs& a, int64_t batch, int G, hipStream_t s) {
  const bool no_fuse = MOJO_SWITCH("MOJO_HIP_DECODE_FUSE", 1) == 0;
  const bool no_pair = MOJO_SWITCH("MOJO_HIP_DECODE_PAIR", 1) == 0;
  const char* nt_tag = NT ? "nt" : "cached";
  if (decode_use_mfma(a, G)) {
    if (a.n_chunks == 4 && batch >= 2 && batch <= 64 && !no_fuse && !no_pair) {
      launch_decode_mfma<T, NT, DEC_PAIRED>(a, dim3(1, static_cast<unsigned>(((batch + 1) / 2) * a.hkv)), dim3(512), G, s);
      MOJO_CHECK_LAUNCH("paged_decode_gqa(mfma, paired)");
      note_launch("decode_mfma:paired:%s", nt_tag);
      return MOJO_OK;
    }
    if (a.n_chunks <= 8 && !no_fuse) {
      launch_decode_mfma<T, NT, DEC_FUSED>(a, dim3(1, static_cast<unsigned>(batch * a.hkv)), dim3(static_cast<unsigned>(64 * a.n_chunks)), G, s);
      MOJO_CHECK_LAUNCH("paged_decode_gqa(mfma, fused)");
      note_launch("decode_mfma:fused:%s", nt_tag);
      return MOJO_OK;
    }
    if (a.fuse_group > 0 && !no_fuse) {
      // small grids (few (sequence, kv-head) rows, many chunks each): eight-wave workgroups merge their chunks in LDS and leave
      // one partial each, so the launch keeps two waves per SIMD busy and the merge reads n_chunks / 8 partials per row
      const unsigned n_sub = static_cast<unsigned>((a.n_chunks + a.fuse_group - 1) / a.fuse_group);
      launch_decode_mfma<T, NT, DEC_FUSED>(a, dim3(n_sub, static_cast<unsigned>(batch * a.hkv)), dim3(static_cast<unsigned>(64 * a.fuse_group)), G, s);
      MOJO_CHECK_LAUNCH("paged_decode_gqa(mfma, grouped)");
      hipLaunchKernelGGL((decode_merge_kernel<T>), dim3(static_cast<unsigned>(batch * a.hkv), G), dim3(256), 0, s, a, G);
      MOJO_CHECK_LAUNCH("paged_decode_gqa(merge)");
      note_launch("decode_mfma:grouped+merge:%s", nt_tag);
      return MOJO_OK;
    }
    launch_decode_mfma<T, NT, DEC_SPLIT>(a, dim3(static_cast<unsigned>(a.n_chunks), static_cast<unsigned>(batch * a.hkv)), dim3(64), G, s);
    MOJO_CHECK_LAUNCH("paged_decode_gqa(mfma, split)");
    hipLaunchKernelGGL((decode_merge_kernel<T>), dim3(static_cast<unsigned>(batch * a.hkv), G), dim3(256), 0, s, a, G);
    MOJO_CHECK_LAUNCH("paged_decode_gqa(merge)");
    note_launch("decode_mfma:split+merge:%s", nt_tag);
    return MOJO_OK;
  }
  if (a.n_chunks == 4 && batch >= 2 && batch <= 64 && a.dim % 4 == 0 && !no_fuse && !no_pair) {
    // four waves per sequence fill the chip once: pair the sequences by length rank and deal each pair's 8 waves by length
    dim3 grid(1, static_cast<unsigned>(((batch + 1) / 2) * a.hkv));
    const size_t lds = static_cast<size_t>(8) * G * (a.dim + 2) * sizeof(float) + (G >= 8 ? static_cast<size_t>(8) * G * DEC_LPT * 16 : 0);
    switch (G) {
      case 1: hipLaunchKernelGGL((decode_split_kernel<T, 1, NT, DEC_PAIRED>), grid, dim3(512), lds, s, a); break;
      case 2: hipLaunchKernelGGL((decode_split_kernel<T, 2, NT, DEC_PAIRED>), grid, dim3(512), lds, s, a); break;
      case 4: hipLaunchKernelGGL((decode_split_kernel<T, 4, NT, DEC_PAIRED>), grid, dim3(512), lds, s, a); break;
      case 8: hipLaunchKernelGGL((decode_split_kernel<T, 8, NT, DEC_PAIRED>), grid, dim3(512), lds, s, a); break;
      default: MOJO_REQUIRE(false, MOJO_EUNSUPPORTED, "paged_decode_gqa: group size %d (supported: 1,2,4,8)", G);
    }
    MOJO_CHECK_LAUNCH("paged_decode_gqa(paired)");
    note_launch("decode_valu:paired:%s", nt_tag);
    return MOJO_OK;
  }
  if (a.n_chunks <= 8 && a.dim % 4 == 0 && !no_fuse) {   // all chunks of a (sequence, kv-head) in one workgroup: merged in LDS
    dim3 grid(1, static_cast<unsigned>(batch * a.hkv));
    const dim3 block(static_cast<unsigned>(64 * a.n_chunks));
    const size_t lds = static_cast<size_t>(a.n_chunks) * G * (a.dim + 2) * sizeof(float) +
                       (G >= 8 ? static_cast<size_t>(a.n_chunks) * G * DEC_LPT * 16 : 0);
    switch (G) {
      case 1: hipLaunchKernelGGL((decode_split_kernel<T, 1, NT, DEC_FUSED>), grid, block, lds, s, a); break;
      case 2: hipLaunchKernelGGL((decode_split_kernel<T, 2, NT, DEC_FUSED>), grid, block, lds, s, a); break;
      case 4: hipLaunchKernelGGL((decode_split_kernel<T, 4, NT, DEC_FUSED>), grid, block, lds, s, a); break;
      case 8: hipLaunchKernelGGL((decode_split_kernel<T, 8, NT, DEC_FUSED>), grid, block, lds, s, a); break;
      default: MOJO_REQUIRE(false, MOJO_EUNSUPPORTED, "paged_decode_gqa: group size %d (supported: 1,2,4,8)", G);
    }
    MOJO_CHECK_LAUNCH("paged_decode_gqa(fused)");
    note_launch("decode_valu:fused:%s", nt_tag);
    return MOJO_OK;
  }
  dim3 grid(static_cast<unsigned>(a.n_chunks), static_cast<unsigned>(batch * a.hkv));
  switch (G) {
    case 1: hipLaunchKernelGGL((decode_split_kernel<T, 1, NT, DEC_SPLIT>), grid, dim3(64), 0, s, a); break;
    case 2: hipLaunchKernelGGL((decode_split_kernel<T, 2, NT, DEC_SPLIT>), grid, dim3(64), 0, s, a); break;
    case 4: hipLaunchKernelGGL((decode_split_kernel<T, 4, NT, DEC_SPLIT>), grid, dim3(64), 0, s, a); break;
    case 8: hipLaunchKernelGGL((decode_split_kernel<T, 8, NT, DEC_SPLIT>), grid, dim3(64), 0, s, a); break;
    default: MOJO_REQUIRE(false, MOJO_EUNSUPPORTED, "paged_decode_gqa: group size %d (supported: 1,2,4,8)", G);
  }
  MOJO_CHECK_LAUNCH("paged_decode_gqa(split)");
  hipLaunchKernelGGL((decode_merge_kernel<T>), dim3(static_cast<unsigned>(batch * a.hkv), G), dim3(256), 0, s, a, G);
  MOJO_CHECK_LAUNCH("paged_decode_gqa(merge)");
  note_launch("decode_valu:split+merge:%s", nt_tag);
  return MOJO_OK;
}

template <typename T>
static int launch_decode(DecodeArgs& a, int64_t batch, int G, hipStream_t s) {
  // K/V are read exactly once: non-temporal loads keep them from displacing the block tables and
  // partials in L2/MALL (measured on MI355X, B=64 ctx=4096: 204 -> 190 us).  MOJO_HIP_STREAM_NT=0 disables (the switch of every
  // streaming kernel, common.h stream_nt()).
  const bool nt = MOJO_SWITCH("MOJO_HIP_STREAM_NT", -1) != 0;
  return nt ? launch_decode_nt<T, true>(a, batch, G, s) : launch_decode_nt<T, false>(a, batch, G, s);
}

}  // namespace mojo

using namespace mojo;

extern "C" int64_t mojo_hip_paged_decode_gqa_workspace_bytes(int64_t batch, int64_t q_heads, int64_t kv_heads,
                                                             int64_t head_dim, int64_t block_size,
                                                             int64_t max_blocks_per_seq, int64_t max_seq_len_hint) {
  if (batch <= 0 || kv_heads <= 0 || q_heads <= 0) return 0;
  const int64_t max_len = decode_max_len(block_size, max_blocks_per_seq, max_seq_len_hint);
  // (groups of 8 query heads can run as two 4-head halves on twice the grid heads — MOJO_HIP_DECODE_G8_HALVES=1 —: size for
  // whichever form cuts the sequences finer)
  int64_t slots = 0;
  for (int64_t mult = 1; mult <= (q_heads / kv_heads == 8 ? 2 : 1); ++mult) {
    const int chunk = decode_chunk_tokens(batch, kv_heads * mult, max_len, decode_grouped(q_heads / (kv_heads * mult), head_dim, block_size));
    const int64_t n_chunks = ceil_div(max_len > 0 ? max_len : 1, chunk);
    const int64_t sl = batch * kv_heads * n_chunks * (q_heads / kv_heads);
    if (sl > slots) slots = sl;
  }
  return slots * (head_dim + 2) * static_cast<int64_t>(sizeof(float)) + 256;
}

extern "C" int mojo_hip_paged_decode_gqa(const void* query, const void* key_cache, const void* value_cache,
                                         const int32_t* total_seq_lens, const int32_t* block_tables, void* out,
                                         void* workspace, int64_t workspace_bytes, int64_t batch, int64_t q_heads,
                                         int64_t kv_heads, int64_t head_dim, int64_t block_size,
                                         int64_t max_blocks_per_seq, int64_t block_table_stride,
                                         int64_t cache_block_stride, int64_t cache_head_stride,
                                         int64_t cache_token_stride, int64_t max_seq_len_hint, float softmax_scale,
                                         int layout_abab, int leave_empty_rows, int dtype, mojo_stream_t stream) {
  if (batch == 0) return MOJO_OK;
  MOJO_REQUIRE(query && key_cache && value_cache && total_seq_lens && block_tables && out, MOJO_EINVAL,
               "paged_decode_gqa: null pointer");
  MOJO_REQUIRE(batch > 0 && q_heads > 0 && kv_heads > 0 && q_heads % kv_heads == 0, MOJO_EINVAL,
               "paged_decode_gqa: bad head counts Hq=%lld Hkv=%lld", (long long)q_heads, (long long)kv_heads);
  MOJO_REQUIRE(dtype == MOJO_BF16 || dtype == MOJO_F16, MOJO_EUNSUPPORTED,
               "paged_decode_gqa: dtype %d (bf16/fp16 only)", dtype);
  MOJO_REQUIRE(head_dim % 8 == 0 && head_dim >= 8 && head_dim <= 8 * DEC_LPT, MOJO_EUNSUPPORTED,
               "paged_decode_gqa: head_dim %lld (multiple of 8, <= %d)", (long long)head_dim, 8 * DEC_LPT);
  MOJO_REQUIRE(block_size % DEC_TPL == 0, MOJO_EUNSUPPORTED, "paged_decode_gqa: block_size %lld must be a multiple of %d",
               (long long)block_size, DEC_TPL);
  MOJO_REQUIRE(cache_token_stride % 8 == 0 && cache_head_stride % 8 == 0 && cache_block_stride % 8 == 0 &&
                   aligned_to(key_cache, 16) && aligned_to(value_cache, 16) && aligned_to(query, 16) &&
                   aligned_to(out, 16),
               MOJO_EUNSUPPORTED, "paged_decode_gqa: tensors must be 16-byte aligned with 16-byte row strides");
  MOJO_REQUIRE(max_blocks_per_seq >= 0 && batch * kv_heads * 2 <= 65535, MOJO_EUNSUPPORTED,
               "paged_decode_gqa: batch*kv_heads %lld exceeds the grid limit", (long long)(batch * kv_heads));

  DecodeArgs a;
  a.q = query; a.kc = key_cache; a.vc = value_cache; a.seq_lens = total_seq_lens; a.tables = block_tables; a.out = out;
  a.hq = static_cast<int>(q_heads); a.hkv = static_cast<int>(kv_heads); a.dim = static_cast<int>(head_dim);
  a.page = static_cast<int>(block_size); a.max_pages = static_cast<int>(max_blocks_per_seq); a.batch = static_cast<int>(batch);
  a.page_shift = (block_size & (block_size - 1)) == 0 ? __builtin_ctzll(block_size) : -1;
  a.table_stride = block_table_stride; a.c_blk = cache_block_stride; a.c_head = cache_head_stride;
  a.c_tok = cache_token_stride;
  const int64_t max_len = decode_max_len(block_size, max_blocks_per_seq, max_seq_len_hint);
  // Groups of 8 query heads per kv head (Llama-3-70B: 64 / 8).  Round 2 ran them as two 4-head halves on twice as many grid
  // heads (twice the K/V bytes: 360 us at B 64, ctx 4096) because the 8-head instance spilled; the instance now walks its heads
  // in blocks of four over a two-tile ring with the query slices parked in LDS (232 registers, no spill) and reads K/V once.
  // MOJO_HIP_DECODE_G8_HALVES=1 selects the halves again (A/B; experiments build only).
  int G = static_cast<int>(q_heads / kv_heads);
  a.hshift = 0;
#ifdef MOJO_HIP_BUILD_EXPERIMENTS
  if (G == 8 && MOJO_SWITCH("MOJO_HIP_DECODE_G8_HALVES", 0) == 1) { a.hshift = 1; a.hkv *= 2; G = 4; }
#endif
  const bool grouped = decode_grouped(G, head_dim, block_size);
  a.chunk_tokens = decode_chunk_tokens(batch, a.hkv, max_len, grouped);
  a.n_chunks = static_cast<int>(ceil_div(max_len > 0 ? max_len : 1, a.chunk_tokens));
  a.fuse_group = grouped && decode_use_mfma(a, G) && a.n_chunks > 8 ? 8 : 0;   // (the merge divides by it: set only where the grouped launch is taken)
  a.scale_log2 = softmax_scale * 1.4426950408889634f;
  a.abab = layout_abab ? 1 : 0;
  a.leave_empty = leave_empty_rows ? 1 : 0;
  const int64_t slots = static_cast<int64_t>(batch) * a.hkv * a.n_chunks * G;
  const int64_t need = slots * (head_dim + 2) * static_cast<int64_t>(sizeof(float));
  MOJO_REQUIRE(workspace && workspace_bytes >= need, MOJO_EWORKSPACE,
               "paged_decode_gqa: workspace %lld B < required %lld B", (long long)workspace_bytes, (long long)need);
  MOJO_REQUIRE(aligned_to(workspace, 16), MOJO_EINVAL, "paged_decode_gqa: workspace must be 16-byte aligned");
  a.ws_acc = static_cast<float*>(workspace);
  a.ws_ml = a.ws_acc + slots * head_dim;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == MOJO_BF16 ? launch_decode<bf16_t>(a, batch, G, s) : launch_decode<f16_t>(a, batch, G, s);
}
