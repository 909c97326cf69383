// MoE routing either side of the grouped GEMM: MojoMoEGating, MojoMoEDispatch, MojoMoECombine.
// Reference semantics: mojo_opset/core/operators/moe.py:299-316 (gating), :344-400 (dispatch), :687-716 (combine).
//
// All three are HBM-bound byte/integer work (the gate matmul has E <= a few hundred columns):
//   gating   reads T x hidden x elt once, writes 2 x T x k x 4 B;
//   dispatch reads T x H x elt (each row k times, from L2 after the first) and writes T x k x H x elt;
//   combine  reads N x H x elt and writes T x H x elt.
// Nothing here syncs with the host; dispatch and combine are deterministic (no atomics decide an order).
#include <math.h>
#include <stdlib.h>

#include "common.h"
#include "gemm.h"

namespace mojo {

// ---------------------------------------------------------------------------------------------------------
// gating: logits = x.float() @ W (fp32), softmax over E, top-k (descending), gates / sum(selected)
// ---------------------------------------------------------------------------------------------------------
// A block serves TPW = (64 / EP) * TT tokens, EP = min(pow2ceil(E), 64) lanes per token, each lane owning experts
// lane%EP + EP*i (EI of them) for TT tokens: a weight element is loaded once per TT tokens, the activation vector is a
// broadcast load.  The four waves of the block split the hidden dimension; their partial logits meet in LDS, where one
// token's E values then sit on consecutive lanes for the softmax and the k rounds of wave-wide arg-max.
constexpr int GATE_MAX_E = 1024;

// One token: logits = sum of `parts` fp32 partials (stride `part_stride` floats), softmax over E, k rounds of wave-wide
// arg-max (descending value, ties -> lowest expert id), gates renormalised over the selected.  Lane l holds experts
// l, l+64, ...; all 64 lanes of the wave call this together.
__device__ __forceinline__ void gate_softmax_topk(const float* logits, int64_t part_stride, int parts, int experts, int top_k,
                                                  int lane, int32_t* out_idx, float* out_gate) {
  constexpr int PER = GATE_MAX_E / 64;
  float v[PER];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int e = lane + 64 * i;
    if (e < experts) {
      float acc = 0.f;
      for (int p = 0; p + 1 < parts; p += 2) acc += logits[e + p * part_stride] + logits[e + (p + 1) * part_stride];
      if (parts & 1) acc += logits[e + (parts - 1) * part_stride];
      v[i] = acc;
    } else {
      v[i] = -INFINITY;
    }
    mx = fmaxf(mx, v[i]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    v[i] = v[i] == -INFINITY ? 0.f : __expf(v[i] - mx);
    sum += v[i];
  }
  sum = wave_sum(sum);
#pragma unroll
  for (int i = 0; i < PER; ++i) v[i] = (lane + 64 * i < experts) ? v[i] / sum : -1.f;      // -1: not a candidate
  float sel_sum = 0.f, my_val = 0.f;
  int my_idx = 0;
  for (int r = 0; r < top_k; ++r) {
    float best = -1.f;
    int best_e = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < PER; ++i)
      if (v[i] > best) { best = v[i]; best_e = lane + 64 * i; }     // ascending e inside a lane: first maximum wins
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o);
      const int oe = __shfl_xor(best_e, o);
      if (ob > best || (ob == best && oe < best_e)) { best = ob; best_e = oe; }
    }
#pragma unroll
    for (int i = 0; i < PER; ++i)
      if (lane + 64 * i == best_e) v[i] = -1.f;
    sel_sum += best;
    if (lane == r) { my_val = best; my_idx = best_e; }           // top_k <= 64 results, one per lane
  }
  if (lane < top_k) {
    out_idx[lane] = my_idx;
    out_gate[lane] = my_val / sel_sum;
  }
}

template <typename T, int TT, int EI>
__global__ __launch_bounds__(256) void moe_gating_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                         int32_t* __restrict__ out_idx, float* __restrict__ out_gate,
                                                         int64_t tokens, int hidden, int experts, int top_k, int ep_log2) {
  extern __shared__ float s_logits[];                 // [4 waves][TPW][E]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int EP = 1 << ep_log2;
  const int sub = lane >> ep_log2, e0 = lane & (EP - 1);
  const int subs = 64 >> ep_log2;
  const int tpw = subs * TT;
  const int64_t tok0 = static_cast<int64_t>(blockIdx.x) * tpw;
  float* my_logits = s_logits + static_cast<size_t>(wave) * tpw * experts;
  {
    float acc[TT][EI];
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
      for (int i = 0; i < EI; ++i) acc[t][i] = 0.f;
    int64_t tok[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) tok[t] = min(tok0 + sub * TT + t, tokens - 1);     // clamped rows are computed, not stored
    constexpr int VEC = 16 / sizeof(T);
    typedef typename vec_of<T, VEC>::type V;
    if (hidden % VEC == 0) {
      const int per_wave = ((hidden / VEC + 3) / 4) * VEC;              // this wave's slice of the hidden dimension
      const int h_end = min(hidden, (wave + 1) * per_wave);
      for (int h = wave * per_wave; h < h_end; h += VEC) {
        V xv[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) xv[t] = load_vec<T, VEC>(x + tok[t] * hidden + h);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
#pragma unroll
          for (int i = 0; i < EI; ++i) {
            const int e = e0 + i * EP;
            const float wv = e < experts ? w[static_cast<int64_t>(h + j) * experts + e] : 0.f;
#pragma unroll
            for (int t = 0; t < TT; ++t) acc[t][i] = fmaf(elt<T>::to_f(vget<T, VEC>(xv[t], j)), wv, acc[t][i]);
          }
        }
      }
    } else {
      for (int h = wave; h < hidden; h += 4) {
#pragma unroll
        for (int i = 0; i < EI; ++i) {
          const int e = e0 + i * EP;
          const float wv = e < experts ? w[static_cast<int64_t>(h) * experts + e] : 0.f;
#pragma unroll
          for (int t = 0; t < TT; ++t) acc[t][i] = fmaf(elt<T>::to_f(x[tok[t] * hidden + h]), wv, acc[t][i]);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
      for (int i = 0; i < EI; ++i) {
        const int e = e0 + i * EP;
        if (e < experts) my_logits[(sub * TT + t) * experts + e] = acc[t][i];
      }
  }
  __syncthreads();
  // softmax + top-k: wave w takes tokens w, w+4, ... of the block; lane l holds experts l, l+64, ...
  const int part = tpw * experts;                                       // stride between the waves' partial logits
  for (int t = wave; t < tpw; t += 4) {
    const int64_t token = tok0 + t;
    if (token >= tokens) break;
    gate_softmax_topk(s_logits + t * experts, part, 4, experts, top_k, lane, out_idx + token * top_k, out_gate + token * top_k);
  }
}

// Eight experts (Mixtral), 16-bit activations, hidden % 512 == 0: a STREAM of the activations with 8 x 8 fp32
// accumulators per lane.  A block takes 8 tokens, its four waves every fourth 512-wide chunk of the hidden dimension;
// lane l owns hidden elements 8l .. 8l+7 of a chunk, so an activation load is one coalesced 1 KiB instruction per token
// and the matching weight rows (8 rows x 8 experts = 256 contiguous bytes per lane) are sixteen 16-byte loads reused by
// all 8 tokens.  The 64 partial logits of a lane (index = token * 8 + expert) are summed across lanes by lane-row swaps
// (v_permlane32_swap, v_permlane16_swap: the index set a lane keeps halves each time) and a DPP row sum, the four waves'
// sums meet in LDS (added in wave order), and softmax + top-k of a token run inside one lane.
// (The general kernel above assigns a lane per EXPERT: eight lanes load the same 16 bytes of x and every lane issues one
// 4-byte weight load per multiply — 88 us for T = 8192, H = 4096 against 67 MB of activations.)
template <typename T>
__global__ __launch_bounds__(256, 2) void moe_gating_e8_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                               int32_t* __restrict__ out_idx, float* __restrict__ out_gate,
                                                               int64_t tokens, int hidden, int top_k) {
  constexpr int E = 8, TT = 8, VEC = 8;
  typedef typename vec_of<T, VEC>::type V;
  __shared__ float s_part[4][4][2 * E];                                  // [wave][lane row = token pair][token-in-pair * 8 + expert]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t tok0 = static_cast<int64_t>(blockIdx.x) * TT;
  float acc[TT * E];
#pragma unroll
  for (int i = 0; i < TT * E; ++i) acc[i] = 0.f;
  const T* xr[TT];
#pragma unroll
  for (int t = 0; t < TT; ++t) xr[t] = x + min(tok0 + t, tokens - 1) * hidden + lane * VEC;   // clamped rows: computed, not stored
  for (int h0 = wave * 64 * VEC; h0 < hidden; h0 += 4 * 64 * VEC) {
    V xv[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) xv[t] = load_vec<T, VEC>(xr[t] + h0);
    f32x4 wv[2 * VEC];                                                   // w[(h + j) * 8 + e] = wv[2 j + (e >> 2)][e & 3]
    const float* wp = w + (static_cast<int64_t>(h0) + lane * VEC) * E;
#pragma unroll
    for (int i = 0; i < 2 * VEC; ++i) wv[i] = *reinterpret_cast<const f32x4*>(wp + 4 * i);
#pragma unroll
    for (int j = 0; j < VEC; ++j)
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        const float xf = elt<T>::to_f(vget<T, VEC>(xv[t], j));
#pragma unroll
        for (int e = 0; e < E; ++e) acc[t * E + e] = fmaf(xf, wv[2 * j + (e >> 2)][e & 3], acc[t * E + e]);
      }
  }
  // lanes l, l ^ 32: lanes < 32 keep indices i, lanes >= 32 keep i + 32
  float r32[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    float p = acc[i], q = acc[i + 32];
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(p), "+v"(q));
    r32[i] = p + q;
  }
  // lanes l, l ^ 16: even 16-lane rows keep i, odd rows i + 16  ->  row R holds indices 16 R .. 16 R + 15
  const int row = lane >> 4, l15 = lane & 15;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    float p = r32[i], q = r32[i + 16];
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(p), "+v"(q));
    const float v = row16_sum(p + q);
    if (l15 == 0) s_part[wave][row][i] = v;
  }
  __syncthreads();
  // wave w, lanes 0 and 1: tokens 2w and 2w + 1 of the block
  const int64_t token = tok0 + 2 * wave + lane;
  if (lane >= 2 || token >= tokens) return;
  float lg[E];
#pragma unroll
  for (int e = 0; e < E; ++e)
    lg[e] = ((s_part[0][wave][lane * E + e] + s_part[1][wave][lane * E + e]) + s_part[2][wave][lane * E + e]) + s_part[3][wave][lane * E + e];
  float mx = lg[0];
#pragma unroll
  for (int e = 1; e < E; ++e) mx = fmaxf(mx, lg[e]);
  float sum = 0.f;
#pragma unroll
  for (int e = 0; e < E; ++e) { lg[e] = __expf(lg[e] - mx); sum += lg[e]; }
#pragma unroll
  for (int e = 0; e < E; ++e) lg[e] = lg[e] / sum;
  float best_v[E];
  int best_e[E];
  float sel_sum = 0.f;
#pragma unroll
  for (int r = 0; r < E; ++r) {                                           // top_k <= 8 rounds; ties -> lowest expert id
    float b = -1.f;
    int be = 0;
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (lg[e] > b) { b = lg[e]; be = e; }
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (e == be) lg[e] = -1.f;
    best_v[r] = b;
    best_e[r] = be;
    if (r < top_k) sel_sum += b;
  }
#pragma unroll
  for (int r = 0; r < E; ++r)
    if (r < top_k) {
      out_idx[token * top_k + r] = best_e[r];
      out_gate[token * top_k + r] = best_v[r] / sel_sum;
    }
}

// Sum of many slabs + softmax + top-k, one WORKGROUP per token: wave w sums the slabs p = w, w + 4, ... (four loads in
// flight), the four partial sums meet in LDS and wave 0 selects.  (moe_gate_select_kernel gives a token one wave, which
// then walks all slabs through dependent loads: 39 us for 64 slabs x 256 experts.)
__global__ __launch_bounds__(256) void moe_gate_reduce_select_kernel(const float* __restrict__ logits, int64_t part_stride, int parts,
                                                                     int e_pad, int32_t* __restrict__ out_idx,
                                                                     float* __restrict__ out_gate, int experts, int top_k) {
  extern __shared__ float s_lg[];                                         // [4][e_pad]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t token = blockIdx.x;
  const float* src = logits + token * e_pad;
  for (int e0 = lane; e0 < experts; e0 += 256) {                          // four expert slots per lane and pass: 16 loads in flight
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int p = wave; p < parts; p += 16) {
      float v[4][4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int u = 0; u < 4; ++u)
          v[q][u] = (e0 + 64 * q < experts && p + 4 * u < parts) ? src[e0 + 64 * q + (p + 4 * u) * part_stride] : 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = (((acc[q] + v[q][0]) + v[q][1]) + v[q][2]) + v[q][3];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (e0 + 64 * q < experts) s_lg[wave * e_pad + e0 + 64 * q] = acc[q];
  }
  __syncthreads();
  if (wave == 0) gate_softmax_topk(s_lg, e_pad, 4, experts, top_k, lane, out_idx + token * top_k, out_gate + token * top_k);
}

// Few tokens (a decode step): the router is a tiny product — T x H x E multiply-adds against an fp32 weight of 1-7 MB —
// and what matters is spreading it over the chip.  The kernels above put one token (E <= 64) or a few on a workgroup and
// each workgroup walked the WHOLE weight through 4-byte loads: 53 us for 64 tokens x 64 experts, 204 us for 64 x 256 at
// H = 7168 (a DeepSeek-V3 decode step pays that in every MoE layer).  Here the grid is (slice of H, 8 tokens, 256 experts):
// a thread owns one expert and 16 hidden positions per step (its 16 weights are coalesced 4-byte loads across the experts,
// the 8 x 16 activations a broadcast read of an LDS image), a workgroup's step covers 256 / EP sub-slices of 16 positions
// (EP = experts per workgroup, a power of two), fp32 FMAs in index order; the sub-slices are summed in LDS in fixed order,
// the slices go to fp32 slabs [slice][token][e_pad] and moe_gate_select_kernel sums them in slice order.
constexpr int GS_TC = 8;      // tokens per workgroup
constexpr int GS_HB = 16;     // hidden positions per thread and step
constexpr int GS_XC = 512;    // hidden positions of the activation image in LDS (a multiple of every step: 16 .. 256)
struct GateSmallPlan { int ep_log2, slice_len, slices, e_chunks, t_chunks; };
static GateSmallPlan gate_small_plan(int64_t tokens, int64_t hidden, int64_t experts) {
  GateSmallPlan p;
  p.ep_log2 = 4;
  while ((1 << p.ep_log2) < experts && p.ep_log2 < 8) ++p.ep_log2;
  const int ep = 1 << p.ep_log2, step = (256 / ep) * GS_HB;
  p.e_chunks = static_cast<int>(ceil_div(experts, static_cast<int64_t>(ep)));
  p.t_chunks = static_cast<int>(ceil_div(tokens, static_cast<int64_t>(GS_TC)));
  const int64_t max_slices = ceil_div(hidden, static_cast<int64_t>(step));
  int64_t want = 512 / (static_cast<int64_t>(p.e_chunks) * p.t_chunks);       // ~2 workgroups per CU
  if (want > 64) want = 64;
  if (want > max_slices) want = max_slices;
  if (want < 1) want = 1;
  p.slice_len = static_cast<int>(ceil_div(ceil_div(hidden, want), static_cast<int64_t>(step)) * step);
  p.slices = static_cast<int>(ceil_div(hidden, static_cast<int64_t>(p.slice_len)));
  return p;
}

template <typename T>
__global__ __launch_bounds__(256) void moe_gating_small_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                               float* __restrict__ slabs, int tokens, int hidden, int experts,
                                                               int e_pad, int ep_log2, int slice_len) {
  __shared__ __attribute__((aligned(16))) float s_x[GS_TC][GS_XC];        // [token][position inside the chunk]
  __shared__ float s_red[256 * GS_TC];                                    // [sub-slice][token][expert of the workgroup]
  const int ep = 1 << ep_log2, hs_n = 256 >> ep_log2, step = hs_n * GS_HB;
  const int e_local = threadIdx.x & (ep - 1), hs = threadIdx.x >> ep_log2;
  const int e = static_cast<int>(blockIdx.z) * ep + e_local;
  const int t0 = static_cast<int>(blockIdx.y) * GS_TC;
  const int hb0 = static_cast<int>(blockIdx.x) * slice_len, hb1 = min(hidden, hb0 + slice_len);
  float acc[GS_TC];
#pragma unroll
  for (int t = 0; t < GS_TC; ++t) acc[t] = 0.f;
  // the weights of the next step are requested before this step's arithmetic (one step is 16 dependent-free 4-byte loads
  // per thread; without the look-ahead every step waited out a full memory round trip)
  float wn[GS_HB];
  auto load_w = [&](int h0, float (&dst)[GS_HB]) {
    const int hh = h0 + hs * GS_HB;
#pragma unroll
    for (int j = 0; j < GS_HB; ++j) dst[j] = (hh + j < hb1 && e < experts) ? w[static_cast<int64_t>(hh + j) * experts + e] : 0.f;
  };
  load_w(hb0, wn);
  // the activations of up to GS_XC positions are staged at once (one memory round trip and two barriers per chunk, not per step)
  for (int c0 = hb0; c0 < hb1; c0 += GS_XC) {
    const int c1 = min(hb1, c0 + GS_XC);
    const int cw = (c1 - c0 + step - 1) / step * step;                    // positions the steps of this chunk read (zeros behind c1)
    for (int i = threadIdx.x; i < GS_TC * cw; i += 256) {
      const int t = i / cw, c = i - t * cw;
      const int h = c0 + c, tok = t0 + t;
      s_x[t][c] = (h < c1 && tok < tokens) ? elt<T>::to_f(x[static_cast<int64_t>(tok) * hidden + h]) : 0.f;
    }
    __syncthreads();
    for (int h0 = c0; h0 < c1; h0 += step) {
      float wv[GS_HB];
#pragma unroll
      for (int j = 0; j < GS_HB; ++j) wv[j] = wn[j];
      if (h0 + step < hb1) load_w(h0 + step, wn);
#pragma unroll
      for (int t = 0; t < GS_TC; ++t) {
        const f32x4* xp = reinterpret_cast<const f32x4*>(&s_x[t][h0 - c0 + hs * GS_HB]);
#pragma unroll
        for (int q = 0; q < GS_HB / 4; ++q) {
          const f32x4 xv = xp[q];
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[t] = fmaf(xv[c], wv[4 * q + c], acc[t]);
        }
      }
    }
    __syncthreads();
  }
  if (hs_n > 1) {
#pragma unroll
    for (int t = 0; t < GS_TC; ++t) s_red[(hs * GS_TC + t) * ep + e_local] = acc[t];
    __syncthreads();
    if (hs == 0) {
#pragma unroll
      for (int t = 0; t < GS_TC; ++t) {
        float v = acc[t];
        for (int k = 1; k < hs_n; ++k) v += s_red[(k * GS_TC + t) * ep + e_local];
        acc[t] = v;
      }
    }
  }
  if (hs == 0 && e < e_pad) {
#pragma unroll
    for (int t = 0; t < GS_TC; ++t)
      if (t0 + t < tokens) slabs[(static_cast<int64_t>(blockIdx.x) * tokens + t0 + t) * e_pad + e] = acc[t];
  }
}

// large expert counts: logits come from the MFMA GEMM (x @ w_hi + x @ w_lo, `parts` fp32 slabs of [tokens, e_pad]);
// one wave per token
__global__ __launch_bounds__(256) void moe_gate_select_kernel(const float* __restrict__ logits, int64_t part_stride, int parts,
                                                              int e_pad, int32_t* __restrict__ out_idx,
                                                              float* __restrict__ out_gate, int64_t tokens, int experts,
                                                              int top_k) {
  const int lane = threadIdx.x & 63;
  const int64_t token = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (token >= tokens) return;
  gate_softmax_topk(logits + token * e_pad, part_stride, parts, experts, top_k, lane, out_idx + token * top_k,
                    out_gate + token * top_k);
}

// fp32 gate weight [hidden, E] -> hi + lo in the activation dtype, [hidden, e_pad] each (zero padded columns):
// x @ w = x @ hi + x @ lo to ~2^-16 relative, which keeps the expert ranking of the fp32 product
template <typename T>
__global__ __launch_bounds__(256) void moe_gate_split_kernel(const float* __restrict__ w, T* __restrict__ hi, T* __restrict__ lo,
                                                             int hidden, int experts, int e_pad) {
  const int64_t total = static_cast<int64_t>(hidden) * e_pad;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<int64_t>(gridDim.x) * 256) {
    const int e = static_cast<int>(i % e_pad);
    const int64_t h = i / e_pad;
    const float v = e < experts ? w[h * experts + e] : 0.f;
    const T a = elt<T>::from_f(v);
    hi[i] = a;
    lo[i] = elt<T>::from_f(v - elt<T>::to_f(a));
  }
}

// ---------------------------------------------------------------------------------------------------------
// dispatch: stable counting sort of the T*k routing slots by expert id
// ---------------------------------------------------------------------------------------------------------
constexpr int DISP_MAX_E = 4096;

__global__ __launch_bounds__(256) void moe_hist_kernel(const int32_t* __restrict__ ids, int64_t n, int experts,
                                                       int32_t* __restrict__ block_hist) {
  extern __shared__ int s_hist[];
  for (int e = threadIdx.x; e < experts; e += 256) s_hist[e] = 0;
  __syncthreads();
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < n) {
    const int e = ids[i];
    if (e >= 0 && e < experts) atomicAdd(&s_hist[e], 1);           // counts only: the order of the adds is irrelevant
  }
  __syncthreads();
  for (int e = threadIdx.x; e < experts; e += 256) block_hist[static_cast<int64_t>(blockIdx.x) * experts + e] = s_hist[e];
}

// Exclusive prefix of `v` over the threads of the block (<= 1024 threads), wave shuffles + one LDS word per wave.
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int u = __shfl_up(v, o);
    if (lane >= o) v += u;
  }
  return v;
}
__device__ __forceinline__ int block_excl_scan(int v, int* s_wave, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = static_cast<int>(blockDim.x >> 6);
  const int inc = wave_incl_scan(v, lane);
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  int base = 0, tot = 0;
  for (int w = 0; w < nw; ++w) {
    const int c = s_wave[w];
    base += w < wave ? c : 0;
    tot += c;
  }
  __syncthreads();
  if (total) *total = tot;
  return base + inc - v;
}

// one block of 1024 threads: per expert, exclusive prefix over the blocks (in place), totals -> tokens_per_expert, exclusive
// scan -> start.  One WAVE per expert walks the blocks 64 at a time (shuffle scan + carry), two experts per step so that
// two loads are in flight; the first version had one THREAD per expert walking the blocks through dependent loads, and
// thread 0 scanning 256 partial sums through LDS one by one (12.5 us for Mixtral's 64 x 8 counters, now ~3 us).
__global__ __launch_bounds__(1024) void moe_scan_kernel(int32_t* __restrict__ block_hist, int blocks, int experts,
                                                        int32_t* __restrict__ tokens_per_expert, int32_t* __restrict__ expert_start) {
  __shared__ int s_tot[DISP_MAX_E];
  __shared__ int s_wave[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int e0 = 2 * wave; e0 < experts; e0 += 32) {
    const int e1 = e0 + 1 < experts ? e0 + 1 : e0;
    int carry0 = 0, carry1 = 0;
    for (int base = 0; base < blocks; base += 64) {
      const int blk = base + lane;
      const bool live = blk < blocks;
      const int64_t at0 = static_cast<int64_t>(blk) * experts + e0, at1 = static_cast<int64_t>(blk) * experts + e1;
      const int c0 = live ? block_hist[at0] : 0;
      const int c1 = live ? block_hist[at1] : 0;
      const int i0 = wave_incl_scan(c0, lane), i1 = wave_incl_scan(c1, lane);
      if (live) {
        block_hist[at0] = carry0 + i0 - c0;
        if (e1 != e0) block_hist[at1] = carry1 + i1 - c1;
      }
      carry0 += __builtin_amdgcn_readlane(i0, 63);
      carry1 += __builtin_amdgcn_readlane(i1, 63);
    }
    if (lane == 0) {
      s_tot[e0] = carry0;
      tokens_per_expert[e0] = carry0;
      if (e1 != e0) { s_tot[e1] = carry1; tokens_per_expert[e1] = carry1; }
    }
  }
  __syncthreads();
  // exclusive scan of the totals: thread t owns the contiguous chunk [t*chunk, (t+1)*chunk)
  const int chunk = (experts + 1023) / 1024;
  int local = 0;
  for (int j = 0; j < chunk; ++j) {
    const int e = threadIdx.x * chunk + j;
    if (e < experts) local += s_tot[e];
  }
  int run = block_excl_scan(local, s_wave, nullptr);
  for (int j = 0; j < chunk; ++j) {
    const int e = threadIdx.x * chunk + j;
    if (e < experts) { expert_start[e] = run; run += s_tot[e]; }
  }
}

// Many experts or many blocks: the same per-expert prefix over the blocks, one wave per expert pair but spread over as many
// workgroups as there are pairs / 16 (the single workgroup above walks 32 experts per pass and blocks / 64 steps per expert
// one after the other: 80 us for 256 experts x 256 blocks, 164 us for 1024 experts); the next step's counters are
// requested before this step's scan.  The experts' totals are then scanned by moe_expert_start_kernel.
__global__ __launch_bounds__(1024) void moe_block_prefix_kernel(int32_t* __restrict__ block_hist, int blocks, int experts,
                                                                int32_t* __restrict__ tokens_per_expert) {
  const int lane = threadIdx.x & 63, wave = static_cast<int>(blockIdx.x) * 16 + (threadIdx.x >> 6);
  const int e0 = 2 * wave;
  if (e0 >= experts) return;
  const int e1 = e0 + 1 < experts ? e0 + 1 : e0;
  int carry0 = 0, carry1 = 0;
  auto fetch = [&](int base, int& c0, int& c1) {
    const int blk = base + lane;
    const bool live = blk < blocks;
    c0 = live ? block_hist[static_cast<int64_t>(blk) * experts + e0] : 0;
    c1 = live ? block_hist[static_cast<int64_t>(blk) * experts + e1] : 0;
  };
  int n0, n1;
  fetch(0, n0, n1);
  for (int base = 0; base < blocks; base += 64) {
    const int c0 = n0, c1 = n1;
    if (base + 64 < blocks) fetch(base + 64, n0, n1);
    const int blk = base + lane;
    const int i0 = wave_incl_scan(c0, lane), i1 = wave_incl_scan(c1, lane);
    if (blk < blocks) {
      block_hist[static_cast<int64_t>(blk) * experts + e0] = carry0 + i0 - c0;
      if (e1 != e0) block_hist[static_cast<int64_t>(blk) * experts + e1] = carry1 + i1 - c1;
    }
    carry0 += __builtin_amdgcn_readlane(i0, 63);
    carry1 += __builtin_amdgcn_readlane(i1, 63);
  }
  if (lane == 0) {
    tokens_per_expert[e0] = carry0;
    if (e1 != e0) tokens_per_expert[e1] = carry1;
  }
}

__global__ __launch_bounds__(1024) void moe_expert_start_kernel(const int32_t* __restrict__ tokens_per_expert, int experts,
                                                                int32_t* __restrict__ expert_start) {
  __shared__ int s_wave[16];
  const int chunk = (experts + 1023) / 1024;
  int local = 0;
  for (int j = 0; j < chunk; ++j) {
    const int e = threadIdx.x * chunk + j;
    if (e < experts) local += tokens_per_expert[e];
  }
  int run = block_excl_scan(local, s_wave, nullptr);
  for (int j = 0; j < chunk; ++j) {
    const int e = threadIdx.x * chunk + j;
    if (e < experts) { expert_start[e] = run; run += tokens_per_expert[e]; }
  }
}

__global__ __launch_bounds__(256) void moe_scatter_kernel(const float* __restrict__ gates, const int32_t* __restrict__ ids,
                                                          const int32_t* __restrict__ block_hist,
                                                          const int32_t* __restrict__ expert_start,
                                                          float* __restrict__ sorted_gates, int32_t* __restrict__ token_indices,
                                                          int64_t n, int top_k, int experts) {
  extern __shared__ int s_wave_cnt[];         // [4][experts]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int e = threadIdx.x; e < 4 * experts; e += 256) s_wave_cnt[e] = 0;
  __syncthreads();
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  int e = -1;
  if (i < n) {
    e = ids[i];
    if (e < 0 || e >= experts) e = -1;        // out-of-range ids are dropped (the golden's bincount would raise)
  }
  // rank among the lower lanes of this wave that route to the same expert (stable inside the wave)
  int rank = 0;
  for (int j = 0; j < 64; ++j) {
    const int ej = __builtin_amdgcn_readlane(e, j);
    rank += (ej == e && j < lane) ? 1 : 0;
  }
  if (e >= 0) atomicAdd(&s_wave_cnt[wave * experts + e], 1);
  __syncthreads();
  if (e >= 0) {
    int base = expert_start[e] + block_hist[static_cast<int64_t>(blockIdx.x) * experts + e];
    for (int wv = 0; wv < wave; ++wv) base += s_wave_cnt[wv * experts + e];
    const int pos = base + rank;
    token_indices[pos] = static_cast<int32_t>(i / top_k);
    sorted_gates[pos] = gates[i];
  }
}

// sorted_hidden[p] = hidden[token_indices[p]]: one wave per destination row, 16 B per lane.  (A separate launch: the
// scatter kernel has one block per 256 slots — 64 blocks for Mixtral's 16K slots — far too few to move 2 x 134 MB.)
template <typename T>
__global__ __launch_bounds__(256) void moe_gather_rows_kernel(const T* __restrict__ hidden, const int32_t* __restrict__ token_indices,
                                                              T* __restrict__ sorted_hidden, int64_t n, int hidden_size,
                                                              int64_t tokens) {
  const int lane = threadIdx.x & 63;
  const int64_t p = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (p >= n) return;
  // (rows past the routed total — only possible when ids were dropped — hold an unwritten index: clamp it)
  const int64_t t = min(max(static_cast<int64_t>(token_indices[p]), static_cast<int64_t>(0)), tokens - 1);
  const T* src = hidden + t * hidden_size;
  T* dst = sorted_hidden + p * hidden_size;
  constexpr int VEC = 16 / sizeof(T);
  const bool wide = hidden_size % VEC == 0 && (reinterpret_cast<uintptr_t>(hidden) % 16 == 0) &&
                    (reinterpret_cast<uintptr_t>(sorted_hidden) % 16 == 0);
  if (wide) {
    // four 16-byte loads in flight per lane before the first store (a load -> store loop left one in flight)
    typedef typename vec_of<T, VEC>::type V;
    constexpr int STEP = 64 * VEC;
    int c = lane * VEC;
    for (; c + 3 * STEP < hidden_size; c += 4 * STEP) {
      V v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = load_vec<T, VEC>(src + c + u * STEP);
#pragma unroll
      for (int u = 0; u < 4; ++u) store_vec<T, VEC>(dst + c + u * STEP, v[u]);
    }
    for (; c < hidden_size; c += STEP) store_vec<T, VEC>(dst + c, load_vec<T, VEC>(src + c));
  } else {
    for (int c = lane; c < hidden_size; c += 64) dst[c] = src[c];
  }
}

// ---------------------------------------------------------------------------------------------------------
// combine: out[t] = sum over the rows routed from token t, in ascending row order (= the golden's scatter order),
// fp32 from zero, product and sum rounded separately (no FMA) -> bit-identical to the golden
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void moe_zero_kernel(int32_t* __restrict__ p, int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) p[i] = 0;
}

__global__ __launch_bounds__(256) void moe_count_kernel(const int32_t* __restrict__ tok, int64_t n, int64_t tokens,
                                                        int32_t* __restrict__ cnt) {
  const int64_t j = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (j < n) {
    const int t = tok[j];
    if (t >= 0 && t < tokens) atomicAdd(&cnt[t], 1);
  }
}

// exclusive scan of cnt[0..tokens) into start[0..tokens], one block of 1024 threads
__global__ __launch_bounds__(1024) void moe_token_scan_kernel(const int32_t* __restrict__ cnt, int64_t tokens,
                                                              int32_t* __restrict__ start) {
  __shared__ int s_wave[16];
  const int64_t chunk = (tokens + 1023) / 1024;
  const int64_t lo = threadIdx.x * chunk, hi = min(lo + chunk, tokens);
  int local = 0;
  for (int64_t t = lo; t < hi; ++t) local += cnt[t];
  int run = block_excl_scan(local, s_wave, nullptr);
  for (int64_t t = lo; t < hi; ++t) { start[t] = run; run += cnt[t]; }
  if (threadIdx.x == 1023) start[tokens] = run;            // the last thread's chunk ends the array (empty chunks carry the total)
}

// The whole row-list plan of the combine in ONE single-block launch when the per-token counters fit LDS: count (LDS
// atomics) -> exclusive scan -> fill (LDS cursors).  Replaces zero + count + scan + zero + fill (five launches, ~35 us
// for 8192 tokens — more than the 30 us the combine itself takes; ~6 us now).  The order inside a token's list is the
// order of the atomics, i.e. undefined, exactly as before: the combine kernel sorts each list by row id.
constexpr int PLAN_MAX_TOKENS = 15 * 1024;
__global__ __launch_bounds__(1024) void moe_token_plan_kernel(const int32_t* __restrict__ tok, int64_t n, int tokens,
                                                              int32_t* __restrict__ start, int32_t* __restrict__ list) {
  extern __shared__ int s_cnt[];                            // [tokens]: counters, then cursors
  __shared__ int s_wave[16];
  for (int t = threadIdx.x; t < tokens; t += 1024) s_cnt[t] = 0;
  __syncthreads();
  // (token ids are fetched eight per thread at a time: a load -> atomic loop waits out one memory round trip per row)
  for (int64_t j0 = threadIdx.x; j0 < n; j0 += 8 * 1024) {
    int t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = j0 + u * 1024 < n ? tok[j0 + u * 1024] : -1;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (t[u] >= 0 && t[u] < tokens) atomicAdd(&s_cnt[t[u]], 1);
  }
  __syncthreads();
  const int chunk = (tokens + 1023) / 1024;
  const int lo = min(static_cast<int>(threadIdx.x) * chunk, tokens), hi = min(lo + chunk, tokens);
  int local = 0;
  for (int t = lo; t < hi; ++t) local += s_cnt[t];
  int run = block_excl_scan(local, s_wave, nullptr);
  for (int t = lo; t < hi; ++t) {
    const int c = s_cnt[t];
    s_cnt[t] = run;
    start[t] = run;
    run += c;
  }
  if (threadIdx.x == 1023) start[tokens] = run;
  __syncthreads();
  for (int64_t j0 = threadIdx.x; j0 < n; j0 += 8 * 1024) {
    int t[8], at[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = j0 + u * 1024 < n ? tok[j0 + u * 1024] : -1;
#pragma unroll
    for (int u = 0; u < 8; ++u) at[u] = (t[u] >= 0 && t[u] < tokens) ? atomicAdd(&s_cnt[t[u]], 1) : -1;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (at[u] >= 0) list[at[u]] = static_cast<int32_t>(j0 + u * 1024);
  }
}

__global__ __launch_bounds__(256) void moe_fill_kernel(const int32_t* __restrict__ tok, int64_t n, int64_t tokens,
                                                       const int32_t* __restrict__ start, int32_t* __restrict__ cursor,
                                                       int32_t* __restrict__ list) {
  const int64_t j = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (j < n) {
    const int t = tok[j];
    if (t >= 0 && t < tokens) list[start[t] + atomicAdd(&cursor[t], 1)] = static_cast<int32_t>(j);   // sorted below
  }
}

constexpr int COMB_MAX_LIST = 1024;

template <typename T>
__global__ __launch_bounds__(256) void moe_combine_kernel(const T* __restrict__ rows, const float* __restrict__ gates,
                                                          const int32_t* __restrict__ start, const int32_t* __restrict__ list,
                                                          T* __restrict__ out, int hidden_size) {
  __shared__ int s_list[COMB_MAX_LIST];
  const int64_t t = blockIdx.x;
  const int s0 = start[t];
  int n = start[t + 1] - s0;
  const bool sortable = n <= COMB_MAX_LIST;
  if (sortable) {
    // rank sort by row id (ids are distinct): s_list[rank] = id
    for (int a = threadIdx.x; a < n; a += 256) {
      const int id = list[s0 + a];
      int rank = 0;
      for (int b = 0; b < n; ++b) rank += list[s0 + b] < id ? 1 : 0;
      s_list[rank] = id;
    }
    __syncthreads();
  }
  constexpr int VEC = 16 / sizeof(T);
  typedef typename vec_of<T, VEC>::type V;
  const bool wide = hidden_size % VEC == 0 && (reinterpret_cast<uintptr_t>(rows) % 16 == 0) && (reinterpret_cast<uintptr_t>(out) % 16 == 0);
  T* dst = out + t * hidden_size;
  if (wide) {
    // two column chunks x two rows per step: four 16-byte loads in flight per lane; the sums keep the list order
    constexpr int STEP = 256 * VEC;
    for (int c = threadIdx.x * VEC; c < hidden_size; c += 2 * STEP) {
      const bool two = c + STEP < hidden_size;
      const int c1 = two ? c + STEP : c;
      float acc[2][VEC];
#pragma unroll
      for (int q = 0; q < VEC; ++q) acc[0][q] = acc[1][q] = 0.f;
      auto add = [&](const V& r, float g, float (&dst_acc)[VEC]) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
          const float f = elt<T>::to_f(vget<T, VEC>(r, q));
          dst_acc[q] = __fadd_rn(dst_acc[q], gates ? __fmul_rn(f, g) : f);
        }
      };
      int a = 0;
      for (; a + 1 < n; a += 2) {
        const int id0 = sortable ? s_list[a] : list[s0 + a];
        const int id1 = sortable ? s_list[a + 1] : list[s0 + a + 1];
        const T* p0 = rows + static_cast<int64_t>(id0) * hidden_size;
        const T* p1 = rows + static_cast<int64_t>(id1) * hidden_size;
        const V r00 = load_vec<T, VEC>(p0 + c), r01 = load_vec<T, VEC>(p0 + c1);
        const V r10 = load_vec<T, VEC>(p1 + c), r11 = load_vec<T, VEC>(p1 + c1);
        const float g0 = gates ? gates[id0] : 1.f, g1 = gates ? gates[id1] : 1.f;
        add(r00, g0, acc[0]); add(r01, g0, acc[1]);
        add(r10, g1, acc[0]); add(r11, g1, acc[1]);
      }
      if (a < n) {
        const int id0 = sortable ? s_list[a] : list[s0 + a];
        const T* p0 = rows + static_cast<int64_t>(id0) * hidden_size;
        const V r00 = load_vec<T, VEC>(p0 + c), r01 = load_vec<T, VEC>(p0 + c1);
        const float g0 = gates ? gates[id0] : 1.f;
        add(r00, g0, acc[0]); add(r01, g0, acc[1]);
      }
      V o;
#pragma unroll
      for (int q = 0; q < VEC; ++q) vset<T, VEC>(o, q, elt<T>::from_f(acc[0][q]));
      store_vec<T, VEC>(dst + c, o);
      if (two) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) vset<T, VEC>(o, q, elt<T>::from_f(acc[1][q]));
        store_vec<T, VEC>(dst + c1, o);
      }
    }
  } else {
    for (int c = threadIdx.x; c < hidden_size; c += 256) {
      float acc = 0.f;
      for (int a = 0; a < n; ++a) {
        const int id = sortable ? s_list[a] : list[s0 + a];
        const float f = elt<T>::to_f(rows[static_cast<int64_t>(id) * hidden_size + c]);
        acc = __fadd_rn(acc, gates ? __fmul_rn(f, gates[id]) : f);
      }
      dst[c] = elt<T>::from_f(acc);
    }
  }
}

static int pow2ceil_log2(int v) {
  int lg = 0;
  while ((1 << lg) < v) ++lg;
  return lg;
}

template <typename T>
static int launch_gating(const void* x, const float* w, int32_t* idx, float* gate, int64_t tokens, int hidden, int experts,
                         int top_k, hipStream_t s) {
  if constexpr (sizeof(T) == 2) {
    const long long route = MOJO_SWITCH("MOJO_HIP_GATING", 0);          // 1 = the general streaming kernel (see gate_route)
    if (experts == 8 && hidden % 512 == 0 && tokens >= 32 && aligned_to(w, 16) && route != 1) {
      hipLaunchKernelGGL((moe_gating_e8_kernel<T>), dim3(static_cast<unsigned>(ceil_div(tokens, static_cast<int64_t>(8)))), dim3(256), 0, s,
                         static_cast<const T*>(x), w, idx, gate, tokens, hidden, top_k);
      MOJO_CHECK_LAUNCH("moe_gating(e8)");
      note_launch("moe_gating:e8");
      return MOJO_OK;
    }
  }
  const int ep_log2 = pow2ceil_log2(experts < 64 ? experts : 64);
  const int ep = 1 << ep_log2, subs = 64 / ep;
  const int ei = (experts + ep - 1) / ep;                       // experts per lane
  // tokens per block = subs * TT: take the largest TT that still leaves >= 512 blocks (2 per CU), else TT = 1
  int tt = 4;
  while (tt > 1 && ceil_div(tokens, static_cast<int64_t>(subs) * tt) < 512) tt >>= 1;
  if (ei > 4 && tt > 2) tt = 2;                                 // register budget: TT * EI accumulators
  const int tpw = subs * tt;
  const int64_t blocks = ceil_div(tokens, tpw);
  const size_t lds = static_cast<size_t>(4) * tpw * experts * sizeof(float);
  MOJO_REQUIRE(lds <= 64 * 1024, MOJO_EUNSUPPORTED, "moe_gating: %d experts need %zu B of LDS", experts, lds);
  const T* xp = static_cast<const T*>(x);
#define GATE_LAUNCH(TT_, EI_)                                                                                          \
  hipLaunchKernelGGL((moe_gating_kernel<T, TT_, EI_>), dim3(static_cast<unsigned>(blocks)), dim3(256), lds, s, xp, w, idx, \
                     gate, tokens, hidden, experts, top_k, ep_log2)
#define GATE_EI(TT_)                                                                                                   \
  do {                                                                                                                 \
    if (ei <= 1) GATE_LAUNCH(TT_, 1); else if (ei <= 2) GATE_LAUNCH(TT_, 2); else if (ei <= 4) GATE_LAUNCH(TT_, 4);      \
    else if (ei <= 8) GATE_LAUNCH(TT_, 8); else GATE_LAUNCH(TT_, 16);                                                   \
  } while (0)
  if (tt == 4) GATE_EI(4); else if (tt == 2) GATE_EI(2); else GATE_EI(1);
#undef GATE_EI
#undef GATE_LAUNCH
  MOJO_CHECK_LAUNCH("moe_gating");
  note_launch("moe_gating:general");
  return MOJO_OK;
}

}  // namespace mojo

using namespace mojo;

// MOJO_HIP_GATING forces the router's kernel (0 / unset = by shape): 1 = the general streaming kernel (no E = 8
// specialisation, no small-batch kernel, no matrix cores), 2 = streaming kernels only (E = 8 specialisation allowed),
// 3 = the small-batch kernel wherever it applies, 4 = the matrix-core route wherever it applies.
// The MFMA route pays when the logits are real GEMM work: many experts, many tokens, 16-bit activations.
static bool gate_use_mfma(int64_t tokens, int64_t hidden, int64_t experts, int dtype) {
  if (const long long e = MOJO_SWITCH("MOJO_HIP_GATING", 0); e != 0) return e == 4 && (dtype == MOJO_BF16 || dtype == MOJO_F16) && hidden % 64 == 0 && hidden >= 64;
  return (dtype == MOJO_BF16 || dtype == MOJO_F16) && experts >= 32 && tokens >= 512 && hidden % 64 == 0 && hidden >= 64;
}
static int64_t gate_e_pad(int64_t experts) { return (experts + 15) / 16 * 16; }
static int gate_splitk(int64_t tokens, int64_t hidden, int64_t e_pad) {
  const int64_t tiles = ceil_div(tokens, 256) * ceil_div(e_pad, 256);
  int64_t sk = ceil_div(256, tiles);
  const int64_t nkt = hidden / 64;
  if (sk > nkt / 4) sk = nkt / 4;
  if (sk > 8) sk = 8;
  return sk < 1 ? 1 : static_cast<int>(sk);
}

// Fewer than 512 tokens, and more than the eight experts or fewer than the 32 tokens the streaming kernel covers:
// moe_gating_small_kernel (any dtype).
// MOJO_HIP_GATING=3 forces it, 1 / 2 / 4 exclude it.
static bool gate_use_small(int64_t tokens, int64_t hidden, int64_t experts) {
  if (hidden >= (1LL << 30) || tokens >= (1LL << 24)) return false;
  if (const long long e = MOJO_SWITCH("MOJO_HIP_GATING", 0); e != 0) return e == 3;
  return tokens < 512 && (experts > 8 || tokens < 32);                   // (from 512 tokens and 32 experts on the matrix-core route takes over)
}

extern "C" int64_t mojo_hip_moe_gating_workspace_bytes(int64_t tokens, int64_t hidden_size, int64_t num_experts, int dtype) {
  if (gate_use_small(tokens, hidden_size, num_experts))
    return static_cast<int64_t>(gate_small_plan(tokens, hidden_size, num_experts).slices) * tokens * gate_e_pad(num_experts) * 4 + 256;
  if (!gate_use_mfma(tokens, hidden_size, num_experts, dtype)) return 64;
  const int64_t e_pad = gate_e_pad(num_experts);
  const int sk = gate_splitk(tokens, hidden_size, e_pad);
  return 2 * hidden_size * e_pad * 2 + 2 * sk * tokens * e_pad * 4 + 256;
}

template <typename T>
static int gating_mfma(const void* x, const float* w, int32_t* idx, float* gate, int64_t tokens, int hidden, int experts,
                       int top_k, int dtype, void* workspace, hipStream_t s) {
  const int e_pad = static_cast<int>(gate_e_pad(experts));
  const int sk = gate_splitk(tokens, hidden, e_pad);
  char* ws = static_cast<char*>(workspace);
  T* hi = reinterpret_cast<T*>(ws);
  T* lo = hi + static_cast<int64_t>(hidden) * e_pad;
  const int64_t w_bytes = (2 * static_cast<int64_t>(hidden) * e_pad * 2 + 255) / 256 * 256;
  float* logits = reinterpret_cast<float*>(ws + w_bytes);                   // [2 * sk][tokens][e_pad]
  const int64_t slab = tokens * e_pad;
  int64_t blocks = ceil_div(static_cast<int64_t>(hidden) * e_pad, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(moe_gate_split_kernel<T>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, w, hi, lo, hidden, experts, e_pad);
  MOJO_CHECK_LAUNCH("moe_gating(split)");
  // ONE product over K' = 2 * hidden: the activations against [w_hi; w_lo] stacked along K (hi and lo are adjacent in the
  // workspace), A's K index wrapping at `hidden` (GemmArgs::a_k_wrap).  With an even slice count no K slice straddles the
  // wrap.  (Round 4: the two products used to be two launches with sk slabs each — 2 x 67 MB of fp32 slabs written and read
  // again at 8192 tokens x 256 experts, more than the 30 GFLOP of either pass cost; one launch writes half of them and pays
  // the per-tile prologue / epilogue once.)  The slices still sum hi parts first, then lo parts, in index order.
  const int sk2 = sk < 2 ? 2 : ((sk + 1) & ~1);
  {
    GemmArgs a;
    a.A = x; a.W = hi; a.bias = nullptr;
    a.C = logits;
    a.lda = hidden; a.ldc = e_pad; a.w_group = 0; a.w_k = e_pad; a.w_n = 1;          // [K, N] weights
    a.K = 2 * hidden; a.N = e_pad; a.G = 1;
    a.a_k_wrap = hidden / 64;
    a.uniform_rows = static_cast<int>(tokens);
    a.splitk = sk2; a.slab = a.C; a.slab_rows = static_cast<int>(tokens);
    const int rc = launch_gemm_mfma256_f32out(a, dtype, 0, tokens, s);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(moe_gate_select_kernel, dim3(static_cast<unsigned>(ceil_div(tokens, 4))), dim3(256), 0, s, logits, slab,
                     sk2, e_pad, idx, gate, tokens, experts, top_k);
  MOJO_CHECK_LAUNCH("moe_gating(select)");
  return MOJO_OK;
}

extern "C" int mojo_hip_moe_gating(const void* hidden, const float* gate_weight, int32_t* top_k_indices,
                                   float* top_k_gates, int64_t tokens, int64_t hidden_size, int64_t num_experts,
                                   int64_t top_k, int dtype, void* workspace, int64_t workspace_bytes,
                                   mojo_stream_t stream) {
  if (tokens == 0) return MOJO_OK;
  MOJO_REQUIRE(hidden && gate_weight && top_k_indices && top_k_gates, MOJO_EINVAL, "moe_gating: null pointer");
  MOJO_REQUIRE(tokens > 0 && hidden_size > 0 && hidden_size < (1LL << 30), MOJO_EINVAL, "moe_gating: bad shape");
  MOJO_REQUIRE(num_experts >= 1 && num_experts <= GATE_MAX_E, MOJO_EUNSUPPORTED, "moe_gating: %lld experts (1..%d)",
               (long long)num_experts, GATE_MAX_E);
  MOJO_REQUIRE(top_k >= 1 && top_k <= num_experts && top_k <= 64, MOJO_EUNSUPPORTED, "moe_gating: top_k %lld (1..min(E,64))",
               (long long)top_k);
  MOJO_REQUIRE(aligned_to(hidden, 16), MOJO_EUNSUPPORTED, "moe_gating: hidden_states must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int h = static_cast<int>(hidden_size), e = static_cast<int>(num_experts), k = static_cast<int>(top_k);
  if (gate_use_small(tokens, hidden_size, num_experts)) {
    MOJO_REQUIRE(dtype == MOJO_F32 || dtype == MOJO_F16 || dtype == MOJO_BF16, MOJO_EUNSUPPORTED, "moe_gating: dtype %d not supported", dtype);
    MOJO_REQUIRE(workspace && workspace_bytes >= mojo_hip_moe_gating_workspace_bytes(tokens, hidden_size, num_experts, dtype) &&
                     aligned_to(workspace, 16),
                 MOJO_EWORKSPACE, "moe_gating: workspace too small");
    const GateSmallPlan p = gate_small_plan(tokens, hidden_size, num_experts);
    const int e_pad = static_cast<int>(gate_e_pad(num_experts)), t = static_cast<int>(tokens);
    float* slabs = static_cast<float*>(workspace);
    const dim3 grid(static_cast<unsigned>(p.slices), static_cast<unsigned>(p.t_chunks), static_cast<unsigned>(p.e_chunks));
#define GATE_SMALL(TY) hipLaunchKernelGGL(moe_gating_small_kernel<TY>, grid, dim3(256), 0, s, static_cast<const TY*>(hidden), gate_weight, slabs, t, h, e, e_pad, p.ep_log2, p.slice_len)
    if (dtype == MOJO_F32) GATE_SMALL(float); else if (dtype == MOJO_F16) GATE_SMALL(f16_t); else GATE_SMALL(bf16_t);
#undef GATE_SMALL
    MOJO_CHECK_LAUNCH("moe_gating(small)");
    hipLaunchKernelGGL(moe_gate_reduce_select_kernel, dim3(static_cast<unsigned>(tokens)), dim3(256), static_cast<size_t>(4) * e_pad * sizeof(float), s,
                       slabs, tokens * e_pad, p.slices, e_pad, top_k_indices, top_k_gates, e, k);
    MOJO_CHECK_LAUNCH("moe_gating(select)");
    note_launch("moe_gating:small");
    return MOJO_OK;
  }
  if (gate_use_mfma(tokens, hidden_size, num_experts, dtype)) {
    MOJO_REQUIRE(workspace && workspace_bytes >= mojo_hip_moe_gating_workspace_bytes(tokens, hidden_size, num_experts, dtype) &&
                     aligned_to(workspace, 256),
                 MOJO_EWORKSPACE, "moe_gating: workspace too small");
    MOJO_REQUIRE(tokens < (1LL << 31), MOJO_EUNSUPPORTED, "moe_gating: too many tokens");
    const int rc = dtype == MOJO_BF16 ? gating_mfma<bf16_t>(hidden, gate_weight, top_k_indices, top_k_gates, tokens, h, e, k, dtype, workspace, s)
                                      : gating_mfma<f16_t>(hidden, gate_weight, top_k_indices, top_k_gates, tokens, h, e, k, dtype, workspace, s);
    if (rc == MOJO_OK) note_launch("moe_gating:mfma");
    return rc;
  }
  switch (dtype) {
    case MOJO_F32: return launch_gating<float>(hidden, gate_weight, top_k_indices, top_k_gates, tokens, h, e, k, s);
    case MOJO_F16: return launch_gating<f16_t>(hidden, gate_weight, top_k_indices, top_k_gates, tokens, h, e, k, s);
    case MOJO_BF16: return launch_gating<bf16_t>(hidden, gate_weight, top_k_indices, top_k_gates, tokens, h, e, k, s);
    default: MOJO_REQUIRE(false, MOJO_EUNSUPPORTED, "moe_gating: dtype %d not supported", dtype);
  }
}

extern "C" int64_t mojo_hip_moe_dispatch_workspace_bytes(int64_t slots, int64_t num_experts) {
  const int64_t blocks = ceil_div(slots > 0 ? slots : 1, 256);
  return (blocks * num_experts + num_experts) * static_cast<int64_t>(sizeof(int32_t)) + 64;
}

extern "C" int mojo_hip_moe_dispatch(const void* hidden, const float* top_k_gates, const int32_t* top_k_indices,
                                     void* sorted_hidden, int32_t* tokens_per_expert, float* sorted_gates,
                                     int32_t* token_indices, int64_t tokens, int64_t hidden_size, int64_t top_k,
                                     int64_t num_experts, int dtype, void* workspace, int64_t workspace_bytes,
                                     mojo_stream_t stream) {
  MOJO_REQUIRE(tokens >= 0 && hidden_size > 0 && top_k >= 1 && num_experts >= 1, MOJO_EINVAL, "moe_dispatch: bad shape");
  MOJO_REQUIRE(num_experts <= DISP_MAX_E, MOJO_EUNSUPPORTED, "moe_dispatch: %lld experts (<= %d)", (long long)num_experts, DISP_MAX_E);
  MOJO_REQUIRE(tokens_per_expert, MOJO_EINVAL, "moe_dispatch: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t n = tokens * top_k;
  if (n == 0) {
    if (hipMemsetAsync(tokens_per_expert, 0, static_cast<size_t>(num_experts) * sizeof(int32_t), s) != hipSuccess) {
      set_error("moe_dispatch: memset failed");
      return MOJO_ELAUNCH;
    }
    return MOJO_OK;
  }
  MOJO_REQUIRE(hidden && top_k_gates && top_k_indices && sorted_hidden && sorted_gates && token_indices, MOJO_EINVAL,
               "moe_dispatch: null pointer");
  MOJO_REQUIRE(n < (1LL << 31) && hidden_size < (1LL << 31), MOJO_EUNSUPPORTED, "moe_dispatch: too many routing slots");
  MOJO_REQUIRE(dtype == MOJO_F32 || dtype == MOJO_F16 || dtype == MOJO_BF16, MOJO_EUNSUPPORTED, "moe_dispatch: dtype %d", dtype);
  MOJO_REQUIRE(workspace && workspace_bytes >= mojo_hip_moe_dispatch_workspace_bytes(n, num_experts) && aligned_to(workspace, 4),
               MOJO_EWORKSPACE, "moe_dispatch: workspace too small");
  const int e = static_cast<int>(num_experts);
  const int blocks = static_cast<int>(ceil_div(n, 256));
  int32_t* block_hist = static_cast<int32_t*>(workspace);
  int32_t* expert_start = block_hist + static_cast<int64_t>(blocks) * e;
  hipLaunchKernelGGL(moe_hist_kernel, dim3(blocks), dim3(256), e * sizeof(int), s, top_k_indices, n, e, block_hist);
  MOJO_CHECK_LAUNCH("moe_dispatch(hist)");
  if (static_cast<int64_t>(e) * blocks <= 64 * 64) {     // few counters: one workgroup does the block prefix and the experts' scan
    hipLaunchKernelGGL(moe_scan_kernel, dim3(1), dim3(1024), 0, s, block_hist, blocks, e, tokens_per_expert, expert_start);
    MOJO_CHECK_LAUNCH("moe_dispatch(scan)");
  } else {
    hipLaunchKernelGGL(moe_block_prefix_kernel, dim3(static_cast<unsigned>(ceil_div((e + 1) / 2, 16))), dim3(1024), 0, s, block_hist, blocks, e, tokens_per_expert);
    MOJO_CHECK_LAUNCH("moe_dispatch(block prefix)");
    hipLaunchKernelGGL(moe_expert_start_kernel, dim3(1), dim3(1024), 0, s, tokens_per_expert, e, expert_start);
    MOJO_CHECK_LAUNCH("moe_dispatch(expert start)");
  }
  const size_t lds = static_cast<size_t>(4) * e * sizeof(int);
  const int k = static_cast<int>(top_k), h = static_cast<int>(hidden_size);
  hipLaunchKernelGGL(moe_scatter_kernel, dim3(blocks), dim3(256), lds, s, top_k_gates, top_k_indices, block_hist, expert_start,
                     sorted_gates, token_indices, n, k, e);
  MOJO_CHECK_LAUNCH("moe_dispatch(scatter)");
  const unsigned row_blocks = static_cast<unsigned>(ceil_div(n, 4));
#define GATHER(TY)                                                                                                     \
  hipLaunchKernelGGL(moe_gather_rows_kernel<TY>, dim3(row_blocks), dim3(256), 0, s, static_cast<const TY*>(hidden),      \
                     token_indices, static_cast<TY*>(sorted_hidden), n, h, tokens)
  if (dtype == MOJO_F32) GATHER(float); else if (dtype == MOJO_F16) GATHER(f16_t); else GATHER(bf16_t);
#undef GATHER
  MOJO_CHECK_LAUNCH("moe_dispatch(gather)");
  return MOJO_OK;
}

extern "C" int64_t mojo_hip_moe_combine_workspace_bytes(int64_t tokens, int64_t rows) {
  return (2 * tokens + 1 + rows) * static_cast<int64_t>(sizeof(int32_t)) + 64;
}

extern "C" int mojo_hip_moe_combine(const void* expert_outputs, const float* sorted_gates, const int32_t* token_indices,
                                    void* out, int64_t tokens, int64_t rows, int64_t hidden_size, int dtype,
                                    void* workspace, int64_t workspace_bytes, mojo_stream_t stream) {
  MOJO_REQUIRE(tokens >= 0 && rows >= 0 && hidden_size > 0 && hidden_size < (1LL << 31), MOJO_EINVAL, "moe_combine: bad shape");
  if (tokens == 0) return MOJO_OK;
  MOJO_REQUIRE(out && (rows == 0 || (expert_outputs && token_indices)), MOJO_EINVAL, "moe_combine: null pointer");
  MOJO_REQUIRE(tokens < (1LL << 31) && rows < (1LL << 31), MOJO_EUNSUPPORTED, "moe_combine: too many rows");
  MOJO_REQUIRE(dtype == MOJO_F32 || dtype == MOJO_F16 || dtype == MOJO_BF16, MOJO_EUNSUPPORTED, "moe_combine: dtype %d", dtype);
  MOJO_REQUIRE(workspace && workspace_bytes >= mojo_hip_moe_combine_workspace_bytes(tokens, rows) && aligned_to(workspace, 4),
               MOJO_EWORKSPACE, "moe_combine: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  int32_t* cnt = static_cast<int32_t*>(workspace);      // [tokens]   (reused as the fill cursor)
  int32_t* start = cnt + tokens;                        // [tokens + 1]
  int32_t* list = start + tokens + 1;                   // [rows]
  // (zeroed by a kernel, not hipMemsetAsync: a captured graph holding two memset nodes on this buffer between kernel
  // nodes aborted at replay on ROCm 7.0)
  const bool no_plan = MOJO_SWITCH("MOJO_HIP_MOE_PLAN", 1) == 0;   // 0: the five-launch plan (the route of > 15 360 tokens)
  if (tokens <= PLAN_MAX_TOKENS && !no_plan) {
    hipLaunchKernelGGL(moe_token_plan_kernel, dim3(1), dim3(1024), static_cast<size_t>(tokens) * sizeof(int), s, token_indices, rows,
                       static_cast<int>(tokens), start, list);
    MOJO_CHECK_LAUNCH("moe_combine(plan)");
  } else {
    const unsigned zero_blocks = static_cast<unsigned>(ceil_div(tokens, 256) > 1024 ? 1024 : ceil_div(tokens, 256));
    hipLaunchKernelGGL(moe_zero_kernel, dim3(zero_blocks), dim3(256), 0, s, cnt, tokens);
    MOJO_CHECK_LAUNCH("moe_combine(zero)");
    const int row_blocks = static_cast<int>(ceil_div(rows > 0 ? rows : 1, 256));
    if (rows > 0) {
      hipLaunchKernelGGL(moe_count_kernel, dim3(row_blocks), dim3(256), 0, s, token_indices, rows, tokens, cnt);
      MOJO_CHECK_LAUNCH("moe_combine(count)");
    }
    hipLaunchKernelGGL(moe_token_scan_kernel, dim3(1), dim3(1024), 0, s, cnt, tokens, start);
    MOJO_CHECK_LAUNCH("moe_combine(scan)");
    if (rows > 0) {
      hipLaunchKernelGGL(moe_zero_kernel, dim3(zero_blocks), dim3(256), 0, s, cnt, tokens);
      MOJO_CHECK_LAUNCH("moe_combine(zero)");
      hipLaunchKernelGGL(moe_fill_kernel, dim3(row_blocks), dim3(256), 0, s, token_indices, rows, tokens, start, cnt, list);
      MOJO_CHECK_LAUNCH("moe_combine(fill)");
    }
  }
  const int h = static_cast<int>(hidden_size);
#define COMBINE(TY)                                                                                                    \
  hipLaunchKernelGGL(moe_combine_kernel<TY>, dim3(static_cast<unsigned>(tokens)), dim3(256), 0, s,                       \
                     static_cast<const TY*>(expert_outputs), sorted_gates, start, list, static_cast<TY*>(out), h)
  if (dtype == MOJO_F32) COMBINE(float); else if (dtype == MOJO_F16) COMBINE(f16_t); else COMBINE(bf16_t);
#undef COMBINE
  MOJO_CHECK_LAUNCH("moe_combine");
  return MOJO_OK;
}
