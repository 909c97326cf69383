// MojoApplyRoPE and MojoRotaryEmbedding.
//
// apply_rope: rotate-half on the trailing `rope_dim` features of every head, the leading
// `head_dim - rope_dim` features pass through.  q and k are handled by ONE launch and are read in the
// caller's layout through strides (the reference's Triton path first copies to a contiguous BSND
// buffer and transposes back — two extra HBM passes, backends/ttx/kernels/ilu/rope.py:24-71,:533-537).
// Arithmetic mirrors the golden bit for bit: a = x*cos, b = rot(x)*sin in fp32 (separately rounded,
// no FMA contraction), a+b, ONE rounding to the storage type.
//
// Algorithmic bytes per token: (Nq+Nk) x D x elt read + the same written + 2 x rope_dim x 4 (cos/sin).
#include "common.h"

namespace mojo {

struct RopeArgs {
  const void* src[2];
  void* dst[2];
  int64_t s_b[2], s_t[2], s_n[2];     // source strides (elements)
  int64_t d_b[2], d_t[2], d_n[2];     // destination strides
  int heads[2];
  const float* cos;
  const float* sin;
  int64_t cos_b, cos_t;
  int64_t batch, tokens;
  int head_dim, rope_dim;
};

// A "row" is one (b, t, head) vector of head_dim elements.  Work items per row: nope/VEC pass-through
// vectors + (rope_dim/2)/VEC rotation pairs.  TPRW threads (a power of two >= items) serve one row.
// CV: the cos/sin rows may be read with 16-byte loads (VEC % 4 == 0, rows 16-byte aligned): 8 float4 loads per work item
// instead of 4*VEC scalar ones.
template <typename T, int VEC, bool CV>
__global__ __launch_bounds__(256) void apply_rope_kernel(RopeArgs a, int items_nope, int items_rot, int tprw_log2) {
  typedef typename vec_of<T, VEC>::type V;
  const int tprw = 1 << tprw_log2;
  const int rows_per_block = 256 >> tprw_log2;
  const int item = threadIdx.x & (tprw - 1);
  const int64_t heads_total = a.heads[0] + a.heads[1];
  const int64_t n_rows = a.batch * a.tokens * heads_total;
  const int half = a.rope_dim / 2;
  const int nope = a.head_dim - a.rope_dim;
  for (int64_t row = static_cast<int64_t>(blockIdx.x) * rows_per_block + (threadIdx.x >> tprw_log2); row < n_rows;
       row += static_cast<int64_t>(gridDim.x) * rows_per_block) {
    if (item >= items_nope + items_rot) continue;
    const int64_t bt = row / heads_total;
    int hh = static_cast<int>(row - bt * heads_total);
    const int which = hh >= a.heads[0];
    if (which) hh -= a.heads[0];
    const int64_t b = bt / a.tokens, t = bt - b * a.tokens;
    const T* src = static_cast<const T*>(a.src[which]) + b * a.s_b[which] + t * a.s_t[which] + hh * a.s_n[which];
    T* dst = static_cast<T*>(a.dst[which]) + b * a.d_b[which] + t * a.d_t[which] + hh * a.d_n[which];
    if (item < items_nope) {
      store_vec<T, VEC>(dst + item * VEC, load_vec<T, VEC>(src + item * VEC));
      continue;
    }
    const int i0 = (item - items_nope) * VEC;                 // offset inside the first half
    const float* c = a.cos + b * a.cos_b + t * a.cos_t;
    const float* s = a.sin + b * a.cos_b + t * a.cos_t;
    const V x1 = load_vec<T, VEC>(src + nope + i0);
    const V x2 = load_vec<T, VEC>(src + nope + half + i0);
    float c1[VEC], c2[VEC], s1[VEC], s2[VEC];
    if constexpr (CV) {
#pragma unroll
      for (int q = 0; q < VEC / 4; ++q) {
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(c + i0 + 4 * q), a2 = *reinterpret_cast<const f32x4*>(c + half + i0 + 4 * q);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(s + i0 + 4 * q), b2 = *reinterpret_cast<const f32x4*>(s + half + i0 + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) { c1[4 * q + e] = a1[e]; c2[4 * q + e] = a2[e]; s1[4 * q + e] = b1[e]; s2[4 * q + e] = b2[e]; }
      }
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) { c1[j] = c[i0 + j]; c2[j] = c[half + i0 + j]; s1[j] = s[i0 + j]; s2[j] = s[half + i0 + j]; }
    }
    V o1, o2;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float f1 = elt<T>::to_f(vget<T, VEC>(x1, j));
      const float f2 = elt<T>::to_f(vget<T, VEC>(x2, j));
      // out[i]      = x1*cos[i]      + (-x2)*sin[i]
      // out[half+i] = x2*cos[half+i] + ( x1)*sin[half+i]
      const float r1 = __fadd_rn(__fmul_rn(f1, c1[j]), __fmul_rn(-f2, s1[j]));
      const float r2 = __fadd_rn(__fmul_rn(f2, c2[j]), __fmul_rn(f1, s2[j]));
      vset<T, VEC>(o1, j, elt<T>::from_f(r1));
      vset<T, VEC>(o2, j, elt<T>::from_f(r2));
    }
    store_vec<T, VEC>(dst + nope + i0, o1);
    store_vec<T, VEC>(dst + nope + half + i0, o2);
  }
}

template <typename T>
static int dispatch_rope(const RopeArgs& a, hipStream_t s) {
  const int half = a.rope_dim / 2;
  const int nope = a.head_dim - a.rope_dim;
  auto ok = [&](int vec) {
    if (half % vec || nope % vec) return false;
    const size_t al = vec * sizeof(T);
    for (int w = 0; w < 2; ++w) {
      if (!aligned_to(a.src[w], al) || !aligned_to(a.dst[w], al)) return false;
      for (int64_t st : {a.s_b[w], a.s_t[w], a.s_n[w], a.d_b[w], a.d_t[w], a.d_n[w]})
        if (st % vec) return false;
    }
    return true;
  };
  int vec = 1;
  for (int v : {static_cast<int>(16 / sizeof(T)), static_cast<int>(8 / sizeof(T)), 2})
    if (v > 1 && ok(v)) { vec = v; break; }
  const int items_nope = nope / vec, items_rot = half / vec;
  int lg = 0;
  while ((1 << lg) < items_nope + items_rot) ++lg;
  MOJO_REQUIRE(lg <= 8, MOJO_EUNSUPPORTED, "apply_rope: head_dim %d too large for this kernel", a.head_dim);
  const int64_t n_rows = a.batch * a.tokens * (a.heads[0] + a.heads[1]);
  int64_t blocks = ceil_div(n_rows, 256 >> lg);
  if (blocks > 256 * 32) blocks = 256 * 32;
  const bool cos_vec = vec % 4 == 0 && half % 4 == 0 && aligned_to(a.cos, 16) && aligned_to(a.sin, 16) && a.cos_b % 4 == 0 && a.cos_t % 4 == 0;
#define LAUNCH(V, CV) hipLaunchKernelGGL((apply_rope_kernel<T, V, CV>), dim3(blocks), dim3(256), 0, s, a, items_nope, items_rot, lg)
  if (vec == 16 / sizeof(T)) {
    if constexpr ((16 / sizeof(T)) % 4 == 0) { if (cos_vec) LAUNCH(16 / sizeof(T), true); else LAUNCH(16 / sizeof(T), false); }
    else LAUNCH(16 / sizeof(T), false);
  } else if (vec == 8 / sizeof(T)) {
    if constexpr ((8 / sizeof(T)) % 4 == 0) { if (cos_vec) LAUNCH(8 / sizeof(T), true); else LAUNCH(8 / sizeof(T), false); }
    else LAUNCH(8 / sizeof(T), false);
  } else if (vec == 2) LAUNCH(2, false);
  else LAUNCH(1, false);
#undef LAUNCH
  MOJO_CHECK_LAUNCH("apply_rope");
  return MOJO_OK;
}

// ---------------------------------------------------------------------------------------------
// rotary embedding: cos/sin rows for a list of positions
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rotary_embedding_kernel(float* __restrict__ cos_out, float* __restrict__ sin_out,
                                                               int64_t n_pos, int rope_dim, int mode,
                                                               const int32_t* __restrict__ position_ids,
                                                               const int32_t* __restrict__ cu_q,
                                                               const int32_t* __restrict__ total_lens, int64_t batch,
                                                               const float* __restrict__ cos_table,
                                                               const float* __restrict__ sin_table, int64_t table_len,
                                                               const float* __restrict__ inv_freq, float scaling) {
  const int64_t total = n_pos * rope_dim;
  const int half = rope_dim / 2;
  for (int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
       idx += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t i = idx / rope_dim;
    const int j = static_cast<int>(idx - i * rope_dim);
    int64_t pos;
    if (mode == 0) {
      pos = position_ids[i];
    } else if (mode == 1) {
      pos = i;
    } else {
      pos = -1;                                              // tokens outside every sequence
      if (i >= cu_q[0] && i < cu_q[batch]) {
        int64_t lo = 0, hi = batch;
        while (hi - lo > 1) {
          const int64_t mid = (lo + hi) >> 1;
          if (cu_q[mid] <= i) lo = mid; else hi = mid;
        }
        const int64_t q_len = cu_q[lo + 1] - cu_q[lo];
        const int64_t ctx = total_lens ? static_cast<int64_t>(total_lens[lo]) - q_len : 0;
        pos = ctx + (i - cu_q[lo]);
      }
    }
    if (cos_table) {
      int64_t rowi = pos < 0 ? pos + table_len : pos;        // python-style negative index
      rowi = rowi < 0 ? 0 : (rowi >= table_len ? table_len - 1 : rowi);
      cos_out[idx] = cos_table[rowi * rope_dim + j];
      sin_out[idx] = sin_table[rowi * rope_dim + j];
    } else {
      const float ang = __fmul_rn(static_cast<float>(pos), inv_freq[j >= half ? j - half : j]);
      cos_out[idx] = __fmul_rn(cosf(ang), scaling);
      sin_out[idx] = __fmul_rn(sinf(ang), scaling);
    }
  }
}

}  // namespace mojo

using namespace mojo;

extern "C" int mojo_hip_apply_rope(const void* q, const void* k, void* q_out, void* k_out, const float* cos,
                                   const float* sin, int64_t batch, int64_t tokens, int64_t q_heads, int64_t k_heads,
                                   int64_t head_dim, int64_t rope_dim, const int64_t q_strides[3],
                                   const int64_t k_strides[3], const int64_t qo_strides[3],
                                   const int64_t ko_strides[3], int64_t cos_b_stride, int64_t cos_t_stride, int dtype,
                                   mojo_stream_t stream) {
  if (batch * tokens == 0) return MOJO_OK;
  MOJO_REQUIRE(q && k && q_out && k_out && cos && sin, MOJO_EINVAL, "apply_rope: null pointer");
  MOJO_REQUIRE(rope_dim > 0 && rope_dim % 2 == 0 && rope_dim <= head_dim, MOJO_EINVAL,
               "apply_rope: rope_dim %lld must be even and <= head_dim %lld", (long long)rope_dim, (long long)head_dim);
  RopeArgs a;
  a.src[0] = q; a.src[1] = k; a.dst[0] = q_out; a.dst[1] = k_out;
  a.s_b[0] = q_strides[0]; a.s_t[0] = q_strides[1]; a.s_n[0] = q_strides[2];
  a.s_b[1] = k_strides[0]; a.s_t[1] = k_strides[1]; a.s_n[1] = k_strides[2];
  a.d_b[0] = qo_strides[0]; a.d_t[0] = qo_strides[1]; a.d_n[0] = qo_strides[2];
  a.d_b[1] = ko_strides[0]; a.d_t[1] = ko_strides[1]; a.d_n[1] = ko_strides[2];
  a.heads[0] = static_cast<int>(q_heads); a.heads[1] = static_cast<int>(k_heads);
  a.cos = cos; a.sin = sin; a.cos_b = cos_b_stride; a.cos_t = cos_t_stride;
  a.batch = batch; a.tokens = tokens;
  a.head_dim = static_cast<int>(head_dim); a.rope_dim = static_cast<int>(rope_dim);
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case MOJO_F32: return dispatch_rope<float>(a, s);
    case MOJO_F16: return dispatch_rope<f16_t>(a, s);
    case MOJO_BF16: return dispatch_rope<bf16_t>(a, s);
    default: MOJO_REQUIRE(false, MOJO_EUNSUPPORTED, "apply_rope: dtype %d not supported", dtype);
  }
}

extern "C" int mojo_hip_rotary_embedding(float* cos_out, float* sin_out, int64_t n_pos, int64_t rope_dim, int mode,
                                         const int32_t* position_ids, const int32_t* cu_q_lens,
                                         const int32_t* total_seq_lens, int64_t batch, const float* cos_table,
                                         const float* sin_table, int64_t table_len, const float* inv_freq,
                                         float attention_scaling, mojo_stream_t stream) {
  if (n_pos == 0) return MOJO_OK;
  MOJO_REQUIRE(cos_out && sin_out && rope_dim > 0 && rope_dim % 2 == 0, MOJO_EINVAL, "rotary_embedding: bad arguments");
  MOJO_REQUIRE(mode >= 0 && mode <= 2, MOJO_EINVAL, "rotary_embedding: mode %d", mode);
  MOJO_REQUIRE(mode != 0 || position_ids, MOJO_EINVAL, "rotary_embedding: mode 0 needs position_ids");
  MOJO_REQUIRE(mode != 2 || (cu_q_lens && batch > 0), MOJO_EINVAL, "rotary_embedding: mode 2 needs cu_q_lens");
  MOJO_REQUIRE((cos_table && sin_table && table_len > 0) || inv_freq, MOJO_EINVAL,
               "rotary_embedding: need a cos/sin table or inv_freq");
  int64_t blocks = ceil_div(n_pos * rope_dim, 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(rotary_embedding_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), cos_out,
                     sin_out, n_pos, static_cast<int>(rope_dim), mode, position_ids, cu_q_lens, total_seq_lens, batch,
                     cos_table, sin_table, table_len, inv_freq, attention_scaling);
  MOJO_CHECK_LAUNCH("rotary_embedding");
  return MOJO_OK;
}
