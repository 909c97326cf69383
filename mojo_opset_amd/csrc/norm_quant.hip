// Per-token activation quantisers that feed MojoQuantGemm:
//   MojoDynamicQuant            mojo_opset/core/operators/quantize.py:153-169
//   MojoResidualAddRMSNormQuant mojo_opset/core/operators/normalization.py:493-526
//
//   y     = x.float() [* inv_smooth]                                   (dynamic quant)
//   y     = rms_norm((h + r rounded to T).float(), w, eps) [* smooth]  (norm + quant; fp32 all the way, never rounded to T)
//   scale = max(amax_row |y|, 1e-12) / q_max        (dynamic quant only: 1.0 where that is < 1e-6)
//   q     = clamp(round_half_even(y / scale), q_min, q_max)  -> int8, or float8_e4m3fn of that INTEGER value
//
// One row per 256-thread block; the row's y values stay in registers between the two reductions (sum of squares, row
// maximum) and the quantisation, so every input byte is read once.  Products and the division are the IEEE single
// operations (no FMA contraction, no reciprocal): with the same row statistics the quantised bytes equal the golden's.
//
// Algorithmic bytes per element: dynamic quant elt + 1; norm+quant (pre) 3*elt + 1 (+ weight, smooth once).
#include <math.h>

#include "common.h"

namespace mojo {

struct NormQuantArgs {
  const void* hidden;        // [rows, dim] T
  const void* residual;      // [rows, dim] T or nullptr
  const float* weight;       // [dim] fp32 or nullptr (no normalisation: dynamic quant)
  const float* smooth;       // [dim] fp32 or nullptr
  void* out_q;               // [rows, dim] int8 / fp8
  void* out_sum;             // [rows, dim] T or nullptr      (norm_pos = pre: hidden + residual)
  float* out_normed;         // [rows, dim] fp32 or nullptr   (norm_pos = post: the normed tensor before smoothing)
  float* out_scale;          // [rows]
  int64_t rows;
  int dim;
  float eps, q_max, q_min;
  int fp8;                   // 0: int8, 1: float8_e4m3fn
  int tiny_scale_is_one;     // MojoDynamicQuant's `where(scale < 1e-6, 1.0, scale)`
};

__device__ __forceinline__ float block_max256(float x, float* smem) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o));
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) smem[wave] = x;
  __syncthreads();
  const float r = fmaxf(fmaxf(smem[0], smem[1]), fmaxf(smem[2], smem[3]));
  __syncthreads();
  return r;
}

__device__ __forceinline__ unsigned char to_fp8_e4m3(float v) {
  // v_cvt_pk_fp8_f32: OCP e4m3 (the "fn" format) on gfx950, round to nearest even
  const int packed = __builtin_amdgcn_cvt_pk_fp8_f32(v, v, 0, false);
  return static_cast<unsigned char>(packed & 0xff);
}

// round_half_even(y / scale), bit-identical to the IEEE division, at the price of one multiply for almost every element:
// t = y * (1/scale) is within a few ulp of the true quotient, so rint(t) can only differ from rint(y / scale) when t sits
// within that distance of a rounding boundary (k + 0.5); only those elements take the real division.  |t| <= ~448.
__device__ __forceinline__ float round_quotient(float y, float scale, float inv_scale) {
  const float t = y * inv_scale;
  const float r = rintf(t);
  if (fabsf(fabsf(t - r) - 0.5f) < 1e-3f) return rintf(__fdiv_rn(y, scale));
  return r;
}

// TPR = threads per row: 256 (a row per workgroup) or 64 (a row per WAVE, four rows per workgroup, no LDS and no barrier:
// dynamic quant only — the row maximum is exact in any order, a sum of squares is not).  Many short rows on a workgroup each
// cost ~2 ns of workgroup turnover per row (32 768 rows x 2048: 70 us for 200 MB; x 128: 54 us for 13 MB).
template <typename T, int VEC, int CACHE, bool NT = false /* stream_nt(): activations by-pass the caches */, int TPR = 256>
__global__ __launch_bounds__(256) void norm_quant_kernel(NormQuantArgs a) {
  typedef typename vec_of<T, VEC>::type V;
  __shared__ float red[4];
  const int n_vec = a.dim / VEC;
  const int tid = threadIdx.x % TPR;
  constexpr int RPB = 256 / TPR;                     // rows per workgroup at a time
  const T* hidden = static_cast<const T*>(a.hidden);
  const T* residual = static_cast<const T*>(a.residual);
  T* out_sum = static_cast<T*>(a.out_sum);
  unsigned char* out_q = static_cast<unsigned char*>(a.out_q);

  for (int64_t row = static_cast<int64_t>(blockIdx.x) * RPB + threadIdx.x / TPR; row < a.rows; row += static_cast<int64_t>(gridDim.x) * RPB) {
    const int64_t base = row * a.dim;
    // y of element (v, j); `from_cache` rows only ever take the first branch
    float y[CACHE][VEC];
    float ss = 0.f;
    auto load_sum = [&](int v, float (&f)[VEC]) {
      V x = NT ? load_vec_nt<T, VEC>(hidden + base + v * VEC) : load_vec<T, VEC>(hidden + base + v * VEC);
      if (residual) {
        const V r = NT ? load_vec_nt<T, VEC>(residual + base + v * VEC) : load_vec<T, VEC>(residual + base + v * VEC);
#pragma unroll
        for (int j = 0; j < VEC; ++j)
          vset<T, VEC>(x, j, elt<T>::from_f(elt<T>::to_f(vget<T, VEC>(x, j)) + elt<T>::to_f(vget<T, VEC>(r, j))));
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) f[j] = elt<T>::to_f(vget<T, VEC>(x, j));
      return x;
    };
    // ---- pass 1: sums, sum of squares ---------------------------------------------------------------------------
    // The cached vectors are loaded in batches — all of hidden, then (one wave-uniform branch) all of residual — with
    // the index clamped instead of a branch per vector: a branch around each load made hipcc wait for every vector before
    // requesting the next (one 16-byte load in flight per lane, four dependent round trips per row).
    {
      V xr[CACHE];
#pragma unroll
      for (int c = 0; c < CACHE; ++c) xr[c] = NT ? load_vec_nt<T, VEC>(hidden + base + min(tid + c * TPR, n_vec - 1) * VEC) : load_vec<T, VEC>(hidden + base + min(tid + c * TPR, n_vec - 1) * VEC);
      if (residual) {
        V rr[CACHE];
#pragma unroll
        for (int c = 0; c < CACHE; ++c) rr[c] = NT ? load_vec_nt<T, VEC>(residual + base + min(tid + c * TPR, n_vec - 1) * VEC) : load_vec<T, VEC>(residual + base + min(tid + c * TPR, n_vec - 1) * VEC);
#pragma unroll
        for (int c = 0; c < CACHE; ++c)
#pragma unroll
          for (int j = 0; j < VEC; ++j)
            vset<T, VEC>(xr[c], j, elt<T>::from_f(elt<T>::to_f(vget<T, VEC>(xr[c], j)) + elt<T>::to_f(vget<T, VEC>(rr[c], j))));
      }
      if (out_sum) {
#pragma unroll
        for (int c = 0; c < CACHE; ++c)
          if (tid + c * TPR < n_vec) { if (NT) store_vec_nt<T, VEC>(out_sum + base + (tid + c * TPR) * VEC, xr[c]); else store_vec<T, VEC>(out_sum + base + (tid + c * TPR) * VEC, xr[c]); }
      }
#pragma unroll
      for (int c = 0; c < CACHE; ++c) {
        const bool live = tid + c * TPR < n_vec;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          y[c][j] = live ? elt<T>::to_f(vget<T, VEC>(xr[c], j)) : 0.f;           // clamped duplicates count as zeros
          ss = fmaf(y[c][j], y[c][j], ss);
        }
      }
    }
    for (int v = tid + CACHE * TPR; v < n_vec; v += TPR) {
      float f[VEC];
      const V x = load_sum(v, f);
      if (out_sum) { if (NT) store_vec_nt<T, VEC>(out_sum + base + v * VEC, x); else store_vec<T, VEC>(out_sum + base + v * VEC, x); }
#pragma unroll
      for (int j = 0; j < VEC; ++j) ss = fmaf(f[j], f[j], ss);
    }
    float rstd = 1.f;
    if (a.weight) {
      ss = block_sum<4>(ss, red);
      __syncthreads();
      rstd = rsqrtf(ss / static_cast<float>(a.dim) + a.eps);
    }
    // y = ((f * rstd) * w) [* smooth]: single IEEE multiplies, the golden's order
    auto load_f32 = [&](const float* p, int v, float (&dst)[VEC]) {        // VEC floats of a per-column vector
      if constexpr (VEC % 4 == 0) {
#pragma unroll
        for (int q = 0; q < VEC / 4; ++q) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(p + v * VEC + 4 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) dst[4 * q + e] = t[e];
        }
      } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) dst[j] = p[v * VEC + j];
      }
    };
    auto finish = [&](int v, float (&f)[VEC]) {
      if (a.weight) {
        float w[VEC];
        load_f32(a.weight, v, w);
#pragma unroll
        for (int j = 0; j < VEC; ++j) f[j] = __fmul_rn(__fmul_rn(f[j], rstd), w[j]);
        if (a.out_normed) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) a.out_normed[base + v * VEC + j] = f[j];
        }
      }
      if (a.smooth) {
        float sm[VEC];
        load_f32(a.smooth, v, sm);
#pragma unroll
        for (int j = 0; j < VEC; ++j) f[j] = __fmul_rn(f[j], sm[j]);
      }
    };
    // ---- pass 2: y, row maximum ----------------------------------------------------------------------------------
    float amax = 0.f;
    // cached part: the per-column vectors are fetched in batches too (clamped index, one uniform branch per tensor)
    if (a.weight) {
      float w[CACHE][VEC];
#pragma unroll
      for (int c = 0; c < CACHE; ++c) load_f32(a.weight, min(tid + c * TPR, n_vec - 1), w[c]);
#pragma unroll
      for (int c = 0; c < CACHE; ++c)
#pragma unroll
        for (int j = 0; j < VEC; ++j) y[c][j] = __fmul_rn(__fmul_rn(y[c][j], rstd), w[c][j]);
      if (a.out_normed) {
#pragma unroll
        for (int c = 0; c < CACHE; ++c)
          if (tid + c * TPR < n_vec) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) a.out_normed[base + (tid + c * TPR) * VEC + j] = y[c][j];
          }
      }
    }
    if (a.smooth) {
      float sm[CACHE][VEC];
#pragma unroll
      for (int c = 0; c < CACHE; ++c) load_f32(a.smooth, min(tid + c * TPR, n_vec - 1), sm[c]);
#pragma unroll
      for (int c = 0; c < CACHE; ++c)
#pragma unroll
        for (int j = 0; j < VEC; ++j) y[c][j] = __fmul_rn(y[c][j], sm[c][j]);
    }
#pragma unroll
    for (int c = 0; c < CACHE; ++c) {
      if (tid + c * TPR < n_vec) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) amax = fmaxf(amax, fabsf(y[c][j]));
      }
    }
    for (int v = tid + CACHE * TPR; v < n_vec; v += TPR) {
      float f[VEC];
      load_sum(v, f);
      finish(v, f);
#pragma unroll
      for (int j = 0; j < VEC; ++j) amax = fmaxf(amax, fabsf(f[j]));
    }
    if constexpr (TPR == 256) {
      amax = block_max256(amax, red);
    } else {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    }
    float scale = __fdiv_rn(fmaxf(amax, 1e-12f), a.q_max);
    if (a.tiny_scale_is_one && scale < 1e-6f) scale = 1.0f;
    if (tid == 0) a.out_scale[row] = scale;
    const float inv_scale = __fdiv_rn(1.0f, scale);
    // ---- pass 3: quantise ------------------------------------------------------------------------------------------
    auto emit = [&](int v, const float (&f)[VEC]) {
      // round_half_even(y / scale) as rint(y * (1 / scale)); the exact-division fallback is decided once per VECTOR (a
      // divergent branch per element costs more vector instructions than the arithmetic it guards).  |t - rint(t)| <= 0.5
      // always, so "within 1e-3 of a rounding boundary" is "its maximum over the vector > 0.499".
      float rq[VEC];
      float near = 0.f;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float t = f[j] * inv_scale;
        rq[j] = rintf(t);
        near = fmaxf(near, fabsf(t - rq[j]));
      }
      if (near > 0.499f) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) rq[j] = round_quotient(f[j], scale, inv_scale);
      }
      unsigned char q[VEC];
      if (a.fp8) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) q[j] = to_fp8_e4m3(fminf(fmaxf(rq[j], a.q_min), a.q_max));
      } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j)
          q[j] = static_cast<unsigned char>(static_cast<signed char>(static_cast<int>(fminf(fmaxf(rq[j], a.q_min), a.q_max))));
      }
      unsigned char* dst = out_q + base + v * VEC;
      if constexpr (VEC == 8) {
        u32x2 w;
        w[0] = q[0] | (q[1] << 8) | (q[2] << 16) | (static_cast<unsigned>(q[3]) << 24);
        w[1] = q[4] | (q[5] << 8) | (q[6] << 16) | (static_cast<unsigned>(q[7]) << 24);
        if (NT) __builtin_nontemporal_store(w, reinterpret_cast<u32x2*>(dst)); else *reinterpret_cast<u32x2*>(dst) = w;
      } else if constexpr (VEC == 4) {
        *reinterpret_cast<unsigned*>(dst) = q[0] | (q[1] << 8) | (q[2] << 16) | (static_cast<unsigned>(q[3]) << 24);
      } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) dst[j] = q[j];
      }
    };
#pragma unroll
    for (int c = 0; c < CACHE; ++c) {
      const int v = tid + c * TPR;
      if (v < n_vec) emit(v, y[c]);
    }
    for (int v = tid + CACHE * TPR; v < n_vec; v += TPR) {
      float f[VEC];
      load_sum(v, f);
      // (the stored normed tensor was written in pass 2; write it again is harmless but wasteful: skip it)
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float t = f[j];
        if (a.weight) t = __fmul_rn(__fmul_rn(t, rstd), a.weight[v * VEC + j]);
        if (a.smooth) t = __fmul_rn(t, a.smooth[v * VEC + j]);
        f[j] = t;
      }
      emit(v, f);
    }
  }
}

template <typename T>
static int launch_norm_quant(const NormQuantArgs& a, hipStream_t s) {
  constexpr int WIDE = 16 / sizeof(T);
  const size_t al = WIDE * sizeof(T);
  const bool wide = a.dim % WIDE == 0 && aligned_to(a.hidden, al) && (!a.residual || aligned_to(a.residual, al)) &&
                    (!a.out_sum || aligned_to(a.out_sum, al)) && aligned_to(a.out_q, WIDE) &&
                    (!a.weight || aligned_to(a.weight, 16)) && (!a.smooth || aligned_to(a.smooth, 16));
  int64_t blocks = a.rows > 256 * 32 ? 256 * 32 : a.rows;
  const long long moved = a.rows * static_cast<long long>(a.dim) * (static_cast<long long>(sizeof(T)) * (1 + (a.residual ? 1 : 0) + (a.out_sum ? 1 : 0)) + 1);
  // a row per wave: dynamic quant (no normalisation: no sum of squares) of many rows that fit one wave's registers
  if (wide && !a.weight && a.dim / WIDE <= 4 * 64 && a.rows >= 2048) {
    blocks = ceil_div(a.rows, 4);
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (stream_nt(moved)) hipLaunchKernelGGL((norm_quant_kernel<T, WIDE, 4, true, 64>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((norm_quant_kernel<T, WIDE, 4, false, 64>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, a);
    MOJO_CHECK_LAUNCH("norm_quant");
    note_launch("norm_quant:row_per_wave");
    return MOJO_OK;
  }
  if (wide && stream_nt(moved)) hipLaunchKernelGGL((norm_quant_kernel<T, WIDE, 4, true>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, a);
  else if (wide) hipLaunchKernelGGL((norm_quant_kernel<T, WIDE, 4>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((norm_quant_kernel<T, 1, 8>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, a);
  MOJO_CHECK_LAUNCH("norm_quant");
  return MOJO_OK;
}

static int dispatch_norm_quant(const NormQuantArgs& a, int dtype, hipStream_t s) {
  switch (dtype) {
    case MOJO_F32: return launch_norm_quant<float>(a, s);
    case MOJO_F16: return launch_norm_quant<f16_t>(a, s);
    case MOJO_BF16: return launch_norm_quant<bf16_t>(a, s);
    default: MOJO_REQUIRE(false, MOJO_EUNSUPPORTED, "norm_quant: dtype %d not supported", dtype);
  }
}

}  // namespace mojo

using namespace mojo;

extern "C" int mojo_hip_dynamic_quant(const void* input, const float* inv_smooth_scale, void* out_q, float* out_scale,
                                      int64_t rows, int64_t dim, int dtype, mojo_stream_t stream) {
  if (rows == 0) return MOJO_OK;
  MOJO_REQUIRE(input && out_q && out_scale, MOJO_EINVAL, "dynamic_quant: null pointer");
  MOJO_REQUIRE(rows > 0 && dim > 0 && dim < (1LL << 30), MOJO_EINVAL, "dynamic_quant: bad shape rows=%lld dim=%lld",
               (long long)rows, (long long)dim);
  NormQuantArgs a{};
  a.hidden = input; a.smooth = inv_smooth_scale; a.out_q = out_q; a.out_scale = out_scale;
  a.rows = rows; a.dim = static_cast<int>(dim);
  a.eps = 0.f; a.q_max = 127.f; a.q_min = -128.f; a.fp8 = 0; a.tiny_scale_is_one = 1;
  return dispatch_norm_quant(a, dtype, static_cast<hipStream_t>(stream));
}

extern "C" int mojo_hip_residual_add_rmsnorm_quant(const void* hidden, const void* residual, const float* weight,
                                                   const float* smooth_scale, void* out_q, void* out_sum,
                                                   float* out_normed, float* out_scale, int64_t rows, int64_t dim,
                                                   int dtype, int quant_dtype, float q_min, float eps,
                                                   mojo_stream_t stream) {
  if (rows == 0) return MOJO_OK;
  MOJO_REQUIRE(hidden && weight && out_q && out_scale, MOJO_EINVAL, "rmsnorm_quant: null pointer");
  MOJO_REQUIRE(rows > 0 && dim > 0 && dim < (1LL << 30), MOJO_EINVAL, "rmsnorm_quant: bad shape rows=%lld dim=%lld",
               (long long)rows, (long long)dim);
  MOJO_REQUIRE(quant_dtype == MOJO_I8 || quant_dtype == MOJO_F8E4M3, MOJO_EUNSUPPORTED,
               "rmsnorm_quant: quant dtype %d (int8 / fp8-e4m3 only)", quant_dtype);
  NormQuantArgs a{};
  a.hidden = hidden; a.residual = residual; a.weight = weight; a.smooth = smooth_scale;
  a.out_q = out_q; a.out_sum = out_sum; a.out_normed = out_normed; a.out_scale = out_scale;
  a.rows = rows; a.dim = static_cast<int>(dim);
  a.eps = eps; a.fp8 = quant_dtype == MOJO_F8E4M3;
  a.q_max = a.fp8 ? 448.f : 127.f;
  a.q_min = q_min;
  a.tiny_scale_is_one = 0;
  return dispatch_norm_quant(a, dtype, static_cast<hipStream_t>(stream));
}
