// MojoResidualAddRMSNorm / MojoRMSNorm.
//   sum    = round_T(hidden + residual)            (residual optional)
//   normed = round_T(sum * rsqrt(mean(sum^2) + eps) * weight)     -- fp32 math, ONE rounding,
// which is what torch's F.rms_norm does on CPU, not the Triton "llama" double rounding.
//
// One row per TPR threads (64 for short rows, 256 otherwise); the row lives in registers between the
// reduction and the scale pass, so every input byte is read once.  Rows too long for the register
// cache are recomputed in a second pass.
//
// Algorithmic bytes per row (pre, with residual): 2 reads + 2 writes of D x elt, + weight once.
#include "common.h"

namespace mojo {

template <typename T, int VEC, int TPR, int CACHE /* vectors cached per thread */, bool NT /* stream_nt(): activations by-pass the caches */>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const T* hidden /* may alias normed (in-place norm) */,
                                                      const T* __restrict__ residual,
                                                      const T* __restrict__ weight, T* normed,
                                                      T* __restrict__ summed, int64_t rows, int dim, float eps) {
  typedef typename vec_of<T, VEC>::type V;
  constexpr int ROWS_PER_BLOCK = 256 / TPR;
  constexpr int NW = TPR / 64;
  __shared__ float red[4];
  const int sub = threadIdx.x / TPR;
  const int tid = threadIdx.x % TPR;
  const int n_vec = dim / VEC;
  const float inv_dim = 1.0f / static_cast<float>(dim);

  // (the trip count is the same for every thread of the block — the NW > 1 forms synchronise inside the loop; a row
  // group past the end recomputes the last row and stores nothing)
  for (int64_t row0 = static_cast<int64_t>(blockIdx.x) * ROWS_PER_BLOCK; row0 < rows;
       row0 += static_cast<int64_t>(gridDim.x) * ROWS_PER_BLOCK) {
    const bool live = row0 + sub < rows;
    const int64_t row = live ? row0 + sub : rows - 1;
    const T* h = hidden + row * dim;
    const T* r = residual ? residual + row * dim : nullptr;
    V cache[CACHE];
    float ss = 0.f;
    // pass 1: sum (rounded to T), square-accumulate; keep the first CACHE vectors in registers
    int c = 0;
    for (int v = tid; v < n_vec; v += TPR, ++c) {
      V x = NT ? load_vec_nt<T, VEC>(h + v * VEC) : load_vec<T, VEC>(h + v * VEC);
      if (r) {
        const V y = NT ? load_vec_nt<T, VEC>(r + v * VEC) : load_vec<T, VEC>(r + v * VEC);
#pragma unroll
        for (int j = 0; j < VEC; ++j)
          vset<T, VEC>(x, j, elt<T>::from_f(elt<T>::to_f(vget<T, VEC>(x, j)) + elt<T>::to_f(vget<T, VEC>(y, j))));
        if (summed && live) {
          if (NT) store_vec_nt<T, VEC>(summed + row * dim + v * VEC, x); else store_vec<T, VEC>(summed + row * dim + v * VEC, x);
        }
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float f = elt<T>::to_f(vget<T, VEC>(x, j));
        ss += f * f;
      }
      if (c < CACHE) cache[c] = x;   // c is uniform across the unrolled prefix; see below
    }
    if constexpr (NW == 1) {
      ss = wave_sum(ss);
    } else {                          // NW waves per row, 4 / NW rows per block: sum the row's own waves
      ss = wave_sum(ss);
      const int wave = threadIdx.x >> 6;
      if ((threadIdx.x & 63) == 0) red[wave] = ss;
      __syncthreads();
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < NW; ++i) t += red[sub * NW + i];
      ss = t;
      __syncthreads();                // `red` is reused by the next row
    }
    const float rstd = rsqrtf(ss * inv_dim + eps);
    // pass 2
    c = 0;
    for (int v = tid; v < n_vec; v += TPR, ++c) {
      V x;
      if (c < CACHE) {
        x = cache[c];
      } else {
        x = load_vec<T, VEC>(h + v * VEC);
        if (r) {
          const V y = load_vec<T, VEC>(r + v * VEC);
#pragma unroll
          for (int j = 0; j < VEC; ++j)
            vset<T, VEC>(x, j, elt<T>::from_f(elt<T>::to_f(vget<T, VEC>(x, j)) + elt<T>::to_f(vget<T, VEC>(y, j))));
        }
      }
      const V w = load_vec<T, VEC>(weight + v * VEC);
      V o;
#pragma unroll
      for (int j = 0; j < VEC; ++j)
        vset<T, VEC>(o, j, elt<T>::from_f(elt<T>::to_f(vget<T, VEC>(x, j)) * rstd * elt<T>::to_f(vget<T, VEC>(w, j))));
      if (live) {
        if (NT) store_vec_nt<T, VEC>(normed + row * dim + v * VEC, o); else store_vec<T, VEC>(normed + row * dim + v * VEC, o);
      }
    }
  }
}

template <typename T, int VEC>
static void launch_rms(const void* hidden, const void* residual, const void* weight, void* normed, void* summed,
                       int64_t rows, int64_t dim, float eps, hipStream_t s) {
  const int64_t n_vec = dim / VEC;
  const T* h = static_cast<const T*>(hidden);
  const T* r = static_cast<const T*>(residual);
  const T* w = static_cast<const T*>(weight);
  T* o = static_cast<T*>(normed);
  T* so = static_cast<T*>(summed);
  // (an in-place norm reads what it wrote only through registers: the non-temporal forms are safe for it too)
  const bool nt = stream_nt(rows * dim * static_cast<long long>(sizeof(T)) * (residual ? (summed ? 4 : 3) : 2));
  auto go = [&](auto tpr_tag, int64_t blocks) {
    constexpr int TPR = decltype(tpr_tag)::value;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (nt) hipLaunchKernelGGL((rmsnorm_kernel<T, VEC, TPR, 4, true>), dim3(blocks), dim3(256), 0, s, h, r, w, o, so, rows, static_cast<int>(dim), eps);
    else hipLaunchKernelGGL((rmsnorm_kernel<T, VEC, TPR, 4, false>), dim3(blocks), dim3(256), 0, s, h, r, w, o, so, rows, static_cast<int>(dim), eps);
  };
  const int tpr = VEC * sizeof(T) == 16 ? rms_threads_per_row(rows, n_vec) : (n_vec <= 64 * 4 ? 64 : (n_vec <= 128 * 4 ? 128 : 256));
  if (tpr == 64) go(std::integral_constant<int, 64>{}, ceil_div(rows, 4));          // short rows: one wave per row, 4 rows per block
  else if (tpr == 128) go(std::integral_constant<int, 128>{}, ceil_div(rows, 2));   // two waves per row, two rows per block
  else go(std::integral_constant<int, 256>{}, rows);
}

template <typename T>
static int dispatch_rms(const void* hidden, const void* residual, const void* weight, void* normed, void* summed,
                        int64_t rows, int64_t dim, float eps, hipStream_t s) {
  constexpr int WIDE = 16 / sizeof(T);
  auto ok = [&](int vec) {
    const size_t a = vec * sizeof(T);
    return dim % vec == 0 && aligned_to(hidden, a) && aligned_to(weight, a) && aligned_to(normed, a) &&
           (!residual || aligned_to(residual, a)) && (!summed || aligned_to(summed, a));
  };
  if (ok(WIDE)) launch_rms<T, WIDE>(hidden, residual, weight, normed, summed, rows, dim, eps, s);
  else if (ok(2)) launch_rms<T, 2>(hidden, residual, weight, normed, summed, rows, dim, eps, s);
  else launch_rms<T, 1>(hidden, residual, weight, normed, summed, rows, dim, eps, s);
  MOJO_CHECK_LAUNCH("residual_add_rmsnorm");
  return MOJO_OK;
}

}  // namespace mojo

using namespace mojo;

extern "C" int mojo_hip_residual_add_rmsnorm(const void* hidden, const void* residual, const void* weight,
                                             void* normed_out, void* sum_out, int64_t rows, int64_t dim, int dtype,
                                             float eps, mojo_stream_t stream) {
  if (rows == 0) return MOJO_OK;
  MOJO_REQUIRE(hidden && weight && normed_out, MOJO_EINVAL, "rmsnorm: null pointer");
  MOJO_REQUIRE(rows > 0 && dim > 0 && dim < (1LL << 30), MOJO_EINVAL, "rmsnorm: bad shape rows=%lld dim=%lld",
               (long long)rows, (long long)dim);
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case MOJO_F32: return dispatch_rms<float>(hidden, residual, weight, normed_out, sum_out, rows, dim, eps, s);
    case MOJO_F16: return dispatch_rms<f16_t>(hidden, residual, weight, normed_out, sum_out, rows, dim, eps, s);
    case MOJO_BF16: return dispatch_rms<bf16_t>(hidden, residual, weight, normed_out, sum_out, rows, dim, eps, s);
    default: MOJO_REQUIRE(false, MOJO_EUNSUPPORTED, "rmsnorm: dtype %d not supported", dtype);
  }
}
