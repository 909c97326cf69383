// MojoSwiGLU: out = silu(gate) * up, optional clamp.  Pure HBM streaming: 3 tensors, 16 B per lane.
// Rounding mirrors the torch golden: for 16-bit types silu(gate) is rounded to the storage type
// BEFORE the multiply (the golden runs two elementwise ops), for fp32 there is nothing to mirror.
//
// Algorithmic bytes per element: 3 x elt.
#include "common.h"

namespace mojo {

template <typename T, int VEC>
__device__ __forceinline__ typename vec_of<T, VEC>::type swiglu_vec(const typename vec_of<T, VEC>::type& g,
                                                                   const typename vec_of<T, VEC>::type& u, float limit) {
  typename vec_of<T, VEC>::type o;
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    float gf = elt<T>::to_f(vget<T, VEC>(g, j));
    float uf = elt<T>::to_f(vget<T, VEC>(u, j));
    if (limit > 0.f) {
      uf = fminf(fmaxf(uf, -limit), limit);
      gf = fminf(gf, limit);
    }
    const float s = elt<T>::to_f(elt<T>::from_f(silu_f(gf)));   // round like the golden's F.silu
    vset<T, VEC>(o, j, elt<T>::from_f(s * uf));
  }
  return o;
}

// Each thread takes UNROLL vectors per trip, all 2*UNROLL loads issued before the first use: with one vector per trip
// a wave has 2 KiB in flight, too little to cover HBM latency at 8 resident waves per SIMD.
template <typename T, int VEC, int UNROLL, bool NT>
__global__ __launch_bounds__(256) void swiglu_kernel(const T* __restrict__ gate, const T* __restrict__ up,
                                                     T* __restrict__ out, int64_t n_vec, float limit) {
  typedef typename vec_of<T, VEC>::type V;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * (256 * UNROLL);
  for (int64_t base = static_cast<int64_t>(blockIdx.x) * (256 * UNROLL); base < n_vec; base += stride) {
    V g[UNROLL], u[UNROLL];
    if (base + 256 * UNROLL <= n_vec) {
#pragma unroll
      for (int k = 0; k < UNROLL; ++k) {
        const int64_t i = base + k * 256 + threadIdx.x;
        g[k] = NT ? load_vec_nt<T, VEC>(gate + i * VEC) : load_vec<T, VEC>(gate + i * VEC);
        u[k] = NT ? load_vec_nt<T, VEC>(up + i * VEC) : load_vec<T, VEC>(up + i * VEC);
      }
#pragma unroll
      for (int k = 0; k < UNROLL; ++k) {
        const int64_t i = base + k * 256 + threadIdx.x;
        if (NT) store_vec_nt<T, VEC>(out + i * VEC, swiglu_vec<T, VEC>(g[k], u[k], limit));
        else store_vec<T, VEC>(out + i * VEC, swiglu_vec<T, VEC>(g[k], u[k], limit));
      }
    } else {
      for (int64_t i = base + threadIdx.x; i < n_vec; i += 256)
        store_vec<T, VEC>(out + i * VEC, swiglu_vec<T, VEC>(load_vec<T, VEC>(gate + i * VEC), load_vec<T, VEC>(up + i * VEC), limit));
    }
  }
}

// row-strided variant: gate / up / out are [rows, cols] views with their own row strides (the two halves of a fused
// [rows, 2*cols] projection output, MojoExperts)
template <typename T, int VEC>
__global__ __launch_bounds__(256) void swiglu_rows_kernel(const T* __restrict__ gate, const T* __restrict__ up,
                                                          T* __restrict__ out, int64_t rows, int cols_vec, int64_t ld_gate,
                                                          int64_t ld_up, int64_t ld_out, float limit) {
  typedef typename vec_of<T, VEC>::type V;
  const int64_t total = rows * cols_vec;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / cols_vec;
    const int c = static_cast<int>(i - r * cols_vec) * VEC;
    const V g = load_vec<T, VEC>(gate + r * ld_gate + c);
    const V u = load_vec<T, VEC>(up + r * ld_up + c);
    store_vec<T, VEC>(out + r * ld_out + c, swiglu_vec<T, VEC>(g, u, limit));
  }
}

template <typename T>
static int launch_swiglu_rows(const void* gate, const void* up, void* out, int64_t rows, int64_t cols, int64_t ld_gate,
                              int64_t ld_up, int64_t ld_out, float limit, hipStream_t s) {
  constexpr int WIDE = 16 / sizeof(T);
  const bool wide = cols % WIDE == 0 && ld_gate % WIDE == 0 && ld_up % WIDE == 0 && ld_out % WIDE == 0 &&
                    aligned_to(gate, 16) && aligned_to(up, 16) && aligned_to(out, 16);
  const int64_t total = wide ? rows * (cols / WIDE) : rows * cols;
  int64_t blocks = ceil_div(total, 256);
  if (blocks > 256 * 32) blocks = 256 * 32;
  if (wide)
    hipLaunchKernelGGL((swiglu_rows_kernel<T, WIDE>), dim3(blocks), dim3(256), 0, s, static_cast<const T*>(gate),
                       static_cast<const T*>(up), static_cast<T*>(out), rows, static_cast<int>(cols / WIDE), ld_gate, ld_up,
                       ld_out, limit);
  else
    hipLaunchKernelGGL((swiglu_rows_kernel<T, 1>), dim3(blocks), dim3(256), 0, s, static_cast<const T*>(gate),
                       static_cast<const T*>(up), static_cast<T*>(out), rows, static_cast<int>(cols), ld_gate, ld_up, ld_out,
                       limit);
  MOJO_CHECK_LAUNCH("swiglu_rows");
  return MOJO_OK;
}

template <typename T>
static int launch_swiglu(const void* gate, const void* up, void* out, int64_t n, float limit, hipStream_t s) {
  constexpr int WIDE = 16 / sizeof(T);
  const bool wide = n % WIDE == 0 && aligned_to(gate, 16) && aligned_to(up, 16) && aligned_to(out, 16);
  const int64_t n_vec = wide ? n / WIDE : n;
  constexpr int UNROLL = 4;
  int64_t blocks = ceil_div(n_vec, 256 * UNROLL);
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (wide) {
    const T* g_ = static_cast<const T*>(gate); const T* u_ = static_cast<const T*>(up); T* o_ = static_cast<T*>(out);
    if (stream_nt(3 * n * static_cast<long long>(sizeof(T))))
      hipLaunchKernelGGL((swiglu_kernel<T, WIDE, UNROLL, true>), dim3(blocks), dim3(256), 0, s, g_, u_, o_, n_vec, limit);
    else
      hipLaunchKernelGGL((swiglu_kernel<T, WIDE, UNROLL, false>), dim3(blocks), dim3(256), 0, s, g_, u_, o_, n_vec, limit);
  } else {
    hipLaunchKernelGGL((swiglu_kernel<T, 1, UNROLL, false>), dim3(blocks), dim3(256), 0, s, static_cast<const T*>(gate),
                       static_cast<const T*>(up), static_cast<T*>(out), n_vec, limit);
  }
  MOJO_CHECK_LAUNCH("swiglu");
  return MOJO_OK;
}

}  // namespace mojo

using namespace mojo;

extern "C" int mojo_hip_swiglu(const void* gate, const void* up, void* out, int64_t n, int dtype, float swiglu_limit,
                               mojo_stream_t stream) {
  if (n == 0) return MOJO_OK;
  MOJO_REQUIRE(gate && up && out && n > 0, MOJO_EINVAL, "swiglu: null pointer or negative size");
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case MOJO_F32: return launch_swiglu<float>(gate, up, out, n, swiglu_limit, s);
    case MOJO_F16: return launch_swiglu<f16_t>(gate, up, out, n, swiglu_limit, s);
    case MOJO_BF16: return launch_swiglu<bf16_t>(gate, up, out, n, swiglu_limit, s);
    default: MOJO_REQUIRE(false, MOJO_EUNSUPPORTED, "swiglu: dtype %d not supported", dtype);
  }
}

extern "C" int mojo_hip_swiglu_rows(const void* gate, const void* up, void* out, int64_t rows, int64_t cols,
                                    int64_t ld_gate, int64_t ld_up, int64_t ld_out, int dtype, float swiglu_limit,
                                    mojo_stream_t stream) {
  if (rows == 0 || cols == 0) return MOJO_OK;
  MOJO_REQUIRE(gate && up && out && rows > 0 && cols > 0 && cols < (1LL << 31), MOJO_EINVAL, "swiglu_rows: bad arguments");
  MOJO_REQUIRE(ld_gate >= cols && ld_up >= cols && ld_out >= cols, MOJO_EINVAL, "swiglu_rows: row stride smaller than the row");
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case MOJO_F32: return launch_swiglu_rows<float>(gate, up, out, rows, cols, ld_gate, ld_up, ld_out, swiglu_limit, s);
    case MOJO_F16: return launch_swiglu_rows<f16_t>(gate, up, out, rows, cols, ld_gate, ld_up, ld_out, swiglu_limit, s);
    case MOJO_BF16: return launch_swiglu_rows<bf16_t>(gate, up, out, rows, cols, ld_gate, ld_up, ld_out, swiglu_limit, s);
    default: MOJO_REQUIRE(false, MOJO_EUNSUPPORTED, "swiglu_rows: dtype %d not supported", dtype);
  }
}
