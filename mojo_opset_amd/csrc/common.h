// Shared device/host helpers for libmojo_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <atomic>

#include <type_traits>
#include <utility>

#include "../../include/mojo_hip.h"

namespace mojo {

// Kernels that were built, measured slower and dropped (DESIGN Appendix A) live in csrc/experiments/ and are compiled
// only when the library is built with MOJO_HIP_BUILD_EXPERIMENTS=1 (csrc/build.py adds the define).
#ifdef MOJO_HIP_BUILD_EXPERIMENTS
constexpr bool kExperimentsBuild = true;
#else
constexpr bool kExperimentsBuild = false;
#endif

typedef __bf16 bf16_t;
typedef _Float16 f16_t;

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// thread-local error text ------------------------------------------------------------------
void set_error(const char* fmt, ...);

#define MOJO_REQUIRE(cond, code, ...)                                                   \
  do {                                                                                  \
    if (!(cond)) {                                                                      \
      ::mojo::set_error(__VA_ARGS__);                                                   \
      return (code);                                                                    \
    }                                                                                   \
  } while (0)

#define MOJO_CHECK_LAUNCH(what)                                                         \
  do {                                                                                  \
    hipError_t e__ = hipGetLastError();                                                 \
    if (e__ != hipSuccess) {                                                            \
      ::mojo::set_error("%s: launch failed: %s", (what), hipGetErrorString(e__));       \
      return MOJO_ELAUNCH;                                                              \
    }                                                                                   \
  } while (0)

// run-time switches -----------------------------------------------------------------------
// ONE semantics for every MOJO_HIP_* switch the library reads (round 5): a call site takes its value
// from the environment the first time it runs and LATCHES it; `mojo_hip_reload_env()` drops every latched value, so the
// next call of every site reads the environment again.  A launcher therefore costs one relaxed atomic load per switch, no
// launcher ever calls getenv() concurrently with a setenv() of the host program (reload when no operator call is in
// flight), and a value can still be changed between two calls of one process (tests, A/B probes: set the variable, reload).
// Values are decimal integers — or a short word (MOJO_HIP_MLA_KERNEL=ps), read as its first four characters packed into an
// integer (`switch_word("ps")`); an unset or empty variable means "the default".  `mojo_hip_switches()` lists what was read.
struct EnvSwitch {
  const char* name;
  std::atomic<uint64_t> state{0};          // bits 63..33 generation the value was read at (0 = never), 32 present, 31..0 value
  EnvSwitch* next = nullptr;               // registration list (for mojo_hip_switches)
  explicit EnvSwitch(const char* n);
};
extern std::atomic<uint32_t> g_env_generation;
uint64_t env_refresh(EnvSwitch& s);
inline long long env_get(EnvSwitch& s, long long def) {
  uint64_t st = s.state.load(std::memory_order_relaxed);
  if (static_cast<uint32_t>(st >> 33) != g_env_generation.load(std::memory_order_relaxed)) st = env_refresh(s);
  return ((st >> 32) & 1) ? static_cast<long long>(static_cast<int32_t>(static_cast<uint32_t>(st))) : def;
}
constexpr long long switch_word(const char* w) {
  long long v = 0;
  for (int i = 0; i < 4 && w[i]; ++i) v |= static_cast<long long>(static_cast<unsigned char>(w[i])) << (8 * i);
  return v;
}
#define MOJO_SWITCH(NAME, DEF) (::mojo::env_get(*[] { static ::mojo::EnvSwitch sw_(NAME); return &sw_; }(), (DEF)))

// What the last operator call of this thread launched ("decode_mfma:paired", "gemm256:staged", ...): a debug / test query
// (`mojo_hip_last_launch()`), so that an A/B test can assert that its two legs really took two different forms.
void note_launch(const char* fmt, ...);

static inline bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }
static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
// Threads per row of the RMSNorm kernels (rmsnorm.hip, and the split-K finalize that norms in gemm_skinny.hip: both must cut a
// row alike, the square sums are added in partition order).  Short rows: one wave per row, four rows per block; up to 512
// 16-byte vectors: two waves; else a whole block.  With few rows (a decode batch) a whole block per row as soon as every
// thread has a vector: 64 rows x 4096 would otherwise run on 32 workgroups.
static inline int rms_threads_per_row(int64_t rows, int64_t n_vec) {
  if (rows <= 128 && n_vec >= 256) return 256;
  return n_vec <= 64 * 4 ? 64 : (n_vec <= 128 * 4 ? 128 : 256);
}

// DPP lane exchange (one VALU instruction, no LDS) --------------------------------------------
constexpr int DPP_QUAD_XOR1 = 0xB1;          // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;          // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;   // lane i <-> 7-i inside each 8
constexpr int DPP_ROW_MIRROR = 0x140;        // lane i <-> 15-i inside each 16

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}

// all-reduce over the 16 lanes of a DPP row; every lane ends with the row's sum
__device__ __forceinline__ float row16_sum(float x) {
  x += dpp_mov<DPP_QUAD_XOR1>(x);
  x += dpp_mov<DPP_QUAD_XOR2>(x);
  x += dpp_mov<DPP_ROW_HALF_MIRROR>(x);
  x += dpp_mov<DPP_ROW_MIRROR>(x);
  return x;
}
__device__ __forceinline__ float row8_sum(float x) {
  x += dpp_mov<DPP_QUAD_XOR1>(x);
  x += dpp_mov<DPP_QUAD_XOR2>(x);
  x += dpp_mov<DPP_ROW_HALF_MIRROR>(x);
  return x;
}

// x[l] + x[l ^ 16], then + the same of lane l ^ 32, by lane-row swaps in the vector unit (v_permlane16_swap exchanges the
// odd 16-lane rows of its first operand with the even rows of its second, v_permlane32_swap the upper half of the first
// with the lower half of the second; fed two copies of x they leave {x[l], x[l ^ 16]} resp. {x[l], x[l ^ 32]} in the
// pair).  Same operands and association as the __shfl_xor form it replaces (bit-identical), without the two LDS-crossbar
// round trips of ds_bpermute.  asm: hipcc folds op(swap(x, x)) of the builtin form to x.  s_nop: VALU write -> permlane read.
__device__ __forceinline__ float xor_sum_16_32(float x) {
  float p = x, q = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(p), "+v"(q));
  const float h = p + q;
  p = h; q = h;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(p), "+v"(q));
  return p + q;
}

__device__ __forceinline__ float wave_sum(float x) {
  return xor_sum_16_32(row16_sum(x));
}

// block-wide sum for blockDim.x == 64 * NWAVES; result valid in every thread
template <int NWAVES>
__device__ __forceinline__ float block_sum(float x, float* smem /* >= NWAVES floats */) {
  x = wave_sum(x);
  if constexpr (NWAVES == 1) return x;
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) smem[wave] = x;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NWAVES; ++i) t += smem[i];
  return t;
}

// 2^x as ONE v_exp_f32 (exp2f() expands to a scale / v_exp / v_ldexp / select sequence for denormal results;
// softmax terms that small contribute nothing, and -inf -> 0 holds for the raw instruction too)
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// x * 1/(1 + 2^(-x*log2 e)): v_mul, v_exp, v_add, v_rcp, v_mul.  The library expf() and the IEEE division expand to
// ~25 VALU instructions per element, which made the 16-bit kernels VALU-bound (4.4 of 6.3 TB/s); the hardware
// exp/rcp are accurate to ~1 ulp of fp32, far inside one unit of the 16-bit output's last place.
__device__ __forceinline__ float silu_f(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}

// max over lanes l, l^16, l^32, l^48 without leaving the vector unit.  v_permlane32_swap exchanges the upper half of its
// first operand with the lower half of its second, v_permlane16_swap the odd 16-lane rows of the first with the even rows
// of the second; fed two copies of x they leave {x[l], x[l ^ 32]} resp. {x[l], x[l ^ 16]} in the pair.  asm: hipcc folds
// max(swap(x, x)) of the builtin form to x.  The s_nop covers the VALU-write -> permlane read hazard.
__device__ __forceinline__ float xor_max_16_32(float x) {
  float p = x, q = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(p), "+v"(q));
  const float h = fmaxf(p, q);
  p = h; q = h;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(p), "+v"(q));
  return fmaxf(p, q);
}


// element traits ----------------------------------------------------------------------------
template <typename T> struct elt;
template <> struct elt<float> {
  static __device__ __forceinline__ float to_f(float v) { return v; }
  static __device__ __forceinline__ float from_f(float v) { return v; }
};
template <> struct elt<bf16_t> {
  static __device__ __forceinline__ float to_f(bf16_t v) { return static_cast<float>(v); }
  static __device__ __forceinline__ bf16_t from_f(float v) { return static_cast<bf16_t>(v); }
};
template <> struct elt<f16_t> {
  static __device__ __forceinline__ float to_f(f16_t v) { return static_cast<float>(v); }
  static __device__ __forceinline__ f16_t from_f(float v) { return static_cast<f16_t>(v); }
};

// acc (+ bias) -> storage type.  The golden has BOTH forms (core/operators/compute_with_comm.py:12-24): `input @ weight + bias`
// for [K,N] weights rounds the product and then the sum (two roundings), `F.linear(input, weight, bias)` for [N,K] weights adds
// the bias to the fp32 accumulator and rounds once (torch's addmm; also `MojoGemm.forward`, core/operators/gemm.py:45-46).
template <typename T>
__device__ __forceinline__ T round_with_bias(float acc, T bias, bool fused) {
  return fused ? elt<T>::from_f(acc + elt<T>::to_f(bias)) : elt<T>::from_f(elt<T>::to_f(elt<T>::from_f(acc)) + elt<T>::to_f(bias));
}

template <typename T, int N> struct vec_of { typedef T type __attribute__((ext_vector_type(N))); };
template <typename T> struct vec_of<T, 1> { typedef T type; };

template <typename T, int N>
__device__ __forceinline__ typename vec_of<T, N>::type load_vec(const T* p) {
  return *reinterpret_cast<const typename vec_of<T, N>::type*>(p);
}
template <typename T, int N>
__device__ __forceinline__ void store_vec(T* p, typename vec_of<T, N>::type v) {
  *reinterpret_cast<typename vec_of<T, N>::type*>(p) = v;
}
// Streaming kernels: tensors that are read once and written once and together exceed the Infinity Cache take non-temporal
// loads and stores (round 3: SwiGLU bf16 65536 x 4096 291 -> 263 us, 5.5 -> 6.1 TB/s; fp32 32768 x 4096 289 -> 261 us; tensors
// that fit the caches measured equal or slightly worse, so small calls keep the default policy).  MOJO_HIP_STREAM_NT=0/1 forces.
inline bool stream_nt(long long bytes_moved) {
  const long long f = MOJO_SWITCH("MOJO_HIP_STREAM_NT", -1);
  if (f == 0) return false;
  if (f == 1) return true;
  return bytes_moved >= (256LL << 20);
}
// streaming forms: data read once / written once (no reuse worth a cache line)
template <typename T, int N>
__device__ __forceinline__ typename vec_of<T, N>::type load_vec_nt(const T* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const typename vec_of<T, N>::type*>(p));
}
template <typename T, int N>
__device__ __forceinline__ void store_vec_nt(T* p, typename vec_of<T, N>::type v) {
  __builtin_nontemporal_store(v, reinterpret_cast<typename vec_of<T, N>::type*>(p));
}
template <typename T, int N>
__device__ __forceinline__ T vget(const typename vec_of<T, N>::type& v, int i) {
  if constexpr (N == 1) return v; else return v[i];
}
template <typename T, int N>
__device__ __forceinline__ void vset(typename vec_of<T, N>::type& v, int i, T x) {
  if constexpr (N == 1) v = x; else v[i] = x;
}

// compile-time unrolled loop: f(std::integral_constant<int, I>{}) for I = 0..N-1 (inline-asm "i" operands need
// constant expressions, which a `#pragma unroll` loop variable is not)
template <int... I, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

// True the first time it is called on each device (per call site: pass a function-local static mask).  Used for
// hipFuncSetAttribute, which applies to the kernel object of the CURRENT device only: a process driving several GPUs
// must set it once per device, and the flag must be safe against concurrent callers.
inline bool first_call_on_device(std::atomic<uint64_t>& mask) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = uint64_t{1} << (dev & 63);
  return (mask.fetch_or(bit, std::memory_order_relaxed) & bit) == 0;
}

}  // namespace mojo
