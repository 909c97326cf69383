// Grouped GEMM for decode-sized groups (K-major "[N,K]" weights, 16-bit types): equal-sized groups of <= 128 rows (dense
// products, the MLA per-head projections) and, in the RAGGED form, groups of any size that average <= 64 rows (MoE experts at decode).
//
// With few rows per group the product is a weight STREAM: the 256x256 tile kernel pads every group to 256 rows and spends
// its time in prologue and epilogue (MLA's per-head projections: 64 rows per head, K = 128 or 512).  Here one workgroup =
// 64 output columns x all rows of one group; structure as quant_skinny_kernel (quant_gemm.hip): each wave owns 16 columns,
// loads their weight rows row-contiguously (4 rows x 256 B per instruction) and restores the MFMA fragment shape through a
// wave-private LDS image; the group's activation block [16*MT rows][256 B of K] is shared through a padded LDS image;
// weights of the next three K blocks and activations of the next two are in flight while a block is multiplied.
// Row maps (gemm.h) are honoured, so the MLA shim's token-major tensors are read and written in place.
//
// Algorithmic bytes: G*K*N*elt (weights) + M*K*elt * (N/64) (activations, from L2) + M*N*elt.
#include "gemm.h"
#ifdef MOJO_HIP_BUILD_EXPERIMENTS        // in-launch split-K combine, measured slower (DESIGN Appendix A #9): opt-in build only
#include "experiments/splitk_combine.h"
#endif

namespace mojo {

template <typename T> struct skinny_mfma;
template <> struct skinny_mfma<bf16_t> {
  static __device__ __forceinline__ f32x4 run(u32x4 w, u32x4 a, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), c, 0, 0, 0);
  }
};
template <> struct skinny_mfma<f16_t> {
  static __device__ __forceinline__ f32x4 run(u32x4 w, u32x4 a, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, a), c, 0, 0, 0);
  }
};

// RB = row blocks per group (1 or 2): groups of more than 64 rows run as two 64-row blocks — a 128-row workgroup needs
// 104 KiB of LDS (one workgroup of four waves per CU) and reads its activation fragments from LDS eight times per weight
// byte; two 64-row workgroups share a CU, and the second reads the weight lines the first just pulled into L2 (default
// cache policy instead of non-temporal loads).
//
// RAGGED: groups of different sizes (the experts of an MoE layer at decode: a few rows each, counts known on the device
// only).  blockIdx.y is then a 64-row block of the prefix arrays built for tile height 64 (gemm_locate_tile): group, first
// row and the group's end; blocks past the last one exit.  Row maps are not supported in this form.
//
// NW = waves per workgroup (each wave owns 16 weight rows; the waves step through K together and share the activation
// image).  GLU (dense, no K split): the weight is [gate | up] = [2 I, K]; a wave's 16 rows are gate rows c .. c+7 and up rows
// I + c .. I + c + 7, lanes l and l ^ 32 hold a (gate, up) pair after the last MFMA, and the epilogue stores
// round(round(silu(round(gate))) * round(up)) — the rounding points of GEMM -> MojoSwiGLU run one after the other — to
// C [M, I].  NW is chosen on the host so that the wave units divide evenly over the CUs (gemm_skinny_glu_waves): with 64-column
// workgroups Llama-3-8B's gate|up projection is 448 workgroups on 512 slots — three CUs in four stream twice the bytes of
// the fourth; seven-wave workgroups are one per CU.
template <typename T, int MT, int RB, bool RAGGED = false, int NW = 4, bool GLU = false>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_kernel(GemmArgs a) {
  constexpr int ROW = 272;                                   // padded LDS row of a 256-byte K block
  constexpr int KB = 128;                                    // elements of K per block
  constexpr int NT = NW * 64;                                // threads
  constexpr int AP = (MT * 256 + NT - 1) / NT;               // activation chunks per thread and K block
  constexpr bool A_EVEN = (MT * 256) % NT == 0;
  static_assert(!GLU || (!RAGGED && RB == 1), "GLU: dense form only");
  static_assert(NW == 4 || GLU, "other workgroup sizes: GLU form only");
  __shared__ __attribute__((aligned(16))) uint8_t s_a[2][MT * 16 * ROW];
  __shared__ __attribute__((aligned(16))) uint8_t s_w[NW][2][16 * ROW];
  __shared__ int s_last;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, g4 = lane >> 4;
  int grp, t0, R, row_base;                                  // group; first row of this block inside the group; rows of the group; the group's first row
  if constexpr (RAGGED) {
    if (static_cast<int>(blockIdx.y) >= a.tile_start[a.G]) return;
    int m0, m_end;
    gemm_locate_tile(a, static_cast<int>(blockIdx.y), MT * 16, grp, m0, m_end);
    row_base = m0;                                           // (rows are counted from this block's first row)
    t0 = 0;
    R = min(MT * 16, m_end - m0);
  } else {
    grp = blockIdx.y / RB;
    t0 = (blockIdx.y % RB) * (MT * 16);
    R = a.uniform_rows;
    row_base = grp * R;
  }
  // wave unit: 16 output columns (GLU: 8 gate + 8 up columns).  A unit past the last one (NW not dividing the count)
  // streams the last unit again and stores nothing.
  const int units = a.N / 16;                                // (GLU: N = 2 I, a unit is 8 + 8 columns)
  const int unit_raw = static_cast<int>(blockIdx.x) * NW + wave;
  const bool unit_live = unit_raw < units;
  const int unit = unit_live ? unit_raw : units - 1;
  const int n0 = GLU ? unit * 8 : unit * 16;                 // first output column (GLU: of C [M, I])
  const int nkb = a.K / KB;
  const int slice = blockIdx.z;
  const int kb0 = static_cast<int>(static_cast<int64_t>(nkb) * slice / a.splitk);
  const int nb = static_cast<int>(static_cast<int64_t>(nkb) * (slice + 1) / a.splitk) - kb0;   // K blocks of this slice
  const T* A = static_cast<const T*>(a.A);
  const T* W = static_cast<const T*>(a.W) + static_cast<int64_t>(grp) * a.w_group;
  // weight rows: instruction sx covers rows 4 sx .. 4 sx + 3 of the wave's 16; lane (row l / 16, 16-byte chunk l % 16)
  const T* wrow[4];
#pragma unroll
  for (int sx = 0; sx < 4; ++sx) {
    const int64_t r = GLU ? (sx < 2 ? n0 + 4 * sx : a.N / 2 + n0 + 4 * (sx - 2)) + (lane >> 4) : n0 + 4 * sx + (lane >> 4);
    wrow[sx] = W + r * a.w_n + (lane & 15) * 8;
  }

  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // activation chunk p of this thread: row idx / 16 of the group, 16-byte chunk idx % 16 of the K block
  const T* arow[AP];
#pragma unroll
  for (int p = 0; p < AP; ++p) {
    const int idx = min(static_cast<int>(threadIdx.x) + NT * p, MT * 256 - 1);
    const int t = min(t0 + (idx >> 4), R - 1);
    arow[p] = A + static_cast<int64_t>(RAGGED ? row_base + t : map_row(row_base + t, a.a_rc, a.a_ml, a.a_off, a.a_mul)) * a.lda + (idx & 15) * 8;
  }
  constexpr int DEPTH = 3;
  u32x4 wreg[DEPTH + 1][4], areg[2][AP];
  auto load_w = [&](int i, u32x4 (&wr)[4]) {
#pragma unroll
    for (int sx = 0; sx < 4; ++sx)
    {
      const u32x4* src = reinterpret_cast<const u32x4*>(wrow[sx] + (kb0 + i) * KB);
      if constexpr (RB == 1) wr[sx] = __builtin_nontemporal_load(src); else wr[sx] = *src;
    }
  };
  auto store_w = [&](int buf, const u32x4 (&wr)[4]) {
#pragma unroll
    for (int sx = 0; sx < 4; ++sx)
      *reinterpret_cast<u32x4*>(&s_w[wave][buf][(4 * sx + (lane >> 4)) * ROW + (lane & 15) * 16]) = wr[sx];
  };
  auto load_a = [&](int i, u32x4 (&ar)[AP]) {
#pragma unroll
    for (int p = 0; p < AP; ++p) ar[p] = *reinterpret_cast<const u32x4*>(arow[p] + (kb0 + i) * KB);
  };
  auto store_a = [&](int buf, const u32x4 (&ar)[AP]) {
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      const int idx = threadIdx.x + NT * p;
      if (A_EVEN || idx < MT * 256) *reinterpret_cast<u32x4*>(&s_a[buf][(idx >> 4) * ROW + (idx & 15) * 16]) = ar[p];
    }
  };
  if (nb > 0) {
    load_a(0, areg[0]);
    if (nb > 1) load_a(1, areg[1]);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
      if (d < nb) load_w(d, wreg[d]);
    store_a(0, areg[0]);
    store_w(0, wreg[0]);
  }
  __syncthreads();
  auto body = [&](int i, auto RC, auto GUARD) {
    constexpr int r = decltype(RC)::value;
    constexpr bool guarded = decltype(GUARD)::value;
    if (!guarded || i + 2 < nb) { if (r & 1) load_a(i + 2, areg[1]); else load_a(i + 2, areg[0]); }
    if (!guarded || i + DEPTH < nb) load_w(i + DEPTH, wreg[(r + DEPTH) % (DEPTH + 1)]);
#pragma unroll
    for (int sx = 0; sx < 4; ++sx) {
      const u32x4 wf = *reinterpret_cast<const u32x4*>(&s_w[wave][r & 1][l15 * ROW + sx * 64 + g4 * 16]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const u32x4 af = *reinterpret_cast<const u32x4*>(&s_a[r & 1][(mt * 16 + l15) * ROW + sx * 64 + g4 * 16]);
        acc[mt] = skinny_mfma<T>::run(wf, af, acc[mt]);
      }
    }
    if (!guarded || i + 1 < nb) {
      if (r & 1) store_a(0, areg[0]); else store_a(1, areg[1]);
      store_w((r & 1) ^ 1, wreg[(r + 1) % (DEPTH + 1)]);
    }
    __syncthreads();
  };
  int i0 = 0;
  for (; i0 + 2 * DEPTH + 1 <= nb; i0 += DEPTH + 1)
    static_for<DEPTH + 1>([&](auto RC) { body(i0 + decltype(RC)::value, RC, std::false_type{}); });
  for (; i0 < nb; i0 += DEPTH + 1)
    static_for<DEPTH + 1>([&](auto RC) {
      constexpr int r = decltype(RC)::value;
      if (i0 + r < nb) body(i0 + r, RC, std::true_type{});
    });
  if (!unit_live) return;
  T* C = static_cast<T*>(a.C);
  typedef typename vec_of<T, 4>::type V4;
  if constexpr (GLU) {
    // lane (g4, l15) holds weight rows 4 g4 .. 4 g4 + 3 of the wave's 16 for token l15: g4 0, 1 = gate columns n0 + 4 g4 + e,
    // g4 2, 3 = the up columns of the same index: the partner sits 32 lanes up
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      f32x4 other;
#pragma unroll
      for (int e = 0; e < 4; ++e) other[e] = __shfl_xor(acc[mt][e], 32);
      const int t = t0 + mt * 16 + l15;
      if (g4 >= 2 || t >= R) continue;
      V4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gf = static_cast<float>(static_cast<T>(acc[mt][e])), uf = static_cast<float>(static_cast<T>(other[e]));
        const float sg = static_cast<float>(static_cast<T>(silu_f(gf)));
        o[e] = static_cast<T>(sg * uf);
      }
      *reinterpret_cast<V4*>(C + static_cast<int64_t>(map_row(row_base + t, a.c_rc, a.c_ml, a.c_off, a.c_mul)) * a.ldc + n0 + 4 * g4) = o;
    }
    return;
  }
  // lane holds rows t = mt*16 + l15 of the group, columns n0 + 4 g4 .. +3
  const int n = n0 + 4 * g4;
  const T* bias = static_cast<const T*>(a.bias);
  auto emit = [&](int t, f32x4 v) {
    V4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = static_cast<T>(v[e]);
    if (bias) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = round_with_bias<T>(v[e], bias[n + e], a.bias_fused != 0);
    }
    *reinterpret_cast<V4*>(C + static_cast<int64_t>(RAGGED ? row_base + t : map_row(row_base + t, a.c_rc, a.c_ml, a.c_off, a.c_mul)) * a.ldc + n) = o;
  };
#ifdef MOJO_HIP_BUILD_EXPERIMENTS
  if (a.splitk > 1 && a.sk_slot >= 0) {
    // raw fp32 partials of this K slice, write-through; the last slice of the tile to arrive sums all of them (splitk_combine.h)
    const long long slice_bytes = static_cast<long long>(a.slab_rows) * a.N * 4;
    const sk_rsrc_t rs = splitk_rsrc(a.slab, slice_bytes * a.splitk);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int t = t0 + mt * 16 + l15;
      if (t < R) splitk_store16(rs, slice * slice_bytes + (static_cast<long long>(t) * a.N + n) * 4, __builtin_bit_cast(u32x4, acc[mt]));
    }
    if (!splitk_arrive(a.sk_slot, static_cast<int>(blockIdx.y * gridDim.x + blockIdx.x), a.splitk, &s_last)) return;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int t = t0 + mt * 16 + l15;
      if (t >= R) continue;
      f32x4 sum = {0.f, 0.f, 0.f, 0.f};
      for (int sx = 0; sx < a.splitk; ++sx)               // slice order: the same bits whoever is last
        sum += __builtin_bit_cast(f32x4, splitk_load16(rs, sx * slice_bytes + (static_cast<long long>(t) * a.N + n) * 4));
      emit(t, sum);
    }
    return;
  }
#endif
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int t = t0 + mt * 16 + l15;
    if (t >= R) continue;
    if (a.splitk > 1) {                                  // raw fp32 partials of this K slice for the finalize kernel (dense GEMMs only: G == 1)
      *reinterpret_cast<f32x4*>(static_cast<float*>(a.slab) + (static_cast<int64_t>(slice) * a.slab_rows + t) * a.N + n) = acc[mt];
      continue;
    }
    emit(t, acc[mt]);
  }
}

// split-K finalize: C[row_map(m)][n] = round_T(sum_s slab[s][m][n]) (+ bias after the rounding), slices in index order
template <typename T>
__global__ __launch_bounds__(256) void gemm_skinny_finalize_kernel(GemmArgs a, int64_t m_total) {
  const float* slab = static_cast<const float*>(a.slab);
  const T* bias = static_cast<const T*>(a.bias);
  T* C = static_cast<T*>(a.C);
  const int64_t total = m_total * a.N / 4;
  typedef typename vec_of<T, 4>::type V4;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<int64_t>(gridDim.x) * 256) {
    const int64_t m = (i * 4) / a.N;
    const int n = static_cast<int>(i * 4 - m * a.N);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int sx = 0; sx < a.splitk; ++sx) acc += *reinterpret_cast<const f32x4*>(slab + (static_cast<int64_t>(sx) * m_total + m) * a.N + n);
    V4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = static_cast<T>(acc[e]);
    if (bias) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = round_with_bias<T>(acc[e], bias[n + e], a.bias_fused != 0);
    }
    *reinterpret_cast<V4*>(C + static_cast<int64_t>(map_row(static_cast<int>(m), a.c_rc, a.c_ml, a.c_off, a.c_mul)) * a.ldc + n) = o;
  }
}

// split-K finalize fused with MojoResidualAddRMSNorm (the consumer of a decoder layer's output / down projection):
//   x      = round_T(sum_s slab[s][m][:]) (+ bias after the rounding)      -- what gemm_skinny_finalize_kernel stores
//   sum    = round_T(x + residual)                                        -- rmsnorm.hip, same thread / row partition and the
//   normed = round_T(sum * rsqrt(mean(sum^2) + eps) * weight)                same order of the square sum: identical bits
// The product never reaches HBM as a tensor (gemm_out == nullptr) and one launch + one round trip fall away.
template <typename T, int TPR>
__global__ __launch_bounds__(256) void gemm_splitk_resnorm_kernel(GemmArgs a, int64_t rows, const T* __restrict__ residual,
                                                                  const T* __restrict__ norm_w, T* __restrict__ normed,
                                                                  T* __restrict__ summed, T* __restrict__ gemm_out, float eps) {
  constexpr int VEC = 8, CACHE = 8;
  typedef typename vec_of<T, VEC>::type V;
  constexpr int ROWS_PER_BLOCK = 256 / TPR;
  constexpr int NWV = TPR / 64;
  __shared__ float red[4];
  const int sub = threadIdx.x / TPR, tid = threadIdx.x % TPR;
  const int dim = a.N, n_vec = dim / VEC;
  const float inv_dim = 1.0f / static_cast<float>(dim);
  const float* slab = static_cast<const float*>(a.slab);
  const T* bias = static_cast<const T*>(a.bias);
  const int64_t row0 = static_cast<int64_t>(blockIdx.x) * ROWS_PER_BLOCK;
  const bool live = row0 + sub < rows;
  const int64_t row = live ? row0 + sub : rows - 1;
  V cache[CACHE];
  float ss = 0.f;
  int c = 0;
  // K-slice sums, slices in index order; four slices' loads are issued together (the loop is otherwise one round trip per slice)
  constexpr int SB = 4;
#pragma unroll
  for (int cc = 0; cc < CACHE; ++cc) {
    const int v = tid + cc * TPR;
    if (v >= n_vec) break;
    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
    for (int sx0 = 0; sx0 < a.splitk; sx0 += SB) {
      f32x4 pl[SB], ph[SB];
#pragma unroll
      for (int i = 0; i < SB; ++i) {
        const int sx = min(sx0 + i, a.splitk - 1);
        const float* p = slab + (static_cast<int64_t>(sx) * rows + row) * dim + v * VEC;
        pl[i] = *reinterpret_cast<const f32x4*>(p);
        ph[i] = *reinterpret_cast<const f32x4*>(p + 4);
      }
#pragma unroll
      for (int i = 0; i < SB; ++i)
        if (sx0 + i < a.splitk) { lo += pl[i]; hi += ph[i]; }
    }
    V x;
#pragma unroll
    for (int j = 0; j < VEC; ++j) x[j] = static_cast<T>(j < 4 ? lo[j & 3] : hi[j & 3]);
    if (bias) {
      const V b = load_vec<T, VEC>(bias + v * VEC);
#pragma unroll
      for (int j = 0; j < VEC; ++j) x[j] = round_with_bias<T>(j < 4 ? lo[j & 3] : hi[j & 3], b[j], a.bias_fused != 0);
    }
    if (gemm_out && live) store_vec<T, VEC>(gemm_out + row * dim + v * VEC, x);
    if (residual) {
      const V y = load_vec<T, VEC>(residual + row * dim + v * VEC);
#pragma unroll
      for (int j = 0; j < VEC; ++j) x[j] = static_cast<T>(static_cast<float>(x[j]) + static_cast<float>(y[j]));
      if (summed && live) store_vec<T, VEC>(summed + row * dim + v * VEC, x);
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float f = static_cast<float>(x[j]);
      ss += f * f;
    }
    cache[cc] = x;
    c = cc + 1;
  }
  ss = wave_sum(ss);
  if constexpr (NWV > 1) {
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[wave] = ss;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NWV; ++i) t += red[sub * NWV + i];
    ss = t;
  }
  const float rstd = rsqrtf(ss * inv_dim + eps);
#pragma unroll
  for (int cc = 0; cc < CACHE; ++cc) {
    const int v = tid + cc * TPR;
    if (cc >= c) break;
    const V w = load_vec<T, VEC>(norm_w + v * VEC);
    V o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) o[j] = static_cast<T>(static_cast<float>(cache[cc][j]) * rstd * static_cast<float>(w[j]));
    if (live) store_vec<T, VEC>(normed + row * dim + v * VEC, o);
  }
}

bool gemm_splitk_resnorm_ok(const GemmArgs& a, int dtype, const void* residual, const void* norm_w, const void* normed, const void* summed) {
  if (dtype != MOJO_BF16 && dtype != MOJO_F16) return false;
  if (!(gemm_skinny_mask() & SKINNY_RESNORM)) return false;
  return a.splitk > 1 && a.slab && a.N % 8 == 0 && a.N <= 16384 && a.c_rc == 0 && aligned_to(norm_w, 16) && aligned_to(normed, 16) &&
         (!residual || aligned_to(residual, 16)) && (!summed || aligned_to(summed, 16)) && (!a.bias || aligned_to(a.bias, 16)) &&
         (!a.C || aligned_to(a.C, 16));
}

int launch_gemm_splitk_resnorm(const GemmArgs& a, int dtype, int64_t m_total, const void* residual, const void* norm_weight,
                               void* normed, void* summed, float eps, hipStream_t s) {
  MOJO_REQUIRE(gemm_splitk_resnorm_ok(a, dtype, residual, norm_weight, normed, summed), MOJO_EUNSUPPORTED, "gemm split-K -> residual RMSNorm: preconditions not met");
  const int n_vec = a.N / 8;
  // the row partition of rmsnorm.hip's launch_rms (16-byte vectors, rms_threads_per_row).  (Its long-row form
  // re-reads what 4 cached vectors per thread do not hold; 8 cached vectors cover every row this kernel accepts.)
#define RESNORM(TY, TPR_, BLOCKS)                                                                                            \
  hipLaunchKernelGGL((gemm_splitk_resnorm_kernel<TY, TPR_>), dim3(static_cast<unsigned>(BLOCKS)), dim3(256), 0, s, a, m_total, \
                     static_cast<const TY*>(residual), static_cast<const TY*>(norm_weight), static_cast<TY*>(normed),        \
                     static_cast<TY*>(summed), static_cast<TY*>(a.C), eps)
#define RESNORM_T(TY)                                                                              \
  do {                                                                                             \
    const int tpr = rms_threads_per_row(m_total, n_vec);                                           \
    if (tpr == 64) RESNORM(TY, 64, ceil_div(m_total, 4));                                          \
    else if (tpr == 128) RESNORM(TY, 128, ceil_div(m_total, 2));                                   \
    else RESNORM(TY, 256, m_total);                                                                \
  } while (0)
  if (dtype == MOJO_BF16) RESNORM_T(bf16_t); else RESNORM_T(f16_t);
#undef RESNORM_T
#undef RESNORM
  MOJO_CHECK_LAUNCH("gemm(split-K -> residual RMSNorm)");
  note_launch("gemm_skinny:splitk->resnorm");
  return MOJO_OK;
}

// Cut K so that the weight stream covers the chip: two workgroups fit a CU (70 KiB of LDS each), so up to 512 are resident;
// the cost of a split is (rounds of 256 workgroups) / slices, and the smallest split with the least cost wins (fewer
// partial slabs to write and sum) — the rule of quant_skinny_splitk.  Measured against every forced split
// (scripts/probes/skinny_split_sweep.py, MOJO_HIP_GEMM_SPLITK=<n>): within 5 % of the best split on
// the decode shapes of a Llama-3-8B layer; all splits between 2 and 8 lie within ~10 % of each other.
int gemm_skinny_splitk(int64_t m, int64_t k, int64_t n, int64_t groups) {
  if (groups != 1 || m > 128 || k % 128 != 0 || n % 64 != 0) return 1;
  const int64_t tiles = n / 64, nkb = k / 128;
  const int forced = static_cast<int>(MOJO_SWITCH("MOJO_HIP_GEMM_SPLITK", 0));
  if (forced > 0) return static_cast<int>(forced > nkb ? nkb : forced);
  int64_t best = 1;
  double best_cost = static_cast<double>((tiles + 255) / 256);
  for (int64_t sk = 2; sk <= 16 && sk <= nkb / 4 && tiles * sk <= 512; ++sk) {
    const double cost = static_cast<double>((tiles * sk + 255) / 256) / static_cast<double>(sk);
    if (cost < best_cost - 1e-9) { best = sk; best_cost = cost; }
  }
  return static_cast<int>(best);
}

bool gemm_skinny_ok(const GemmArgs& a, int dtype) {
  if (dtype != MOJO_BF16 && dtype != MOJO_F16) return false;
  if (!(gemm_skinny_mask() & SKINNY_UNIFORM)) return false;
  return a.uniform_rows > 0 && a.uniform_rows <= 128 && a.w_k == 1 && a.K % 128 == 0 && a.N % 64 == 0 && a.lda % 8 == 0 &&
         a.w_n % 8 == 0 && a.w_group % 8 == 0 && a.ldc % 4 == 0 && (a.splitk == 1 || (a.G == 1 && a.slab)) && aligned_to(a.A, 16) && aligned_to(a.W, 16) &&
         aligned_to(a.C, 8);
}

// Ragged groups whose rows average at most 64 per group (the experts of a decode step): the product is the weight stream of
// the groups that have rows, which the 256-row tile kernel reads at ~4 TB/s (it pads every group to 256 rows; measured 500 us
// for 40 experts x 50 MB).  Needs prefix arrays built for tile height 64.  MOJO_HIP_GEMM_SKINNY without bit 2 disables.
bool gemm_skinny_ragged_ok(const GemmArgs& a, int dtype, int64_t m_total) {
  if (dtype != MOJO_BF16 && dtype != MOJO_F16) return false;
  if (!(gemm_skinny_mask() & SKINNY_RAGGED)) return false;
  return a.uniform_rows == 0 && a.G >= 2 && m_total > 0 && m_total <= static_cast<int64_t>(64) * a.G && a.w_k == 1 && a.K % 128 == 0 &&
         a.N % 64 == 0 && a.lda % 8 == 0 && a.w_n % 8 == 0 && a.w_group % 8 == 0 && a.ldc % 4 == 0 && a.splitk == 1 && !a.glu &&
         a.a_rc == 0 && a.c_rc == 0 && !a.bias && aligned_to(a.A, 16) && aligned_to(a.W, 16) && aligned_to(a.C, 8);
}

int launch_gemm_skinny_ragged(const GemmArgs& a, int dtype, int64_t m_total, hipStream_t s) {
  MOJO_REQUIRE(gemm_skinny_ragged_ok(a, dtype, m_total), MOJO_EUNSUPPORTED, "gemm_skinny(ragged): preconditions not met");
  // sum over groups of ceil(rows / 64) <= G + m_total / 64: the grid's y extent; blocks past the real count exit at once
  const dim3 grid(static_cast<unsigned>(a.N / 64), static_cast<unsigned>(a.G + m_total / 64), 1u);
  if (dtype == MOJO_BF16) hipLaunchKernelGGL((gemm_skinny_kernel<bf16_t, 4, 1, true>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((gemm_skinny_kernel<f16_t, 4, 1, true>), grid, dim3(256), 0, s, a);
  MOJO_CHECK_LAUNCH("gemm_skinny(ragged)");
  note_launch("gemm_skinny:ragged");
  return MOJO_OK;
}

int launch_gemm_skinny(const GemmArgs& a_in, int dtype, hipStream_t s) {
  MOJO_REQUIRE(gemm_skinny_ok(a_in, dtype), MOJO_EUNSUPPORTED, "gemm_skinny: preconditions not met");
  GemmArgs a = a_in;
  const int mt = (a.uniform_rows + 15) / 16;
  a.sk_slot = -1;
#ifdef MOJO_HIP_BUILD_EXPERIMENTS
  if (a.splitk > 1)                                      // K slices combined by the last one to arrive, inside this launch
    a.sk_slot = splitk_take_slot(static_cast<int64_t>(a.N / 64) * a.G * (mt > 4 ? 2 : 1));
#endif
  const dim3 grid(static_cast<unsigned>(a.N / 64), static_cast<unsigned>(a.G * (mt > 4 ? 2 : 1)), static_cast<unsigned>(a.splitk));
#define SKINNY(TY, MT_, RB_) hipLaunchKernelGGL((gemm_skinny_kernel<TY, MT_, RB_>), grid, dim3(256), 0, s, a)
#define SKINNY_MT(TY)                                                                          \
  do {                                                                                         \
    if (mt <= 1) SKINNY(TY, 1, 1); else if (mt <= 2) SKINNY(TY, 2, 1); else if (mt <= 4) SKINNY(TY, 4, 1); else SKINNY(TY, 4, 2); \
  } while (0)
  if (dtype == MOJO_BF16) SKINNY_MT(bf16_t); else SKINNY_MT(f16_t);
#undef SKINNY_MT
#undef SKINNY
  MOJO_CHECK_LAUNCH("gemm_skinny");
  note_launch("gemm_skinny:uniform:splitk%d", a.splitk);
  if (a.splitk > 1 && a.sk_slot < 0 && !a.defer_finalize) return launch_gemm_splitk_finalize(a, dtype, a.uniform_rows, s);
  return MOJO_OK;
}

// ---- dense GEMM + SwiGLU in one launch (decode-sized M, [2 I, K] weights) --------------------------------------------------
bool gemm_skinny_glu_ok(const GemmArgs& a, int dtype) {
  if (dtype != MOJO_BF16 && dtype != MOJO_F16) return false;
  if (!(gemm_skinny_mask() & SKINNY_GLU)) return false;
  return a.G == 1 && a.uniform_rows > 0 && a.uniform_rows <= 64 && a.w_k == 1 && a.K % 128 == 0 && a.N % 16 == 0 && a.lda % 8 == 0 &&
         a.w_n % 8 == 0 && a.ldc % 4 == 0 && a.splitk == 1 && !a.bias && aligned_to(a.A, 16) && aligned_to(a.W, 16) && aligned_to(a.C, 8);
}

// Waves per workgroup: the count whose workgroups deal the wave units (8 + 8 weight rows each) most evenly over 256 CUs —
// the stream is per-CU bound, so the launch takes as long as the CU with the most units.  Ties: fewer waves.
int gemm_skinny_glu_waves(int64_t inter) {
  if (const int v = static_cast<int>(MOJO_SWITCH("MOJO_HIP_GEMM_WAVES", 0)); v >= 4 && v <= 8) return v;
  const int64_t units = inter / 8;
  int best = 4;
  int64_t best_cost = -1;
  for (int nw = 4; nw <= 8; ++nw) {
    const int64_t cost = ceil_div(ceil_div(units, nw), 256) * nw;
    if (best_cost < 0 || cost < best_cost) { best = nw; best_cost = cost; }
  }
  return best;
}

int launch_gemm_skinny_glu(const GemmArgs& a, int dtype, hipStream_t s) {
  MOJO_REQUIRE(gemm_skinny_glu_ok(a, dtype), MOJO_EUNSUPPORTED, "gemm_skinny(glu): preconditions not met");
  const int mt = (a.uniform_rows + 15) / 16;
  const int nw = gemm_skinny_glu_waves(a.N / 2);
  const dim3 grid(static_cast<unsigned>(ceil_div(a.N / 16, nw)), 1u, 1u);
#define GLU_K(TY, MT_, NW_) hipLaunchKernelGGL((gemm_skinny_kernel<TY, MT_, 1, false, NW_, true>), grid, dim3(NW_ * 64), 0, s, a)
#define GLU_NW(TY, MT_)                                                                                   \
  do {                                                                                                    \
    switch (nw) { case 4: GLU_K(TY, MT_, 4); break; case 5: GLU_K(TY, MT_, 5); break; case 6: GLU_K(TY, MT_, 6); break; \
                  case 7: GLU_K(TY, MT_, 7); break; default: GLU_K(TY, MT_, 8); break; }                  \
  } while (0)
#define GLU_MT(TY)                                                                                        \
  do { if (mt <= 1) GLU_NW(TY, 1); else if (mt <= 2) GLU_NW(TY, 2); else GLU_NW(TY, 4); } while (0)
  if (dtype == MOJO_BF16) GLU_MT(bf16_t); else GLU_MT(f16_t);
#undef GLU_MT
#undef GLU_NW
#undef GLU_K
  MOJO_CHECK_LAUNCH("gemm_skinny(glu)");
  note_launch("gemm_skinny:glu:waves%d", nw);
  return MOJO_OK;
}

// C = round(sum over the K slices of slab[s][m][n]) (+ bias): shared by the decode-sized kernel above and the 256x256 tile
// kernel's dense split (gemm_api.hip); a.slab holds [splitk][m_total][N] fp32, N % 4 == 0.
int launch_gemm_splitk_finalize(const GemmArgs& a, int dtype, int64_t m_total, hipStream_t s) {
  MOJO_REQUIRE((dtype == MOJO_BF16 || dtype == MOJO_F16) && a.slab && a.N % 4 == 0, MOJO_EUNSUPPORTED, "gemm split-K finalize: preconditions not met");
  int64_t blocks = ceil_div(m_total * a.N / 4, 256);
  if (blocks > 2048) blocks = 2048;
  if (dtype == MOJO_BF16) hipLaunchKernelGGL(gemm_skinny_finalize_kernel<bf16_t>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, a, m_total);
  else hipLaunchKernelGGL(gemm_skinny_finalize_kernel<f16_t>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, a, m_total);
  MOJO_CHECK_LAUNCH("gemm(split-K finalize)");
  return MOJO_OK;
}

}  // namespace mojo
