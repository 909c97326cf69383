// Generic grouped GEMM: correctness path for shapes/dtypes outside the MFMA kernel's preconditions
// (fp32 inputs, K not a multiple of 64, rows not 16-byte aligned...).  64x64 output tile per 256-thread
// block, 16-deep K steps staged through LDS as fp32, 4x4 outputs per thread, fp32 accumulation.
#include "gemm.h"

namespace mojo {

// One 256-thread block: exclusive prefix sums of the per-group row counts (row_start) and of the per-group
// tile counts (tile_start).  Counts are clamped so that no row beyond m_total is ever addressed.
// Rows behind the last group (sum(counts) < m_total) belong to no product: the golden does not return them
// (core/operators/gemm.py:111-117 concatenates the groups), the caller's [m_total, N] buffer reads as zeros there.  The GEMM
// launch never touches these rows, so no ordering issue.  Workgroups 1 .. gridDim.x - 1 do this: each sums the counts itself and
// zeroes its share of the tail, 16 bytes per lane where the row allows.  (Until round 5 the ONE prefix workgroup zeroed the tail
// two bytes at a time — "normally there are none" — and a padded buffer cost milliseconds: the second slice of an MLA prefill
// batch, 920 MB of tail, 223 ms.)
template <typename IdxT>
__device__ void zero_group_tail(const IdxT* counts, int G, long long m_total, GemmTail tail) {
  __shared__ long long part[256];
  long long sum = 0;
  for (int g = threadIdx.x; g < G; g += 256) {
    const long long c = static_cast<long long>(counts[g]);
    sum += c > 0 ? c : 0;
  }
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
    __syncthreads();
  }
  const long long e = part[0] > m_total ? m_total : part[0];
  if (e >= m_total) return;
  char* base = static_cast<char*>(tail.C);
  const long long workers = static_cast<long long>(gridDim.x - 1) * 256, me = static_cast<long long>(blockIdx.x - 1) * 256 + threadIdx.x;
  const bool wide = (reinterpret_cast<uintptr_t>(base) & 15) == 0 && tail.ld_bytes % 16 == 0;
  const long long vec_per_row = wide ? tail.row_bytes / 16 : 0, rest = (tail.row_bytes - vec_per_row * 16) / 2;
  const long long per_row = vec_per_row + rest;
  for (long long i = me; i < (m_total - e) * per_row; i += workers) {
    const long long r = e + i / per_row, c = i % per_row;
    char* row = base + r * tail.ld_bytes;
    if (c < vec_per_row) *reinterpret_cast<i32x4*>(row + c * 16) = i32x4{0, 0, 0, 0};
    else *reinterpret_cast<uint16_t*>(row + vec_per_row * 16 + (c - vec_per_row) * 2) = 0;
  }
}

template <typename IdxT>
__global__ __launch_bounds__(256) void prefix_kernel(const IdxT* counts, int G, int bm, long long m_total,
                                                     int32_t* row_start, int32_t* tile_start, GemmTail tail) {
  if (blockIdx.x > 0) {                              // (only launched with a tail to look after)
    zero_group_tail(counts, G, m_total, tail);
    return;
  }
  __shared__ long long s_rows[256];
  __shared__ long long s_tiles[256];
  __shared__ long long carry[2];
  const int tid = threadIdx.x;
  if (tid == 0) { carry[0] = 0; carry[1] = 0; }
  __syncthreads();
  for (int base = 0; base < G; base += 256) {
    const int g = base + tid;
    long long c = 0;
    if (g < G) {
      c = static_cast<long long>(counts[g]);
      if (c < 0) c = 0;
    }
    s_rows[tid] = c;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {                 // inclusive scan of the raw counts
      const long long v = tid >= off ? s_rows[tid - off] : 0;
      __syncthreads();
      s_rows[tid] += v;
      __syncthreads();
    }
    const long long r0 = carry[0];
    // clamp against m_total: group g covers [min(start, M), min(end, M))
    long long end = r0 + s_rows[tid], start = end - c;
    if (end > m_total) end = m_total;
    if (start > m_total) start = m_total;
    const long long rows = end - start;
    s_tiles[tid] = (rows + bm - 1) / bm;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
      const long long v = tid >= off ? s_tiles[tid - off] : 0;
      __syncthreads();
      s_tiles[tid] += v;
      __syncthreads();
    }
    if (g < G) {
      row_start[g] = static_cast<int32_t>(start);
      tile_start[g] = static_cast<int32_t>(carry[1] + s_tiles[tid] - (rows + bm - 1) / bm);
    }
    __syncthreads();
    if (tid == 255) {
      long long e = r0 + s_rows[255];
      carry[0] = e;                                            // un-clamped running sum keeps later starts consistent
      carry[1] += s_tiles[255];
    }
    __syncthreads();
  }
  const long long e = carry[0] > m_total ? m_total : carry[0];
  if (tid == 0) {
    row_start[G] = static_cast<int32_t>(e);
    tile_start[G] = static_cast<int32_t>(carry[1]);
  }
}

int launch_group_prefix(const void* counts, int counts_are_i64, int G, int bm, int64_t m_total, int32_t* row_start,
                        int32_t* tile_start, hipStream_t s, GemmTail tail) {
  // workgroup 0: the prefix arrays; the others: the tail's zeros, one per 256 KiB of the whole buffer (they exit at once when
  // the counts add up to m_total)
  int64_t zero_blocks = 0;
  if (tail.C) {
    zero_blocks = ceil_div(m_total * tail.row_bytes, 256 << 10);
    zero_blocks = zero_blocks < 1 ? 1 : zero_blocks > 1024 ? 1024 : zero_blocks;
  }
  const dim3 grid(static_cast<unsigned>(1 + zero_blocks));
  if (counts_are_i64)
    hipLaunchKernelGGL(prefix_kernel<int64_t>, grid, dim3(256), 0, s, static_cast<const int64_t*>(counts), G, bm,
                       static_cast<long long>(m_total), row_start, tile_start, tail);
  else
    hipLaunchKernelGGL(prefix_kernel<int32_t>, grid, dim3(256), 0, s, static_cast<const int32_t*>(counts), G, bm,
                       static_cast<long long>(m_total), row_start, tile_start, tail);
  MOJO_CHECK_LAUNCH("group_prefix");
  return MOJO_OK;
}

constexpr int GT = 64;   // tile edge
constexpr int GK = 16;   // k step

template <typename T>
__global__ __launch_bounds__(256) void gemm_generic_kernel(GemmArgs a) {
  __shared__ float sa[GK][GT + 4];
  __shared__ float sb[GK][GT + 4];
  const int n_tiles = (a.N + GT - 1) / GT;
  const int total = gemm_m_tiles(a, GT) * n_tiles;
  for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
    const int mi = tile / n_tiles, ni = tile - mi * n_tiles;
    int g, m0, m_end;
    gemm_locate_tile(a, mi, GT, g, m0, m_end);
    const int n0 = ni * GT;
    const T* A = static_cast<const T*>(a.A);
    const T* W = static_cast<const T*>(a.W) + static_cast<int64_t>(g) * a.w_group;
    const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < a.K; k0 += GK) {
      for (int i = threadIdx.x; i < GT * GK; i += 256) {
        const int r = i / GK, kk = i % GK;
        const int m = m0 + r, k = k0 + kk;
        sa[kk][r] = (m < m_end && k < a.K) ? elt<T>::to_f(A[static_cast<int64_t>(map_row(m, a.a_rc, a.a_ml, a.a_off, a.a_mul)) * a.lda + k]) : 0.f;
        const int kk2 = i / GT, c = i % GT;
        const int n = n0 + c, k2 = k0 + kk2;
        sb[kk2][c] = (n < a.N && k2 < a.K) ? elt<T>::to_f(W[static_cast<int64_t>(k2) * a.w_k + static_cast<int64_t>(n) * a.w_n]) : 0.f;
      }
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < GK; ++kk) {
        float av[4], bv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) av[i] = sa[kk][ty * 4 + i];
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = sb[kk][tx * 4 + j];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
      }
      __syncthreads();
    }
    T* C = static_cast<T*>(a.C);
    const T* bias = static_cast<const T*>(a.bias);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + ty * 4 + i;
      if (m >= m_end) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + tx * 4 + j;
        if (n >= a.N) continue;
        T o = elt<T>::from_f(acc[i][j]);
        if (bias) o = round_with_bias<T>(acc[i][j], bias[n], a.bias_fused != 0);
        C[static_cast<int64_t>(map_row(m, a.c_rc, a.c_ml, a.c_off, a.c_mul)) * a.ldc + n] = o;
      }
    }
  }
}

int launch_gemm_generic(const GemmArgs& a, int dtype, int64_t m_total, hipStream_t s) {
  const int64_t n_tiles = ceil_div(a.N, GT);
  int64_t blocks = (ceil_div(m_total, GT) + a.G) * n_tiles;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  switch (dtype) {
    case MOJO_F32: hipLaunchKernelGGL(gemm_generic_kernel<float>, dim3(blocks), dim3(256), 0, s, a); break;
    case MOJO_F16: hipLaunchKernelGGL(gemm_generic_kernel<f16_t>, dim3(blocks), dim3(256), 0, s, a); break;
    case MOJO_BF16: hipLaunchKernelGGL(gemm_generic_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, a); break;
    default: MOJO_REQUIRE(false, MOJO_EUNSUPPORTED, "gemm: dtype %d not supported", dtype);
  }
  MOJO_CHECK_LAUNCH("gemm_generic");
  note_launch("gemm_generic");
  return MOJO_OK;
}

}  // namespace mojo
