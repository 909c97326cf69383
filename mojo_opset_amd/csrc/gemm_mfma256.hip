// Grouped / dense GEMM on MFMA for gfx950: 256x256 output tile, BK = 64, 8 waves (2 x 4), bf16/fp16
// inputs, fp32 accumulation.
//
// Structure (after the "256^2 8-phase" recipe of the CDNA4 programming guide, re-derived here):
//   * LDS = 2 K-tile buffers x 4 half-tiles (A rows 0-127 / 128-255, W cols 0-127 / 128-255) x 16 KiB.
//   * global -> LDS by direct-to-LDS loads (global_load_lds_dwordx4, 1 KiB per wave instruction).  The
//     LDS image is lane-linear, so bank-conflict avoidance lives in the per-lane SOURCE address:
//       - K-major operands (A always; W when stored [N,K]) use 16-row x 32-k sub-tiles (1 KiB) with the
//         two 32-byte halves of rows 8-15 swapped ("st_16x32"); a fragment is one ds_read_b128 per lane.
//       - W stored [K,N] uses an [k/8][n/16][8][16] image with k-rows 0-3 <-> 4-7 swapped in odd k-blocks;
//         fragments come from ds_read_b64_tr_b16 (hardware transpose), one glds = 8 k-rows x 128 B.
//   * a K-tile is 4 phases; each phase = {issue one half-tile of a future K-tile, read this phase's
//     fragments, 16 MFMAs of one 64x32 C-quadrant per wave, one barrier}.  Loads are retired with a
//     COUNTED s_waitcnt vmcnt(6) once per K-tile, so three half-tiles (48 KiB per CU) stay in flight
//     across barriers; the wait sits before the last barrier of a K-tile and the data is first read in
//     the next phase.
//   * MFMA operands are swapped (W fragment as "A", A fragment as "B") so that a lane owns 4 consecutive
//     output columns of one row -> 8-byte stores.
//   * blocks are mapped to tiles through an XCD-aware, panel-major order: the 32 blocks that run
//     concurrently on one XCD cover 4 m-tiles x 8 n-tiles and share their operands in that XCD's L2.
//
// Algorithmic FLOPs per launch: 2 * M * K * N.   Bound: MFMA (bf16 dense peak ~2.5 PFLOP/s).
#include "gemm.h"

namespace mojo {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HALF_BYTES = 128 * BK * 2;          // 16 KiB
constexpr int KTILE_BYTES = 4 * HALF_BYTES;       // A0 A1 W0 W1
constexpr int LDS_BYTES = 2 * KTILE_BYTES;        // 128 KiB
constexpr int PANEL = 8;                          // n-tiles per panel

typedef __attribute__((address_space(3))) char lds_char;

template <typename T> struct mfma_t;
template <> struct mfma_t<bf16_t> {
  typedef bf16x8 frag;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct mfma_t<f16_t> {
  typedef f16x8 frag;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

__device__ __forceinline__ void glds16(const void* gsrc, lds_char* dst_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc),
                                   (__attribute__((address_space(3))) void*)(dst_wave_base), 16, 0, 0);
}

template <typename T, bool W_NMAJOR /* true: W is [K,N] (n contiguous); false: [N,K] */>
__global__ __launch_bounds__(512, 2) void gemm_mfma256_kernel(GemmArgs a) {
  typedef typename mfma_t<T>::frag frag;
  extern __shared__ __attribute__((aligned(1024))) char smem_generic[];
  lds_char* smem = (lds_char*)smem_generic;

  // ---- which tile ---------------------------------------------------------------------------
  const int n_tiles = (a.N + BN - 1) / BN;
  const int m_tiles = a.tile_start[a.G];
  const int total = m_tiles * n_tiles;
  const int bid = blockIdx.x;
  if (bid >= total) return;
  int tile;
  {  // bijective XCD remap: blocks b, b+8, ... share an XCD; give each XCD one contiguous run of tiles
    const int q = total >> 3, r = total & 7, x = bid & 7, i = bid >> 3;
    tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  int mi, ni;
  {  // panel-major order: panels of PANEL n-tiles; inside a panel m-tile by m-tile
    const int full_panels = n_tiles / PANEL, rem = n_tiles - full_panels * PANEL;
    const int in_full = full_panels * m_tiles * PANEL;
    if (tile < in_full) {
      const int p = tile / (m_tiles * PANEL), t = tile - p * (m_tiles * PANEL);
      mi = t / PANEL;
      ni = p * PANEL + (t - mi * PANEL);
    } else {
      const int t = tile - in_full;
      mi = t / rem;
      ni = full_panels * PANEL + (t - mi * rem);
    }
  }
  int g;
  {
    int lo = 0, hi = a.G;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (a.tile_start[mid] <= mi) lo = mid; else hi = mid;
    }
    g = lo;
  }
  const int m0 = a.row_start[g] + (mi - a.tile_start[g]) * BM;
  const int m_end = a.row_start[g + 1];              // exclusive; m0 < m_end by construction
  const int n0 = ni * BN;
  const int nkt = a.K / BK;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  // ---- staging: per-lane global source pointers (advance by BK per K-tile) -----------------------
  // K-major half-tile h: wave w fills row-block w (16 rows) with two glds (k-blocks 0,1).
  //   lane l -> row l/4, 16-byte chunk (l%4) ^ (2 if row >= 8)           [st_16x32 on the source side]
  const T* A = static_cast<const T*>(a.A);
  const T* W = static_cast<const T*>(a.W) + static_cast<int64_t>(g) * a.w_group;
  const T* srcA[2];
  {
    const int row = lane >> 2;
    const int chunk = (lane & 3) ^ ((row & 8) ? 2 : 0);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int m = m0 + h * 128 + wave * 16 + row;
      if (m >= m_end) m = m_end - 1;                 // rows past the group: re-read a valid row, never stored
      srcA[h] = A + static_cast<int64_t>(m) * a.lda + chunk * 8;
    }
  }
  const T* srcW[2];
  int64_t w_step;                                   // element advance per K-tile
  if constexpr (!W_NMAJOR) {
    const int row = lane >> 2;
    const int chunk = (lane & 3) ^ ((row & 8) ? 2 : 0);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int n = n0 + h * 128 + wave * 16 + row;
      if (n >= a.N) n = a.N - 1;
      srcW[h] = W + static_cast<int64_t>(n) * a.w_n + chunk * 8;
    }
    w_step = BK;
  } else {
    // [k/8][n/16][8][16] image: wave w fills k-block w with two glds (n-blocks 0-3, 4-7).
    //   lane l -> n-block l/16, stored row (l%16)/2, columns (l%2)*8..+8; odd k-blocks hold rows 4-7 first
    const int rr = ((lane & 15) >> 1) ^ ((wave & 1) ? 4 : 0);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int n = n0 + h * 128 + (lane >> 4) * 16 + (lane & 1) * 8;     // second glds adds 64 columns
      srcW[h] = W + static_cast<int64_t>(wave * 8 + rr) * a.w_k + n;
    }
    w_step = static_cast<int64_t>(BK) * a.w_k;
  }
  // columns of the second W glds in N-major mode (clamped so a partial n-tile never reads past the row)
  int w2_off[2] = {0, 0};
  if constexpr (W_NMAJOR) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int n = n0 + h * 128 + (lane >> 4) * 16 + (lane & 1) * 8;
      const int n_a = n > a.N - 8 ? a.N - 8 : n;
      const int n_b = n + 64 > a.N - 8 ? a.N - 8 : n + 64;
      srcW[h] += n_a - n;
      w2_off[h] = n_b - n_a;
    }
  }

  // stage half-tile `which` (0:A0 1:A1 2:W0 3:W1) of K-tile kt into buffer buf
  auto stage = [&](int which, int kt, int buf) {
    if (kt >= nkt) kt = nkt - 1;                    // keep the vmcnt bookkeeping uniform at the tail
    lds_char* dst = smem + buf * KTILE_BYTES + which * HALF_BYTES + wave * 2048;
    if (which < 2) {
      const T* p = srcA[which] + static_cast<int64_t>(kt) * BK;
      glds16(p, dst);
      glds16(p + 32, dst + 1024);
    } else {
      const int h = which - 2;
      const T* p = srcW[h] + static_cast<int64_t>(kt) * w_step;
      if constexpr (!W_NMAJOR) {
        glds16(p, dst);
        glds16(p + 32, dst + 1024);
      } else {
        glds16(p, dst);
        glds16(p + w2_off[h], dst + 1024);
      }
    }
  };

  // ---- fragment read offsets ----------------------------------------------------------------------
  // K-major: sub-tile (rb, ks) at (rb*2+ks)*1024; lane reads row l&15, chunk (l>>4) ^ (2 if row >= 8)
  const int kmaj_lane = (lane & 15) * 64 + (((lane >> 4) ^ ((lane & 8) ? 2 : 0)) * 16);
  // N-major W: block (kb, nb) at (kb*8+nb)*256; kb = ks*4 + (l>>4); lane 4q+p of a 16-group -> row q, cols 4p
  const int grp = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
  const int nmaj_lane0 = grp * 2048 + (qq + ((grp & 1) ? 4 : 0)) * 32 + pp * 8;   // k rows 0-3 of the block
  const int nmaj_lane1 = grp * 2048 + (qq + ((grp & 1) ? 0 : 4)) * 32 + pp * 8;   // k rows 4-7

  auto read_a = [&](frag (&fa)[4][2], int h, int buf) {
    const lds_char* base = smem + buf * KTILE_BYTES + h * HALF_BYTES + (wm * 4) * 2048 + kmaj_lane;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fa[i][ks] = *reinterpret_cast<const __attribute__((address_space(3))) frag*>(base + i * 2048 + ks * 1024);
  };
  auto read_w = [&](frag (&fw)[2][2], int h, int buf) {           // K-major W ([N,K])
    const lds_char* base = smem + buf * KTILE_BYTES + (2 + h) * HALF_BYTES + (wn * 2) * 2048 + kmaj_lane;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fw[j][ks] = *reinterpret_cast<const __attribute__((address_space(3))) frag*>(base + j * 2048 + ks * 1024);
  };
  // N-major W ([K,N]): transposed reads.  hipcc drains vmcnt(0) in front of the ds_read_tr builtin (it
  // cannot prove the read independent of the LDS-DMA writes in flight), which serialises the pipeline;
  // so the reads are issued from inline asm and retired by an explicit lgkmcnt wait that names every
  // destination register (the compiler may not touch them in between).
  struct TrRegs { s16x4 r[8]; };                                   // [j][ks][k rows 0-3 | 4-7]
  const unsigned smem_u32 = static_cast<unsigned>(reinterpret_cast<size_t>(smem));
  auto issue_w_tr = [&](TrRegs& t, int h, int buf) {
    const unsigned base = smem_u32 + buf * KTILE_BYTES + (2 + h) * HALF_BYTES + (wn * 2) * 256;
    const unsigned a0 = base + nmaj_lane0, a1 = base + nmaj_lane1;
    asm volatile(
        "ds_read_b64_tr_b16 %0, %8\n\t"
        "ds_read_b64_tr_b16 %1, %9\n\t"
        "ds_read_b64_tr_b16 %2, %8 offset:8192\n\t"
        "ds_read_b64_tr_b16 %3, %9 offset:8192\n\t"
        "ds_read_b64_tr_b16 %4, %8 offset:256\n\t"
        "ds_read_b64_tr_b16 %5, %9 offset:256\n\t"
        "ds_read_b64_tr_b16 %6, %8 offset:8448\n\t"
        "ds_read_b64_tr_b16 %7, %9 offset:8448"
        : "=v"(t.r[0]), "=v"(t.r[1]), "=v"(t.r[2]), "=v"(t.r[3]), "=v"(t.r[4]), "=v"(t.r[5]), "=v"(t.r[6]), "=v"(t.r[7])
        : "v"(a0), "v"(a1)
        : "memory");
  };
  auto retire_w_tr = [&](TrRegs& t, frag (&fw)[2][2]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(t.r[0]), "+v"(t.r[1]), "+v"(t.r[2]), "+v"(t.r[3]), "+v"(t.r[4]), "+v"(t.r[5]), "+v"(t.r[6]), "+v"(t.r[7])
                 :
                 : "memory");
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const s16x4 lo = t.r[(j * 2 + ks) * 2], hi = t.r[(j * 2 + ks) * 2 + 1];
        const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        fw[j][ks] = __builtin_bit_cast(frag, both);
      }
  };

  // acc[mt][nt]: mt = h_m*4 + i (16-row tiles of this wave), nt = h_n*2 + j (16-col tiles)
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto quadrant = [&](const frag (&fa)[4][2], const frag (&fw)[2][2], int hm, int hn) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[hm * 4 + i][hn * 2 + j] = mfma_t<T>::run(fw[j][ks], fa[i][ks], acc[hm * 4 + i][hn * 2 + j]);
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- schedule -------------------------------------------------------------------------------------
  // A K-tile is 4 phases; a phase is two segments separated by barriers:
  //     R_p : issue one half-tile of a future K-tile (2 glds) + this phase's fragment reads
  //     M_p : 16 MFMAs (one 64x32 C-quadrant), then s_waitcnt vmcnt(6)
  // Waves 4-7 (wm = 1) run ONE barrier behind waves 0-3, so on every SIMD one wave is in its M
  // segment while its partner is in R: the matrix pipe and the LDS/VMEM pipes ping-pong.
  // Hazards under that stagger (a lagging reader / stager is one segment late):
  //   WAR: a half-tile is restaged >= 2 phases after the phase that last ds_read it;
  //   RAW: the counted wait that retires a half-tile sits at the end of the phase TWO before the
  //        phase that first reads it (wait -> barrier -> barrier -> read, for either group).
  // Steady state, K-tile t in buffer b (reads: P1 A0+W0, P2 W1, P3 A1, P4 none - W0 stays in VGPRs):
  //     P1 stages W1(t+1)->b^1   P2 stages A1(t+1)->b^1   P3 stages A0(t+2)->b   P4 stages W0(t+2)->b
  // After every phase "all but the last 3 half-tiles issued" have landed, which is exactly what the
  // read two phases later needs (GroupGemm schedule table).
  stage(0, 0, 0); stage(1, 0, 0); stage(2, 0, 0); stage(3, 0, 0);
  stage(0, 1, 1); stage(2, 1, 1);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // K-tile 0 has landed
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();            // the stagger

  frag fa[4][2], fw0[2][2], fw1[2][2];

  auto seg_end = [&]() {
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  TrRegs tr;
  auto ktile = [&](int t, int buf) {
    // P1
    stage(3, t + 1, buf ^ 1);
    if constexpr (W_NMAJOR) issue_w_tr(tr, 0, buf); else read_w(fw0, 0, buf);
    read_a(fa, 0, buf);
    __builtin_amdgcn_s_barrier();
    if constexpr (W_NMAJOR) retire_w_tr(tr, fw0);
    quadrant(fa, fw0, 0, 0);
    seg_end();
    // P2
    stage(1, t + 1, buf ^ 1);
    if constexpr (W_NMAJOR) issue_w_tr(tr, 1, buf); else read_w(fw1, 1, buf);
    __builtin_amdgcn_s_barrier();
    if constexpr (W_NMAJOR) retire_w_tr(tr, fw1);
    quadrant(fa, fw1, 0, 1);
    seg_end();
    // P3
    stage(0, t + 2, buf);
    read_a(fa, 1, buf);
    __builtin_amdgcn_s_barrier();
    quadrant(fa, fw1, 1, 1);
    seg_end();
    // P4
    stage(2, t + 2, buf);
    __builtin_amdgcn_s_barrier();
    quadrant(fa, fw0, 1, 0);
    seg_end();
  };

  int t = 0;
  for (; t + 1 < nkt; t += 2) {
    ktile(t, 0);
    ktile(t + 1, 1);
  }
  if (t < nkt) ktile(t, 0);
  if (wm == 0) __builtin_amdgcn_s_barrier();            // pair the stagger barrier
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue: lane owns row (lane&15), columns 4*(lane>>4)..+3 of each 16x16 tile ---------------
  T* C = static_cast<T*>(a.C);
  const T* bias = static_cast<const T*>(a.bias);
  typedef typename vec_of<T, 4>::type V4;
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) {
    const int m = m0 + (mt >> 2) * 128 + wm * 64 + (mt & 3) * 16 + (lane & 15);
    if (m >= m_end) continue;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n = n0 + (nt >> 1) * 128 + wn * 32 + (nt & 1) * 16 + (lane >> 4) * 4;
      if (n >= a.N) continue;
      V4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = static_cast<T>(acc[mt][nt][e]);
      if (bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < a.N) o[e] = static_cast<T>(static_cast<float>(o[e]) + static_cast<float>(bias[n + e]));
      }
      T* dst = C + static_cast<int64_t>(m) * a.ldc + n;
      if (n + 4 <= a.N) {
        *reinterpret_cast<V4*>(dst) = o;
      } else {
        for (int e = 0; e < 4 && n + e < a.N; ++e) dst[e] = o[e];
      }
    }
  }
}

bool gemm_mfma256_ok(const GemmArgs& a, int dtype) {
  if (dtype != MOJO_BF16 && dtype != MOJO_F16) return false;
  if (a.K < BK || a.K % BK != 0 || a.N < 8) return false;
  if (a.lda % 8 != 0 || a.ldc % 4 != 0) return false;
  if (!aligned_to(a.A, 16) || !aligned_to(a.W, 16) || !aligned_to(a.C, 8)) return false;
  if (a.w_n == 1) {                                   // [K,N]
    if (a.w_k % 8 != 0 || a.w_group % 8 != 0 || a.N % 8 != 0) return false;
  } else if (a.w_k == 1) {                            // [N,K]
    if (a.w_n % 8 != 0 || a.w_group % 8 != 0) return false;
  } else {
    return false;
  }
  return true;
}

template <typename T>
static int launch_t(const GemmArgs& a, int64_t m_total, hipStream_t s) {
  const int64_t n_tiles = ceil_div(a.N, BN);
  const int64_t blocks = (ceil_div(m_total, BM) + a.G) * n_tiles;   // upper bound; surplus blocks exit
  MOJO_REQUIRE(blocks < (1LL << 31), MOJO_EUNSUPPORTED, "gemm: grid too large");
  static bool attr_done[2] = {false, false};
  if (a.w_n == 1) {
    auto* fn = gemm_mfma256_kernel<T, true>;
    if (!attr_done[0]) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); attr_done[0] = true; }
    hipLaunchKernelGGL(fn, dim3(static_cast<unsigned>(blocks)), dim3(512), LDS_BYTES, s, a);
  } else {
    auto* fn = gemm_mfma256_kernel<T, false>;
    if (!attr_done[1]) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); attr_done[1] = true; }
    hipLaunchKernelGGL(fn, dim3(static_cast<unsigned>(blocks)), dim3(512), LDS_BYTES, s, a);
  }
  MOJO_CHECK_LAUNCH("gemm_mfma256");
  return MOJO_OK;
}

int launch_gemm_mfma256(const GemmArgs& a, int dtype, int64_t m_total, hipStream_t s) {
  MOJO_REQUIRE(gemm_mfma256_ok(a, dtype), MOJO_EUNSUPPORTED, "gemm_mfma256: preconditions not met");
  return dtype == MOJO_BF16 ? launch_t<bf16_t>(a, m_total, s) : launch_t<f16_t>(a, m_total, s);
}

}  // namespace mojo
