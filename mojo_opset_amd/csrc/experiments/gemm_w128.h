// EXPERIMENT (built only with MOJO_HIP_BUILD_EXPERIMENTS=1, taken with MOJO_HIP_GEMM_W128=1):
// the 256x256 GroupGemm tile on FOUR waves at one wave per SIMD, 128x128 of C per wave (256 accumulator registers).
//
// Why: the shipped kernel (gemm256_core.h) gives each of its 8 waves a 128x64 piece, so a K-tile costs
// (128 + 64) * 128 B of LDS reads per 64 MFMAs per wave; a 128x128 piece reads (128 + 128) * 128 B per 128 MFMAs — a third
// fewer LDS bytes per FLOP (rule 28's second lever: energy per MFMA, which sets the clock the chip holds under this load).
//
// Structure: K is walked in HALF K-tiles of 64 bytes per row ("k-halves": one 16x16x32 step).  A k-half of the workgroup
// tile is 32 blocks of 1 KiB (16 rows x 64 B; blocks 0-15 A rows, 16-31 W rows), lane-linear as LDS-DMA writes it, with
// the shipped kernel's source-side swizzle (rows 8-15 swap their 32-byte halves) so that a fragment is one conflict-free
// ds_read_b128.  Four k-half buffers (128 KiB) form a ring: in phase p a wave runs the 64 MFMAs of k-half p from
// registers, reads the fragments of k-half p + 1 into the other register set, and requests k-half p + 4 into the buffer
// k-half p just left.  One barrier per phase (behind `vmcnt(16)`: k-half p + 1 has landed; and `lgkmcnt(0)`: the fragment
// reads of this phase are done, so the buffer can be overwritten).  [N,K] weights, 16-bit operands, no GLU / split-K / bias.
#pragma once
#include "../gemm256_core.h"

namespace mojo {
namespace w128 {

constexpr int BM = 256, BN = 256;
constexpr int KH_BYTES = 64;                       // bytes of K per row per k-half
constexpr int BLK = 1024;                          // one 16-row block of a k-half
constexpr int KH_BUF = 32 * BLK;                   // 32 KiB
constexpr int NBUF = 4;
constexpr int LDS_BYTES = NBUF * KH_BUF;           // 128 KiB
constexpr int PANEL = 4;

typedef g256::lds_char lds_char;
typedef g256::frag16 frag16;

template <typename P, typename Epi, int ABL = 0 /* timing-only ablations: 1 no barrier, 2 no LDS-DMA, 3 no fragment reads in the K loop */,
          int AHEAD = 4 /* k-halves between a request and its phase: 4 (three in flight across a barrier) or 3 (two) */>
__global__ __launch_bounds__(256, 1) void gemm_w128_kernel(GemmArgs a, Epi epi) {
  typedef typename P::acc_t acc_t;
  constexpr int EB = P::EB;
  static_assert(EB == 2, "16-bit operands");
  constexpr int KH = KH_BYTES / EB;                  // elements of K per k-half (32)
  extern __shared__ __attribute__((aligned(1024))) char smem_generic[];
  lds_char* smem = (lds_char*)smem_generic;

  const int n_tiles = (a.N + BN - 1) / BN;
  const int m_tiles = gemm_m_tiles(a, BM);
  const int total = m_tiles * n_tiles;
  const int bid = blockIdx.x;
  if (bid >= total) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // the shipped kernel's tile order: every XCD walks its own run of panels of PANEL n-tiles, m-tile by m-tile
  int tile;
  {
    const int q = total >> 3, r = total & 7, x = bid & 7, i = bid >> 3;
    tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  int mi, ni;
  {
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
  int g, m0, m_end;
  gemm_locate_tile(a, mi, BM, g, m0, m_end);
  const int n0 = ni * BN;
  const int nkh = a.K / KH;

  // ---- this lane's global sources: 4 A blocks and 4 W blocks per k-half (block = 16 rows x 64 B, one LDS-DMA each) ----
  const char* srcA[4];
  const char* srcW[4];
  {
    const int row = lane >> 2;
    const int chunk = (lane & 3) ^ ((row & 8) ? 2 : 0);
    const char* A = static_cast<const char*>(a.A);
    const char* W = static_cast<const char*>(a.W) + static_cast<int64_t>(g) * a.w_group * EB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int m = m0 + (wave * 4 + i) * 16 + row;
      if (m >= m_end) m = m_end - 1;                 // rows past the group: re-read a valid row, never stored
      srcA[i] = A + (static_cast<int64_t>(map_row(m, a.a_rc, a.a_ml, a.a_off, a.a_mul)) * a.lda) * EB + chunk * 16;
      int n = n0 + (wave * 4 + i) * 16 + row;
      if (n >= a.N) n = a.N - 1;
      srcW[i] = W + (static_cast<int64_t>(n) * a.w_n) * EB + chunk * 16;
    }
  }
  // piece i (0..7) of the request for k-half kh -> buffer kh % NBUF: 8 LDS-DMA instructions per wave and k-half
  auto stage_piece = [&](int kh, int i) {
    lds_char* dst = smem + (kh & (NBUF - 1)) * KH_BUF + (wave * 4) * BLK;
    if (kh >= nkh) kh = nkh - 1;                      // past the end: the same count of loads (uniform vmcnt), into a buffer nobody reads again
    const int64_t off = static_cast<int64_t>(kh) * KH_BYTES;
    if (i < 4) g256::glds16(srcA[i] + off, dst + i * BLK);
    else g256::glds16(srcW[i - 4] + off, dst + (16 + i - 4) * BLK);
  };
  auto stage = [&](int kh) {
#pragma unroll
    for (int i = 0; i < 8; ++i) stage_piece(kh, i);
  };

  // fragment of block b: lane reads row l & 15, 16-byte chunk (l >> 4) ^ (2 if row >= 8).  The reads are asm (hipcc would
  // sink them to their first use, next phase) and are retired by a wait that names every destination.
  const unsigned smem_u32 = static_cast<unsigned>(reinterpret_cast<size_t>(smem));
  const unsigned frag_a = smem_u32 + (wm * 8) * BLK + (lane & 15) * 64 + (((lane >> 4) ^ ((lane & 8) ? 2 : 0)) * 16);
  const unsigned frag_w = frag_a + (16 + wn * 8 - wm * 8) * BLK;
  struct Frags { frag16 fa[8], fw[8]; };
  auto read_pair = [&](Frags& f, int kh, int i) {      // A block i and W block i of k-half kh
    const unsigned o = (kh & (NBUF - 1)) * KH_BUF;
#define W128_RD(I)                                                                                                     \
    asm volatile("ds_read_b128 %0, %2 offset:" #I "*1024\n\tds_read_b128 %1, %3 offset:" #I "*1024"                    \
                 : "=&v"(f.fa[I]), "=&v"(f.fw[I]) : "v"(frag_a + o), "v"(frag_w + o) : "memory")
    switch (i) {
      case 0: W128_RD(0); break; case 1: W128_RD(1); break; case 2: W128_RD(2); break; case 3: W128_RD(3); break;
      case 4: W128_RD(4); break; case 5: W128_RD(5); break; case 6: W128_RD(6); break; default: W128_RD(7); break;
    }
#undef W128_RD
  };
  auto retire = [&](Frags& f, const char*) {};
  (void)retire;

  acc_t acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = acc_t{0, 0, 0, 0};
  // every accumulator is named by the wait-state statements below: hipcc must not move its own AGPR writes / reads across them
#define W128_NAME_ACC(STR)                                                                                             \
  _Pragma("unroll") for (int i_ = 0; i_ < 8; i_ += 2)                                                                  \
    asm volatile(STR : "+a"(acc[i_][0]), "+a"(acc[i_][1]), "+a"(acc[i_][2]), "+a"(acc[i_][3]), "+a"(acc[i_][4]),      \
                       "+a"(acc[i_][5]), "+a"(acc[i_][6]), "+a"(acc[i_][7]), "+a"(acc[i_ + 1][0]), "+a"(acc[i_ + 1][1]), \
                       "+a"(acc[i_ + 1][2]), "+a"(acc[i_ + 1][3]), "+a"(acc[i_ + 1][4]), "+a"(acc[i_ + 1][5]),         \
                       "+a"(acc[i_ + 1][6]), "+a"(acc[i_ + 1][7]))
  W128_NAME_ACC("s_nop 7");

  // one phase: the 64 MFMAs of `cur` in 8 groups (A block s against the 8 W blocks); behind group s the reads of the
  // fragment pair s of k-half p + 1 and piece s of the request for k-half p + 4
  auto phase = [&](const Frags& cur, Frags& nxt, int p) {
    const int pn = p + 1 < nkh ? p + 1 : p;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if constexpr (std::is_same<typename P::elem, bf16_t>::value)
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[s][j]) : "v"(cur.fw[j]), "v"(cur.fa[s]));
        else
          asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[s][j]) : "v"(cur.fw[j]), "v"(cur.fa[s]));
      }
      if (s < 4) {                                       // all memory instructions in the first half: the phase-end waits find them done
        if constexpr (ABL != 3) {
          read_pair(nxt, pn, 2 * s);
          read_pair(nxt, pn, 2 * s + 1);
        }
        if constexpr (ABL != 2) {
          stage_piece(p + AHEAD, 2 * s);
          stage_piece(p + AHEAD, 2 * s + 1);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // k-halves p + 3, p + 4 may stay in flight: p + 2 has landed; this phase's fragment reads are done, so buffer
    // (p + 1) % 4 may be overwritten by the next phase's request
    asm volatile("s_waitcnt vmcnt(%16)\n\ts_waitcnt lgkmcnt(0)"
                 : "+v"(nxt.fa[0]), "+v"(nxt.fa[1]), "+v"(nxt.fa[2]), "+v"(nxt.fa[3]), "+v"(nxt.fa[4]), "+v"(nxt.fa[5]), "+v"(nxt.fa[6]), "+v"(nxt.fa[7]),
                   "+v"(nxt.fw[0]), "+v"(nxt.fw[1]), "+v"(nxt.fw[2]), "+v"(nxt.fw[3]), "+v"(nxt.fw[4]), "+v"(nxt.fw[5]), "+v"(nxt.fw[6]), "+v"(nxt.fw[7])
                 : "n"((AHEAD - 2) * 8) : "memory");
    if constexpr (ABL != 1) __builtin_amdgcn_s_barrier();
  };

  stage(0); stage(1); stage(2);
  if constexpr (AHEAD == 4) stage(3);
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"((AHEAD - 1) * 8) : "memory");          // k-half 0 has landed
  __builtin_amdgcn_s_barrier();
  Frags f0, f1;
#pragma unroll
  for (int i = 0; i < 8; ++i) read_pair(f0, 0, i);
  asm volatile("s_waitcnt vmcnt(%16)\n\ts_waitcnt lgkmcnt(0)"   // k-half 1 landed; k-half 0's buffer is free
               : "+v"(f0.fa[0]), "+v"(f0.fa[1]), "+v"(f0.fa[2]), "+v"(f0.fa[3]), "+v"(f0.fa[4]), "+v"(f0.fa[5]), "+v"(f0.fa[6]), "+v"(f0.fa[7]),
                 "+v"(f0.fw[0]), "+v"(f0.fw[1]), "+v"(f0.fw[2]), "+v"(f0.fw[3]), "+v"(f0.fw[4]), "+v"(f0.fw[5]), "+v"(f0.fw[6]), "+v"(f0.fw[7])
               : "n"((AHEAD - 2) * 8) : "memory");
  __builtin_amdgcn_s_barrier();
  int p = 0;
  for (; p + 1 < nkh; p += 2) {
    phase(f0, f1, p);
    phase(f1, f0, p + 1);
  }
  if (p < nkh) phase(f0, f1, p);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  W128_NAME_ACC("s_nop 15\n\ts_nop 15");
#undef W128_NAME_ACC

  // ---- epilogue: a lane owns row (lane & 15) of an A block and 4 consecutive columns of a W block ------------------
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + (wm * 8 + i) * 16 + (lane & 15);
    if (m >= m_end) continue;
    epi.row_begin(m);
    const int mc = map_row(m, a.c_rc, a.c_ml, a.c_off, a.c_mul);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int n = n0 + (wn * 8 + j) * 16 + (lane >> 4) * 4;
      if (n >= a.N) continue;
      epi.store(mc, n, a.N, acc[i][j]);
    }
  }
}

template <typename P, typename Epi, int ABL = 0, int AHEAD = 4>
inline int gemm_w128_launch(const GemmArgs& a, const Epi& epi, int64_t m_total, hipStream_t s) {
  const int64_t n_tiles = ceil_div(a.N, BN);
  const int64_t blocks = (ceil_div(m_total, BM) + a.G) * n_tiles;
  MOJO_REQUIRE(blocks < (1LL << 31), MOJO_EUNSUPPORTED, "gemm_w128: grid too large");
  auto* fn = gemm_w128_kernel<P, Epi, ABL, AHEAD>;
  static std::atomic<uint64_t> attr_set{0};
  if (first_call_on_device(attr_set)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  hipLaunchKernelGGL(fn, dim3(static_cast<unsigned>(blocks)), dim3(256), LDS_BYTES, s, a, epi);
  MOJO_CHECK_LAUNCH("gemm_w128");
  return MOJO_OK;
}

}  // namespace w128
}  // namespace mojo
