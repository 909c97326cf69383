// mla512_oct_kernel: the r = 512 / rope = 64 latent-attention kernel with TWO waves per SIMD.
// Included by mla_attn.hip (shares MlaArgs, mla_mfma); the default for pages below 16 tokens, MOJO_HIP_MLA_KERNEL=oct elsewhere.
//
// Same paired-halves idea as mla512_pair_kernel, but a wave owns 16 heads instead of 32: wave = (hq, half) with
// hq = wave & 3 (heads 16*hq .. +15 of the workgroup's 64) and half = wave >> 2 (QK^T over keys 32*half .. +31, PV over
// latent dims 256*half .. +255); the partner is wave ^ 4.  That cuts the register footprint to 72 (query) + 64 (O^T)
// + ~90 working registers, so eight waves fit a CU — two per SIMD — and one wave's LDS round trips, exchange barriers
// and LDS-DMA issue stalls are covered by the other's MFMAs.  The price is that an LDS fragment feeds one MFMA instead
// of two (4 x (36 + 32) KiB -> 8 x (36 + 32) KiB of LDS reads per tile).
#pragma once

namespace mojo {

// K-fragment batch B of the paired kernel's QK^T: 4 reads, linear index n = 4B + i -> key tile n / 18, k-step n % 18
template <int B, int I = 0>
__device__ __forceinline__ void mlap_k_issue(u32x4 (&dst)[4], const unsigned (&kav)[4], const unsigned (&kbv)[2]) {
  constexpr int n = 4 * B + I, tt = n / 18, ks = n % 18;
  if constexpr (ks < 16)
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[I]) : "v"(kav[ks & 3]), "i"(tt * 16384 + (ks >> 2) * 256) : "memory");
  else
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[I]) : "v"(kbv[ks - 16]), "i"(tt * 2048) : "memory");
  if constexpr (I + 1 < 4) mlap_k_issue<B, I + 1>(dst, kav, kbv);
}


#ifdef MLA_STAMPS
// In-kernel stamps (build with MOJO_HIP_EXTRA_CXXFLAGS=-DMLA_STAMPS): lane i of `tacc` accumulates the cycles between stamp
// i-1 and stamp i of the tile loop; read back with mojo_hip_debug_mla_stamps.  Timing tool only.
__device__ unsigned g_mla_stamps[1024 * 8 * 16];
#define MLA_STAMP(i)                                                             \
  do {                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                           \
    const unsigned long long t_ = __builtin_readcyclecounter();                  \
    const unsigned d_ = static_cast<unsigned>(t_ - t_prev);                      \
    t_prev = t_;                                                                 \
    tacc += (lane == (i)) ? d_ : 0u;                                             \
    __builtin_amdgcn_sched_barrier(0);                                           \
  } while (0)
#else
#define MLA_STAMP(i)
#endif

template <typename T>
__global__ __launch_bounds__(512, 1) void mla512_oct_kernel(MlaArgs a) {
  typedef typename mla_mfma<T>::frag frag;
  constexpr int R = 512, NK = 18, WAVES = 8, HPB = 64;
  constexpr int A_BYTES = MLA_KEYS * 1024, B_BYTES = MLA_KEYS * 128, TILE = A_BYTES + B_BYTES;   // 72 KiB
  constexpr int TABLE_ENTRIES = 1024;
  constexpr int TABLE_OFF = 2 * TILE, MAX_OFF = TABLE_OFF + TABLE_ENTRIES * 4, P_OFF = MAX_OFF + WAVES * 16 * 4;
  extern __shared__ __attribute__((aligned(1024))) char smem_generic[];
  lds_m* smem = (lds_m*)smem_generic;
  const unsigned smem_u32 = static_cast<unsigned>(reinterpret_cast<size_t>(smem));

  const int tile = blockIdx.x % a.n_tiles, hb = blockIdx.x / a.n_tiles, split = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hq = wave & 3, half = wave >> 2;          // 16-head group, key / latent half
  const int grp = lane >> 4, l15 = lane & 15;

  int b, n_vis;
  if (a.cu_q == nullptr) {
    b = tile;
    n_vis = a.seq_lens[b];
  } else {
    if (tile < a.cu_q[0] || tile >= a.cu_q[a.batch]) return;
    int lo = 0, hi = a.batch;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (a.cu_q[mid] <= tile) lo = mid; else hi = mid;
    }
    b = lo;
    const int q_len = a.cu_q[b + 1] - a.cu_q[b];
    const int kv_len = a.cu_kv ? a.cu_kv[b + 1] - a.cu_kv[b] : q_len;
    n_vis = min(kv_len, kv_len - q_len + (tile - a.cu_q[b]) + 1);
  }
  const int32_t* table = a.tables + static_cast<int64_t>(b) * a.table_stride;
  if (n_vis > 0) {                                    // the golden stops at the first negative page id
    int p1 = (n_vis + a.page - 1) / a.page;
    int fn = -1;
    if (p1 > a.max_pages) { fn = a.max_pages; p1 = a.max_pages; }
    for (int base = 0; base < p1; base += 64) {
      const int idx = base + lane;
      const int v = idx < p1 ? table[idx] : 0;
      const unsigned long long neg = __ballot(v < 0);
      if (neg) { fn = base + __builtin_ctzll(neg); break; }
    }
    if (fn >= 0) n_vis = min(n_vis, fn * a.page);
  }
  const int k_begin = split * a.split_keys;
  const int k_end = min(n_vis, k_begin + a.split_keys);
  const int n_kt = k_end > k_begin ? (k_end - k_begin + MLA_KEYS - 1) / MLA_KEYS : 0;

  int* s_table = reinterpret_cast<int*>(smem_generic + TABLE_OFF);
  const unsigned table_u32 = smem_u32 + TABLE_OFF;
  int win_base = 0;
  auto fill_window = [&](int p0) {
    for (int i = threadIdx.x; i < TABLE_ENTRIES; i += 512) s_table[i] = (p0 + i < a.max_pages) ? table[p0 + i] : -1;
    win_base = p0;
    __syncthreads();
  };
  auto page_of = [&](int key) { return key >> a.page_shift; };
  fill_window(page_of(k_begin));

  const int head0 = hb * HPB + hq * 16;                 // first head of this wave
  const bool active = head0 < a.heads;                  // identical for the two waves of a pair
  const int head = min(head0 + l15, a.heads - 1);

  frag qf[NK];
  {
    const int64_t qrow = static_cast<int64_t>(tile) * a.heads + head;
    const T* qp = static_cast<const T*>(a.q_lat) + qrow * a.q_stride + grp * 8;
    const T* qr = static_cast<const T*>(a.q_rope) + qrow * a.q_rope_stride + grp * 8;
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) qf[ks] = *reinterpret_cast<const frag*>(ks * 32 < R ? qp + ks * 32 : qr + (ks * 32 - R));
  }
  // retire the query loads where the compiler's wait-count pass can see it
  __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0)

  // ---- staging: 9 LDS-DMA pieces per wave per tile (8 c_kv rows of 1 KiB + one k_pe block of 8 rows) ---------------
  const T* ckv = static_cast<const T*>(a.ckv);
  const T* kpe = static_cast<const T*>(a.kpe);
  struct StagePlan { int addr_lo, addr_hi, buf; const T* pe; };
  constexpr int PIECES = MLA_KEYS / WAVES + 1;
  auto stage_prep = [&](int kt, int buf) {
    StagePlan sp;
    sp.buf = buf;
    const int k_first = k_begin + kt * MLA_KEYS;
    {
      const int p_last = page_of(min(k_first + MLA_KEYS - 1, k_end - 1));
      if (p_last >= win_base + TABLE_ENTRIES) {
        __syncthreads();
        fill_window(page_of(k_first));
      }
    }
    const int mask = a.page - 1;
    int my_phys, phys_b;
    const int key_l = min(k_first + wave + 8 * (lane & 7), k_end - 1);      // lane i < 8 owns c_kv row wave + 8i
    const int row_b = wave * 8 + (lane >> 3);
    const int key_b = min(k_first + row_b, k_end - 1);
    asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(my_phys), "=&v"(phys_b)
                 : "v"(table_u32 + 4 * ((key_l >> a.page_shift) - win_base)), "v"(table_u32 + 4 * ((key_b >> a.page_shift) - win_base))
                 : "memory");
    my_phys = max(my_phys, 0);
    phys_b = max(phys_b, 0);
    const uint64_t row_addr = reinterpret_cast<uint64_t>(ckv) +
                              2 * (static_cast<uint64_t>(static_cast<unsigned>(my_phys)) * static_cast<uint64_t>(a.ckv_blk) +
                                   static_cast<uint64_t>(static_cast<unsigned>(key_l & mask)) * static_cast<uint64_t>(a.ckv_tok));
    sp.addr_lo = static_cast<int>(row_addr);
    sp.addr_hi = static_cast<int>(row_addr >> 32);
    sp.pe = kpe + static_cast<int64_t>(phys_b) * a.kpe_blk + static_cast<int64_t>(key_b & mask) * a.kpe_tok + ((lane & 7) ^ (row_b & 7)) * 8;
    return sp;
  };
  const int cs_row = (lane ^ (wave << 1)) * 16;             // rows wave + 8i: (row & 7) = wave
  auto stage_piece = [&](const StagePlan& sp, int i) {      // i: compile-time constant after unrolling
    lds_m* ta = smem + sp.buf * TILE;
    if (i < MLA_KEYS / WAVES) {
      const int row = i * WAVES + wave;
      const unsigned lo = __builtin_amdgcn_readlane(sp.addr_lo, i), hi = __builtin_amdgcn_readlane(sp.addr_hi, i);
      const char* src = reinterpret_cast<const char*>((static_cast<uint64_t>(hi) << 32) | lo);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + cs_row),
                                       (__attribute__((address_space(3))) void*)(ta + row * 1024), 16, 0, 0);
    } else {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sp.pe,
                                       (__attribute__((address_space(3))) void*)(ta + A_BYTES + wave * 1024), 16, 0, 0);
    }
  };

  // ---- per-lane read offsets (bytes inside a tile buffer) ---------------------------------------------------------
  const int x2 = (l15 & 7) << 1;
  unsigned ka[4], kb2[2];
#pragma unroll
  for (int v = 0; v < 4; ++v) ka[v] = half * 32768 + l15 * 1024 + (((4 * v) | grp) ^ x2) * 16;   // + tt*16384 + (ks>>2)*256
#pragma unroll
  for (int v = 0; v < 2; ++v) kb2[v] = A_BYTES + half * 4096 + l15 * 128 + (((4 * v + grp) ^ (l15 & 7)) * 16);   // + tt*2048
  const int tq = l15 >> 2, tp = l15 & 3;
  const int trow = 4 * grp + tq;
  unsigned tr8[8];                                       // d tile dtl of this wave's half: v = dtl & 7, + (dtl>>3)*256
#pragma unroll
  for (int v = 0; v < 8; ++v) tr8[v] = half * 512 + trow * 1024 + ((((2 * v) | (tp >> 1)) ^ ((trow & 7) << 1)) * 16) + (tp & 1) * 8;
  const unsigned own_off = half * 32768, oth_off = (half ^ 1) * 32768;     // key rows of the own / the partner's QK^T half

  // exchange slots (the partner of wave w is w ^ 4: same heads, other half)
  const unsigned max_mine = smem_u32 + MAX_OFF + (wave * 16 + l15) * 4;
  const unsigned max_other = smem_u32 + MAX_OFF + ((wave ^ 4) * 16 + l15) * 4;
  const unsigned p_mine = smem_u32 + P_OFF + (wave * 64 + lane) * 16;
  const unsigned p_other = smem_u32 + P_OFF + ((wave ^ 4) * 64 + lane) * 16;

  f32x4 o[16];
#pragma unroll
  for (int dt = 0; dt < 16; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, lsum = 0.f;

  if (n_kt > 0) {
    const StagePlan sp0 = stage_prep(0, 0);
#pragma unroll
    for (int i = 0; i < PIECES; ++i) stage_piece(sp0, i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
#ifdef MLA_STAMPS
  unsigned tacc = 0;
  unsigned long long t_prev = __builtin_readcyclecounter();
#endif
  for (int kt = 0; kt < n_kt; ++kt) {
    const int buf = kt & 1;
    const bool prefetch = kt + 1 < n_kt;
    StagePlan sp{};
    MLA_STAMP(0);
    if (prefetch) sp = stage_prep(kt + 1, buf ^ 1);
    MLA_STAMP(1);
    if (prefetch && !active) {
#pragma unroll
      for (int i = 0; i < PIECES; ++i) stage_piece(sp, i);
    }
    const unsigned vt = smem_u32 + buf * TILE;
    frag pf[2];                                          // [own / partner]
    float alpha = 1.f, ref_own = -INFINITY, ps = 0.f;
    if (active) {
      // ---- S^T (own 32 keys x 16 heads) = K_lat Q_lat^T: 36 fragment reads in 9 batches, one DMA piece behind each ----
      f32x4 s[2];
      s[0] = f32x4{0.f, 0.f, 0.f, 0.f};
      s[1] = f32x4{0.f, 0.f, 0.f, 0.f};
      {
        unsigned kav[4], kbv[2];
#pragma unroll
        for (int v = 0; v < 4; ++v) kav[v] = vt + ka[v];
#pragma unroll
        for (int v = 0; v < 2; ++v) kbv[v] = vt + kb2[v];
        u32x4 kr[2][4];
        mlap_k_issue<0>(kr[0], kav, kbv);
        static_for<9>([&](auto BC) {
          constexpr int B = decltype(BC)::value;
          if constexpr (B + 1 < 9) mlap_k_issue<B + 1>(kr[(B + 1) & 1], kav, kbv);
          u32x4 (&cur)[4] = kr[B & 1];
          if constexpr (B + 1 < 9)
            asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]) : : "memory");
          else
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]) : : "memory");
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            constexpr int n0 = 4 * B;
            const int n = n0 + i, tt = n / 18, ks = n % 18;
            s[tt] = mla_mfma<T>::run(__builtin_bit_cast(frag, cur[i]), qf[ks], s[tt]);
          }
          if (prefetch) stage_piece(sp, B);
        });
      }
      // wait states between the last MFMA and the first vector read of the scores: on the last tile of a split (nothing to
      // stage) with no key to mask that read follows the MFMA across two branches, where hipcc places none (mla512_pp.h)
      asm volatile("s_nop 7\n\ts_nop 7" : "+v"(s[0]), "+v"(s[1]));
      MLA_STAMP(2);
      // ---- own-half maxima and probabilities -------------------------------------------------------------------------
      const int key0 = k_begin + kt * MLA_KEYS + 32 * half + 4 * grp;
      if (k_begin + (kt + 1) * MLA_KEYS > k_end) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (key0 + 16 * tt + r >= k_end) s[tt][r] = -INFINITY;
      }
      // The pair no longer agrees on a reference maximum BEFORE the exponentials (that took an LDS round trip, a workgroup
      // barrier and a second round trip per tile): each wave takes the maximum of its own 32 keys in the vector unit
      // (lane-row swaps, no ds_bpermute), keeps or advances its OWN reference (lazily, by > 2^8), and publishes reference
      // and probabilities together.  Behind the one barrier both waves form the same joint reference max(own, partner's);
      // the side whose reference was lower rescales its bf16 probabilities (a wave-uniform branch, taken on the first tile
      // and on the rare tiles where a maximum moves).
      float mx = -INFINITY;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[tt][r]);
      mx = xor_max_16_32(mx);
      ref_own = m;
      if ((mx - m) * a.scale_log2 > 8.0f) ref_own = mx;       // m = -inf: any finite score; NaN (-inf - -inf): keep
      {
        const float ms = (ref_own == -INFINITY ? 0.f : ref_own) * a.scale_log2;
        frag f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p0 = fast_exp2(fmaf(s[0][r], a.scale_log2, -ms));
          const float p1 = fast_exp2(fmaf(s[1][r], a.scale_log2, -ms));
          ps += p0 + p1;
          f[r] = static_cast<T>(p0);
          f[4 + r] = static_cast<T>(p1);
        }
        pf[0] = f;
      }
      {
        const u32x4 w0 = __builtin_bit_cast(u32x4, pf[0]);
        asm volatile("ds_write_b128 %0, %1\n\tds_write_b32 %2, %3\n\ts_waitcnt lgkmcnt(0)"
                     : : "v"(p_mine), "v"(w0), "v"(max_mine), "v"(ref_own) : "memory");
      }
      MLA_STAMP(5);
    }
    __builtin_amdgcn_s_barrier();                                                          // (2) references and probabilities visible
    MLA_STAMP(6);
    if (active) {
      {
        u32x4 r0;
        float ref_oth;
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(r0), "=&v"(ref_oth) : "v"(p_other), "v"(max_other) : "memory");
        pf[1] = __builtin_bit_cast(frag, r0);
        const float ref = fmaxf(ref_own, ref_oth);
        const float f_own = ref_own == ref ? 1.f : fast_exp2((ref_own - ref) * a.scale_log2);   // -inf vs finite: 0
        const float f_oth = ref_oth == ref ? 1.f : fast_exp2((ref_oth - ref) * a.scale_log2);
        if (!__all(f_own == 1.f)) {
          ps *= f_own;
#pragma unroll
          for (int e = 0; e < 8; ++e) pf[0][e] = static_cast<T>(static_cast<float>(pf[0][e]) * f_own);
        }
        if (!__all(f_oth == 1.f)) {
#pragma unroll
          for (int e = 0; e < 8; ++e) pf[1][e] = static_cast<T>(static_cast<float>(pf[1][e]) * f_oth);
        }
        alpha = m == ref ? 1.f : fast_exp2((m - ref) * a.scale_log2);                            // m = -inf: 0 (O and the sum are 0)
        m = ref;
        lsum = lsum * alpha + ps;
      }
      if (!__all(alpha == 1.0f)) {
#pragma unroll
        for (int dt = 0; dt < 16; ++dt) o[dt] *= alpha;
      }
      MLA_STAMP(7);
      // ---- O^T (own 256 d) += C_kv^T P^T over all 64 keys: 8 batches of 8 transposed reads, double-buffered ----------
      // batch J = d tiles 2J, 2J+1;   regs [i*4 + which*2 + {lo,hi}], which = 0: own key half, 1: partner's
      const unsigned v_own = vt + own_off, v_oth = vt + oth_off;
      s16x4 va[8], vb[8];
#define MLAO_ISSUE(dst, J)                                                                                             \
      asm volatile(                                                                                                   \
          "ds_read_b64_tr_b16 %0, %8 offset:%12\n\tds_read_b64_tr_b16 %1, %8 offset:%13\n\t"                          \
          "ds_read_b64_tr_b16 %2, %9 offset:%12\n\tds_read_b64_tr_b16 %3, %9 offset:%13\n\t"                          \
          "ds_read_b64_tr_b16 %4, %10 offset:%12\n\tds_read_b64_tr_b16 %5, %10 offset:%13\n\t"                        \
          "ds_read_b64_tr_b16 %6, %11 offset:%12\n\tds_read_b64_tr_b16 %7, %11 offset:%13"                            \
          : "=&v"(dst[0]), "=&v"(dst[1]), "=&v"(dst[2]), "=&v"(dst[3]), "=&v"(dst[4]), "=&v"(dst[5]), "=&v"(dst[6]),  \
            "=&v"(dst[7])                                                                                             \
          : "v"(v_own + tr8[((J) * 2 + 0) & 7]), "v"(v_oth + tr8[((J) * 2 + 0) & 7]), "v"(v_own + tr8[((J) * 2 + 1) & 7]), \
            "v"(v_oth + tr8[((J) * 2 + 1) & 7]), "i"(((J) >> 2) * 256), "i"(((J) >> 2) * 256 + 16384)                 \
          : "memory")
#define MLAO_RETIRE(dst, N)                                                                                            \
      asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                        \
                   : "+v"(dst[0]), "+v"(dst[1]), "+v"(dst[2]), "+v"(dst[3]), "+v"(dst[4]), "+v"(dst[5]), "+v"(dst[6]),  \
                     "+v"(dst[7])                                                                                       \
                   : : "memory")
#define MLAO_PV(src, J)                                                                                                \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int w = 0; w < 2; ++w) {                   \
        const s16x4 lo = src[i * 4 + w * 2], hi = src[i * 4 + w * 2 + 1];                                              \
        const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};                                   \
        o[(J) * 2 + i] = mla_mfma<T>::run(__builtin_bit_cast(frag, both), pf[w], o[(J) * 2 + i]);                      \
      }
#define MLAO_STEP2(J)                                                                                                  \
      MLAO_ISSUE(vb, (J) + 1); MLAO_RETIRE(va, 8); MLAO_PV(va, (J));                                                   \
      MLAO_ISSUE(va, (J) + 2); MLAO_RETIRE(vb, 8); MLAO_PV(vb, (J) + 1);
      MLAO_ISSUE(va, 0);
      MLAO_STEP2(0) MLAO_STEP2(2) MLAO_STEP2(4)
      MLAO_ISSUE(vb, 7); MLAO_RETIRE(va, 8); MLAO_PV(va, 6);
      MLAO_RETIRE(vb, 0); MLAO_PV(vb, 7);
#undef MLAO_STEP2
#undef MLAO_ISSUE
#undef MLAO_RETIRE
#undef MLAO_PV
      MLA_STAMP(8);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MLA_STAMP(9);
    __builtin_amdgcn_s_barrier();                                                          // (3) next tile staged, this one free
    MLA_STAMP(10);
  }
#ifdef MLA_STAMPS
  {
    const unsigned wg = blockIdx.y * gridDim.x + blockIdx.x;
    if (wg < 1024) {
      if (lane < 15) g_mla_stamps[(wg * 8 + wave) * 16 + lane] = tacc;
      if (lane == 15) g_mla_stamps[(wg * 8 + wave) * 16 + 15] = static_cast<unsigned>(n_kt);
    }
  }
#endif

  // ---- epilogue: the pair's row sums are added through the maxima slots, each wave stores its 256 d ------------------
  float lt = lsum;
  lt += __shfl_xor(lt, 16);
  lt += __shfl_xor(lt, 32);
  asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : : "v"(max_mine), "v"(lt) : "memory");
  __builtin_amdgcn_s_barrier();
  {
    float l0;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(l0) : "v"(max_other) : "memory");
    lt += l0;
  }
  if (!active || head0 + l15 >= a.heads) return;
  typedef typename vec_of<T, 4>::type V4;
  const int d0 = half * 256 + grp * 4;
  if (a.n_splits == 1) {
    float den = lt;
    float w = 1.f;
    const float ml2 = m * a.scale_log2;
    if (a.sink) {
      const float sk = a.sink[head] * 1.4426950408889634f;
      const float M = fmaxf(ml2, sk);
      w = (m == -INFINITY) ? 0.f : fast_exp2(ml2 - M);
      den = lt * w + fast_exp2(sk - M);
    }
    const float inv = den > 0.f ? w / den : 0.f;
    T* dst = static_cast<T*>(a.o_lat) + (static_cast<int64_t>(tile) * a.heads + head) * R + d0;
#pragma unroll
    for (int dt = 0; dt < 16; ++dt) {
      V4 ov;
#pragma unroll
      for (int r = 0; r < 4; ++r) ov[r] = static_cast<T>(o[dt][r] * inv);
      *reinterpret_cast<V4*>(dst + dt * 16) = ov;
    }
  } else {
    const int64_t slot = (static_cast<int64_t>(tile) * a.n_splits + split) * a.heads + head;
    float* po = a.part_o + slot * R + d0;
#pragma unroll
    for (int dt = 0; dt < 16; ++dt) *reinterpret_cast<f32x4*>(po + dt * 16) = o[dt];
    if (grp == 0 && half == 0) {
      a.part_ml[slot * 2] = m * a.scale_log2;
      a.part_ml[slot * 2 + 1] = lt;
    }
  }
}

constexpr int MLA512_OCT_LDS = 2 * (MLA_KEYS * 1024 + MLA_KEYS * 128) + 1024 * 4 + 8 * 16 * 4 + 8 * 64 * 16;

}  // namespace mojo
