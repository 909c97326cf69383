// mla512_pp_kernel: the r = 512 / rope = 64 latent-attention kernel as a PING-PONG of the two waves of a SIMD.
// Included by mla_attn.hip after mla512_oct.h (shares MlaArgs, mla_mfma and the stamp macro).
//
// mla512_oct_kernel keeps the two waves of a SIMD in lock-step: they are the two halves of one pair, meet at the exchange
// barrier of every tile and therefore reach their softmax, their exchange and their barrier waits together — its counters
// show the matrix pipe busy 0.29 and the LDS array 0.31 of the time (profiles/r2_attention_counters.json).  Here the partner
// of a wave sits on ANOTHER SIMD (wave ^ 1), waves 0-3 run one segment ahead of waves 4-7 (the SIMD-mate of wave w is w + 4),
// and the workgroup barrier between segments is what keeps them exactly one segment apart:
//
//      segment        2t                         2t + 1                      2t + 2
//      waves 0-3      QK^T(t), softmax(t)        exchange, PV(t)             QK^T(t + 1) ...
//      waves 4-7      exchange, PV(t - 1)        QK^T(t), softmax(t)         exchange, PV(t) ...
//
// so in every segment each SIMD has one wave in the QK^T + softmax chain and one in the PV chain.  Running two tiles at once
// needs more than two tile buffers, and 160 KiB of LDS hold only two 64-key tiles; tiles are therefore 32 keys (36 KiB) in a
// ring of four: tile t + 2 is requested (LDS-DMA) in segment 2t - 1, right after the last reader of its slot (PV(t - 2) of
// waves 4-7, segment 2t - 2) has passed the barrier, and must have landed before segment 2t + 4 — five segments of lead.
// A pair splits the 32 keys of a tile for QK^T (16 each: one 16 x 16 score block, 18 MFMAs) and the 512 latent dims for PV
// (256 each: 16 MFMAs over all 32 keys); the probabilities of the partner's 16 keys come through LDS lane for lane (8 bytes),
// published before the barrier that ends the QK^T segment and read after it — the exchange costs no barrier of its own.
//
// Measured (B 64, H 128, ctx 4096, scripts/probes/mla_pp_stamps.py): correct and deterministic, but SLOWER than the lock-step
// kernel — 124 us against 108 us — so it is not the default (MOJO_HIP_MLA_KERNEL=pp selects it; tests cover all three
// kernels).  A wave's own instruction stream takes ~3 490 cycles per 32-key tile: QK^T 600 (18 MFMAs = 288), mask / max / exp /
// publish 450, partner read + joint reference 560, PV 630 (16 MFMAs = 256), staging a tile (addresses, 4-5 LDS-DMA pieces,
// counted wait) 680, loop top + two barriers 560.  The chains that do not shrink with the tile — softmax, exchange,
// staging, barriers — are paid per 32 keys here and per 64 keys in the lock-step kernel (5 745 cycles per 64 keys = 2 870
// per 32): what bounds both kernels is the length of ONE wave's serial stream per key, not how well the two waves of a
// SIMD overlap, and the 160 KiB of LDS force the smaller tile on the variant that overlaps them.
#pragma once

namespace mojo {

constexpr int MLAPP_KEYS = 32;

// K-fragment batch B of QK^T: k-steps 2B, 2B + 1 (18 in all: 16 over the latent, 2 over the rope part).  Batches of two keep
// the fragment ring at 16 registers: with 72 query + 64 accumulator registers per wave the file has none to spare.
template <int B, int I = 0>
__device__ __forceinline__ void mlapp_k_issue(u32x4 (&dst)[2], const unsigned (&kav)[4], const unsigned (&kbv)[2]) {
  constexpr int ks = 2 * B + I;
  if constexpr (ks < 16)
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[I]) : "v"(kav[ks & 3]), "i"((ks >> 2) * 256) : "memory");
  else
    asm volatile("ds_read_b128 %0, %1" : "=v"(dst[I]) : "v"(kbv[ks - 16]) : "memory");
  if constexpr (I + 1 < 2) mlapp_k_issue<B, I + 1>(dst, kav, kbv);
}

template <typename T>
__global__ __launch_bounds__(512, 1) void mla512_pp_kernel(MlaArgs a) {
  typedef typename mla_mfma<T>::frag frag;
  constexpr int R = 512, NK = 18, WAVES = 8, HPB = 64, KEYS = MLAPP_KEYS, SLOTS = 4;
  constexpr int A_BYTES = KEYS * 1024, B_BYTES = KEYS * 128, SLOT = A_BYTES + B_BYTES;           // 36 KiB
  constexpr int TABLE_ENTRIES = 1024;
  constexpr int TABLE_OFF = SLOTS * SLOT, MAX_OFF = TABLE_OFF + TABLE_ENTRIES * 4, P_OFF = MAX_OFF + WAVES * 16 * 4;
  extern __shared__ __attribute__((aligned(1024))) char smem_generic[];
  lds_m* smem = (lds_m*)smem_generic;
  const unsigned smem_u32 = static_cast<unsigned>(reinterpret_cast<size_t>(smem));

  const int tile = blockIdx.x % a.n_tiles, hb = blockIdx.x / a.n_tiles, split = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int late = wave >> 2;                            // 0: waves 0-3 lead, 1: waves 4-7 run one segment behind
  const int hq = (wave >> 2) * 2 + ((wave >> 1) & 1);   // 16-head group; the partner (other half) is wave ^ 1
  const int half = wave & 1;                             // key half for QK^T, latent-dim half for PV
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
  const int n_kt = k_end > k_begin ? (k_end - k_begin + KEYS - 1) / KEYS : 0;

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
  __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0): retire the query loads where the wait-count pass sees it

  // ---- staging: per tile 4 c_kv rows of 1 KiB per wave (rows wave + 8 i) + one k_pe block of 8 rows for waves 0-3 -------
  const T* ckv = static_cast<const T*>(a.ckv);
  const T* kpe = static_cast<const T*>(a.kpe);
  struct StagePlan { int addr_lo, addr_hi; const T* pe; };
  auto stage_prep = [&](int kt) {
    StagePlan sp;
    const int k_first = k_begin + kt * KEYS;
    {
      const int p_last = page_of(min(k_first + KEYS - 1, k_end - 1));
      if (p_last >= win_base + TABLE_ENTRIES) {
        __syncthreads();
        fill_window(page_of(k_first));
      }
    }
    const int mask = a.page - 1;
    int my_phys, phys_b;
    const int key_l = min(k_first + wave + 8 * (lane & 3), k_end - 1);      // lane i < 4 owns c_kv row wave + 8i
    const int row_b = (wave & 3) * 8 + (lane >> 3);                           // k_pe rows of waves 0-3
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
  const int cs_row = (lane ^ (wave << 1)) * 16;             // rows wave + 8i: (row & 7) = wave; chunk c is kept at c ^ (2 (row & 7))
  auto stage_tile = [&](int kt) {
    const StagePlan sp = stage_prep(kt);
    lds_m* ta = smem + (kt & (SLOTS - 1)) * SLOT;
#pragma unroll
    for (int i = 0; i < KEYS / WAVES; ++i) {
      const unsigned lo = __builtin_amdgcn_readlane(sp.addr_lo, i), hi = __builtin_amdgcn_readlane(sp.addr_hi, i);
      const char* src = reinterpret_cast<const char*>((static_cast<uint64_t>(hi) << 32) | lo);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + cs_row),
                                       (__attribute__((address_space(3))) void*)(ta + (i * WAVES + wave) * 1024), 16, 0, 0);
    }
    if (wave < 4)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sp.pe,
                                       (__attribute__((address_space(3))) void*)(ta + A_BYTES + wave * 1024), 16, 0, 0);
  };
  // all but the pieces of `younger` later tiles have landed (pieces per tile: 5 for waves 0-3, 4 for waves 4-7)
  auto wait_landed = [&](int younger) {
    if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (wave < 4) {
      if (younger == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    } else {
      if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
  };

  // ---- per-lane read offsets (bytes inside a slot) -----------------------------------------------------------------------
  const int x2 = (l15 & 7) << 1;
  unsigned ka[4], kb2[2];
#pragma unroll
  for (int v = 0; v < 4; ++v) ka[v] = half * 16384 + l15 * 1024 + (((4 * v) | grp) ^ x2) * 16;       // + (ks >> 2) * 256
#pragma unroll
  for (int v = 0; v < 2; ++v) kb2[v] = A_BYTES + half * 2048 + l15 * 128 + (((4 * v + grp) ^ (l15 & 7)) * 16);
  const int tq = l15 >> 2, tp = l15 & 3;
  const int trow = 4 * grp + tq;
  // transposed reads of d tile dtl (16 latent dims) of this wave's half: row trow of a 16-row block, chunk
  // (2 v | tp >> 1) ^ 2 (trow & 7) with v = dtl & 7, + (dtl >> 3) * 256; 32 v ^ tr_sw is formed per use (registers)
  const unsigned tr_k0 = half * 512 + trow * 1024 + (tp >> 1) * 16 + (tp & 1) * 8;
  const unsigned tr_sw = (trow & 7) << 5;
  // exchange slots (the partner of wave w is w ^ 1: same heads, other half, another SIMD)
  const unsigned max_mine = smem_u32 + MAX_OFF + (wave * 16 + l15) * 4;
  const unsigned max_other = smem_u32 + MAX_OFF + ((wave ^ 1) * 16 + l15) * 4;
  const unsigned p_mine = smem_u32 + P_OFF + (wave * 64 + lane) * 8;
  const unsigned p_other = smem_u32 + P_OFF + ((wave ^ 1) * 64 + lane) * 8;

  f32x4 o[16];
#pragma unroll
  for (int dt = 0; dt < 16; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, lsum = 0.f;
  // carried from the QK^T segment of a tile to its PV segment
  typedef typename vec_of<T, 4>::type T4;
  T4 p_own;
#pragma unroll
  for (int r = 0; r < 4; ++r) p_own[r] = static_cast<T>(0.f);
  float ref_own = -INFINITY, ps = 0.f;

  if (n_kt > 0) {
    for (int t = 0; t < 3 && t < n_kt; ++t) stage_tile(t);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
#ifdef MLA_STAMPS
  unsigned tacc = 0;
  unsigned long long t_prev = __builtin_readcyclecounter();
#endif
  // Both groups run the SAME loop; waves 4-7 pass one extra barrier in front of it (and waves 0-3 one behind it), which
  // puts them one segment behind for good: the barrier after QK^T(t) of the leading waves is the barrier after PV(t - 1) of
  // the late ones.  Staging sits at the end of every ODD global segment — behind PV(t) for the leading waves, behind QK^T(t)
  // for the late ones: tile t + 3 is requested (its slot was last read by PV(t - 1) of the late waves, one segment ago) and
  // tile t + 1, which the leading waves read next, must have landed; the pieces of tiles t + 2 and t + 3 may be in flight.
  auto stage_and_wait = [&](int kt) {
    if (kt + 3 < n_kt) stage_tile(kt + 3);
    wait_landed((kt + 2 < n_kt ? 1 : 0) + (kt + 3 < n_kt ? 1 : 0));
  };
  if (late && n_kt > 0) __builtin_amdgcn_s_barrier();
  for (int kt = 0; kt < n_kt; ++kt) {
    MLA_STAMP(0);
    if (active) {
      // ---- S^T (own 16 keys x 16 heads) = K_lat Q_lat^T: 18 fragment reads in 9 batches of two ----------------------------------
      const unsigned vt = smem_u32 + (kt & (SLOTS - 1)) * SLOT;
      f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
      {
        unsigned kav[4], kbv[2];
#pragma unroll
        for (int v = 0; v < 4; ++v) kav[v] = vt + ka[v];
#pragma unroll
        for (int v = 0; v < 2; ++v) kbv[v] = vt + kb2[v];
        u32x4 kr[3][2];                                  // three batches of two in flight
        mlapp_k_issue<0>(kr[0], kav, kbv);
        mlapp_k_issue<1>(kr[1], kav, kbv);
        static_for<9>([&](auto BC) {
          constexpr int B = decltype(BC)::value;
          if constexpr (B + 2 < 9) mlapp_k_issue<B + 2>(kr[(B + 2) % 3], kav, kbv);
          u32x4 (&cur)[2] = kr[B % 3];
          if constexpr (B + 2 < 9)
            asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(cur[0]), "+v"(cur[1]) : : "memory");
          else if constexpr (B + 1 < 9)
            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(cur[0]), "+v"(cur[1]) : : "memory");
          else
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]) : : "memory");
          s = mla_mfma<T>::run(__builtin_bit_cast(frag, cur[0]), qf[2 * B], s);
          s = mla_mfma<T>::run(__builtin_bit_cast(frag, cur[1]), qf[2 * B + 1], s);
        });
      }
      // The scores leave the matrix pipe several cycles after the last MFMA issues, and when no key of the tile needs masking the
      // first vector instruction to read them follows that MFMA directly (hipcc placed no wait states on that path): the
      // maximum was then taken from an accumulator one k-step short — a valid reference, but not the same from run to run.
      asm volatile("s_nop 7\n\ts_nop 7" : "+v"(s));
      MLA_STAMP(1);
      // ---- own 16 keys: mask, maximum, probabilities; reference and probabilities published for the partner ---------------
      const int key0 = k_begin + kt * KEYS + 16 * half + 4 * grp;
      if (k_begin + (kt + 1) * KEYS > k_end) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (key0 + r >= k_end) s[r] = -INFINITY;
      }
      float mx = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
      mx = xor_max_16_32(mx);
      ref_own = m;
      if ((mx - m) * a.scale_log2 > 8.0f) ref_own = mx;       // m = -inf: any finite score; NaN (-inf - -inf): keep
      const float ms = (ref_own == -INFINITY ? 0.f : ref_own) * a.scale_log2;
      ps = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = fast_exp2(fmaf(s[r], a.scale_log2, -ms));
        ps += p;
        p_own[r] = static_cast<T>(p);
      }
      {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 w0 = __builtin_bit_cast(u32x2, p_own);
        asm volatile("ds_write_b64 %0, %1\n\tds_write_b32 %2, %3\n\ts_waitcnt lgkmcnt(0)"
                     : : "v"(p_mine), "v"(w0), "v"(max_mine), "v"(ref_own) : "memory");
      }
      MLA_STAMP(2);
    }
    if (late) stage_and_wait(kt);
    MLA_STAMP(3);
    __builtin_amdgcn_s_barrier();
    MLA_STAMP(4);
    if (active) {
      const unsigned vt = smem_u32 + (kt & (SLOTS - 1)) * SLOT;
      frag pf;
      float alpha;
      {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        u32x2 r0;
        float ref_oth;
        asm volatile("ds_read_b64 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(r0), "=&v"(ref_oth) : "v"(p_other), "v"(max_other) : "memory");
        T4 p_oth = __builtin_bit_cast(T4, r0);
        const float ref = fmaxf(ref_own, ref_oth);
        const float f_own = ref_own == ref ? 1.f : fast_exp2((ref_own - ref) * a.scale_log2);   // -inf vs finite: 0
        const float f_oth = ref_oth == ref ? 1.f : fast_exp2((ref_oth - ref) * a.scale_log2);
        T4 p_me = p_own;
        if (!__all(f_own == 1.f)) {
          ps *= f_own;
#pragma unroll
          for (int e = 0; e < 4; ++e) p_me[e] = static_cast<T>(static_cast<float>(p_me[e]) * f_own);
        }
        if (!__all(f_oth == 1.f)) {
#pragma unroll
          for (int e = 0; e < 4; ++e) p_oth[e] = static_cast<T>(static_cast<float>(p_oth[e]) * f_oth);
        }
        // B operand k = 8 grp + e: e < 4 keys 4 grp + e of the tile's first 16 (half 0's), e >= 4 of its last 16 (half 1's)
        const T4 p_lo = half ? p_oth : p_me, p_hi = half ? p_me : p_oth;
#pragma unroll
        for (int e = 0; e < 4; ++e) { pf[e] = p_lo[e]; pf[4 + e] = p_hi[e]; }
        alpha = m == ref ? 1.f : fast_exp2((m - ref) * a.scale_log2);                            // m = -inf: 0 (O and the sum are 0)
        m = ref;
        lsum = lsum * alpha + ps;
      }
      if (!__all(alpha == 1.0f)) {
#pragma unroll
        for (int dt = 0; dt < 16; ++dt) o[dt] *= alpha;
      }
      MLA_STAMP(5);
      // ---- O^T (own 256 d) += C_kv^T P^T over the tile's 32 keys: 4 batches of 8 transposed reads, double-buffered --------
      // batch J = d tiles 2J, 2J + 1;   regs [i * 2 + which], which = 0: key rows 0-15 (k = 8 grp + 0..3), 1: rows 16-31 (+4..7)
      const unsigned tr_base = vt + tr_k0;
      s16x4 va[4], vb[4], vc[4];
#define MLAPP_ISSUE(dst, J)                                                                                            \
      asm volatile(                                                                                                 \
          "ds_read_b64_tr_b16 %0, %4 offset:%6\n\tds_read_b64_tr_b16 %1, %4 offset:%7\n\t"                          \
          "ds_read_b64_tr_b16 %2, %5 offset:%6\n\tds_read_b64_tr_b16 %3, %5 offset:%7"                              \
          : "=&v"(dst[0]), "=&v"(dst[1]), "=&v"(dst[2]), "=&v"(dst[3])                                              \
          : "v"(tr_base + ((32u * (((J) * 2 + 0) & 7)) ^ tr_sw)), "v"(tr_base + ((32u * (((J) * 2 + 1) & 7)) ^ tr_sw)), \
            "i"(((J) >> 2) * 256), "i"(((J) >> 2) * 256 + 16384)                                                    \
          : "memory")
#define MLAPP_RETIRE(dst, N)                                                                                           \
      asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(dst[0]), "+v"(dst[1]), "+v"(dst[2]), "+v"(dst[3]) : : "memory")
#define MLAPP_PV(src, J)                                                                                               \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                               \
        const s16x4 lo = src[i * 2], hi = src[i * 2 + 1];                                                           \
        const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};                                \
        o[(J) * 2 + i] = mla_mfma<T>::run(__builtin_bit_cast(frag, both), pf, o[(J) * 2 + i]);                      \
      }
      // three batches of four reads in flight
      MLAPP_ISSUE(va, 0); MLAPP_ISSUE(vb, 1);
      MLAPP_ISSUE(vc, 2); MLAPP_RETIRE(va, 8); MLAPP_PV(va, 0);
      MLAPP_ISSUE(va, 3); MLAPP_RETIRE(vb, 8); MLAPP_PV(vb, 1);
      MLAPP_ISSUE(vb, 4); MLAPP_RETIRE(vc, 8); MLAPP_PV(vc, 2);
      MLAPP_ISSUE(vc, 5); MLAPP_RETIRE(va, 8); MLAPP_PV(va, 3);
      MLAPP_ISSUE(va, 6); MLAPP_RETIRE(vb, 8); MLAPP_PV(vb, 4);
      MLAPP_ISSUE(vb, 7); MLAPP_RETIRE(vc, 8); MLAPP_PV(vc, 5);
      MLAPP_RETIRE(va, 4); MLAPP_PV(va, 6);
      MLAPP_RETIRE(vb, 0); MLAPP_PV(vb, 7);
#undef MLAPP_ISSUE
#undef MLAPP_RETIRE
#undef MLAPP_PV
      MLA_STAMP(6);
    }
    if (!late) stage_and_wait(kt);
    MLA_STAMP(7);
    __builtin_amdgcn_s_barrier();
    MLA_STAMP(8);
  }
  if (!late && n_kt > 0) __builtin_amdgcn_s_barrier();
#ifdef MLA_STAMPS
  {
    const unsigned wg = blockIdx.y * gridDim.x + blockIdx.x;
    if (wg < 1024) {
      if (lane < 15) g_mla_stamps[(wg * 8 + wave) * 16 + lane] = tacc;
      if (lane == 15) g_mla_stamps[(wg * 8 + wave) * 16 + 15] = static_cast<unsigned>(n_kt);
    }
  }
#endif

  // ---- epilogue: the pair's row sums are added through the maxima slots, each wave stores its 256 d -------------------------
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

constexpr int MLA512_PP_LDS = 4 * (MLAPP_KEYS * 1024 + MLAPP_KEYS * 128) + 1024 * 4 + 8 * 16 * 4 + 8 * 64 * 8;

}  // namespace mojo
