// mla512_ps_kernel: the r = 512 / rope = 64 latent-attention kernel with SPECIALISED waves (round 3).
// Included by mla_attn.hip after mla512_oct.h (shares MlaArgs, lds_m).
//
// What bounded the lock-step kernel (mla512_oct.h) and the ping-pong kernel (mla512_pp.h) was the length of ONE wave's serial
// stream per key: every wave did QK^T, softmax, an exchange with its partner, PV and its share of the staging, so at most
// ~1 100 of its ~5 700 cycles per 64 keys were MFMA issue, and with 16 heads per wave every 1 KiB LDS fragment fed a single
// 16-cycle MFMA (LDS traffic 544 KiB per 64 keys per CU = as many LDS cycles as MFMA cycles).  Here the 8 waves of a
// workgroup (64 heads, 2 per SIMD) have three roles:
//
//   waves 0, 1   PRODUCERS   heads 32 g .. 32 g + 31 (g = wave): S^T[32 keys x 32 heads] = K Q^T of one 32-key slot with
//                            v_mfma_f32_32x32x16 (36 MFMAs, ONE accumulator chain), online softmax (lazy reference maximum,
//                            row sums), probabilities rounded to the storage type and published through LDS in the B-operand
//                            layout of the second product (accumulator-as-operand rule: no lane movement), together with the
//                            per-head rescale factor.  144 query registers, no output accumulators.
//   waves 2,3,6,7 CONSUMERS  group g = wave & 1, latent dims 256 (wave >> 2) .. + 255: O^T[256 d x 32 heads] += C_kv^T P^T over
//                            the slot's 32 keys (16 MFMAs 32x32x16, transposed LDS reads), 128 accumulator registers, no query.
//   waves 4, 5   LOADERS     the SIMD-mates of the producers: all LDS-DMA (18 pieces of 1 KiB each per slot: 16 c_kv rows of one
//                            page + 2 k_pe blocks), page ids by scalar loads, counted waits.  A 1-KiB vector-memory
//                            instruction stalls its issuing wave for 60-180 cycles (mla512_pair.h); here that wave has
//                            nothing else to do and its SIMD-mate keeps the matrix pipe.
//
// A 32-head fragment halves the LDS traffic per FLOP (a 1 KiB K fragment feeds a 32-cycle MFMA, a 1 KiB V^T fragment too),
// and nobody waits for a partner inside a slot: the only synchronisation is ONE workgroup barrier per 32-key slot.  In step k
// the producers work on slot k + 1, the consumers on slot k (with the probabilities published in step k - 1), the loaders
// request slot k + 3 and make sure slot k + 2 has landed: a ring of four 36-KiB slots, two in use, two in flight.
//
// LDS image of a slot: c_kv rows [32 keys][1024 B] + k_pe rows [32 keys][128 B].  16-byte chunk c of c_kv row s sits at
// position c ^ f(s & 15) (low four bits), f(x) = ((x & 3) << 2) | (x >> 2): conflict-free both for the row reads of the
// 32x32x16 A operand (ds_read_b128: 16-lane groups {0-3,12-15,20-27}, ... take 16 distinct f) and for the transposed reads
// (ds_read_b64_tr_b16: the 8 row/column blocks of a half-wave land in 8 distinct 32-byte bank slots) — the layout (b) of the
// CDNA4 guide's dual-use image, widened to 1-KiB rows.  k_pe chunk c of row s sits at c ^ ((s >> 1) & 7).  The swizzle is
// applied to the SOURCE address of the LDS-DMA (the LDS side of a DMA piece is lane-linear by construction).
//
// Key order inside a 16-key step of the second product: element j of lane half h is key 16 st + 8 (j >> 2) + 4 h + (j & 3),
// on both operands (the accumulator registers 8 st .. 8 st + 7 of S^T ARE that order; the transposed reads pick their four
// rows accordingly).  Power-of-two pages of >= 16 tokens (a loader's 16 rows then share one page id); other page sizes take
// the lock-step kernel.
#pragma once

namespace mojo {

template <typename T> struct mla_mfma32;
template <> struct mla_mfma32<bf16_t> {
  typedef bf16x8 frag;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct mla_mfma32<f16_t> {
  typedef f16x8 frag;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

constexpr int MLAPS_KEYS = 32, MLAPS_SLOTS = 4, MLAPS_HPB = 64;
constexpr int MLAPS_A = MLAPS_KEYS * 1024, MLAPS_B = MLAPS_KEYS * 128, MLAPS_SLOT = MLAPS_A + MLAPS_B;   // 36 KiB
constexpr int MLAPS_P_OFF = MLAPS_SLOTS * MLAPS_SLOT;          // probabilities: [group 2][parity 2][k-step 2][64 lanes][16 B]
constexpr int MLAPS_ALPHA_OFF = MLAPS_P_OFF + 2 * 2 * 2 * 1024;  // rescale factors / final normalisers: [group 2][parity 2][64 lanes][4 B]
constexpr int MLAPS_PF_OFF = MLAPS_ALPHA_OFF + 2 * 2 * 256;      // sink of the L2 prefetch loads: [loader 2][256 B]
constexpr int MLA512_PS_LDS = MLAPS_PF_OFF + 2 * 256;

// QK^T fragment read n of a slot (k-step ks = n: 32 over the latent, 4 over the rope part), batch of four
template <int B, int I = 0>
__device__ __forceinline__ void mlaps_k_issue(u32x4 (&dst)[4], const unsigned (&kav)[8], const unsigned (&kbv)[4]) {
  constexpr int ks = 4 * B + I;
  if constexpr (ks < 32)
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[I]) : "v"(kav[ks & 7]), "i"((ks >> 3) * 256) : "memory");
  else
    asm volatile("ds_read_b128 %0, %1" : "=v"(dst[I]) : "v"(kbv[ks - 32]) : "memory");
  if constexpr (I + 1 < 4) mlaps_k_issue<B, I + 1>(dst, kav, kbv);
}

template <typename T>
__global__ __launch_bounds__(512, 2) void mla512_ps_kernel(MlaArgs a) {
  typedef typename mla_mfma32<T>::frag frag;
  constexpr int R = 512, NKS = 36, KEYS = MLAPS_KEYS, SLOT = MLAPS_SLOT;
  extern __shared__ __attribute__((aligned(1024))) char smem_generic[];
  lds_m* smem = (lds_m*)smem_generic;
  const unsigned smem_u32 = static_cast<unsigned>(reinterpret_cast<size_t>(smem));

  const int tile = blockIdx.x % a.n_tiles, hb = blockIdx.x / a.n_tiles, split = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r31 = lane & 31, h = lane >> 5;

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
  n_vis = __builtin_amdgcn_readfirstlane(n_vis);
  const int k_begin = split * a.split_keys;
  const int k_end = min(n_vis, k_begin + a.split_keys);
  const int n_kt = k_end > k_begin ? (k_end - k_begin + KEYS - 1) / KEYS : 0;      // 32-key slots of this split

  // roles
  const bool is_producer = wave < 2, is_loader = (wave & 6) == 4;
  const int grp = wave & 1;                             // head group of a producer / consumer; loader index of a loader
  const int dhalf = wave >> 2;                          // consumers: latent dims 256 dhalf .. + 255
  const int head0 = hb * MLAPS_HPB + grp * 32;
  const bool active = head0 < a.heads;                  // (a group without heads idles through the barriers)
  const int head = min(head0 + r31, a.heads - 1);
  const unsigned p_base = smem_u32 + MLAPS_P_OFF + grp * 4096 + lane * 16;          // + parity * 2048 + kstep * 1024
  const unsigned al_base = smem_u32 + MLAPS_ALPHA_OFF + grp * 512 + lane * 4;       // + parity * 256

  // One barrier per step; steps -1 .. n_kt - 1.  Step k: producers slot k + 1, consumers slot k, loaders request slot k + 3
  // and wait for slot k + 2.  (n_kt = 0: nothing to stream; the epilogue writes the empty state.)
  // ---- LDS-DMA of a share of a slot: c_kv rows 16 li + LO .. + N - 1 and k_pe blocks 2 li + pe_first .. of slot kt -----------------
  // (who issues what: `six` below)
  const char* const ckv_b = static_cast<const char*>(a.ckv);
  const char* const kpe_b = static_cast<const char*>(a.kpe);
  const int pmask = a.page - 1;
  const unsigned lane16 = lane * 16;
  auto issue_share = [&](int kt, int li, auto LOC, auto NC, int pe_first, int pe_count) {
    constexpr int LO = decltype(LOC)::value, N = decltype(NC)::value;
    const int key_g = k_begin + kt * KEYS + 16 * li;                            // first key of the 16-row group (uniform)
    const int key_c = min(key_g, k_end - 1);
    // A SCALAR load, written out: behind a barrier or an LDS-DMA hipcc no longer proves the table unclobbered and falls back
    // to a vector load, whose vmcnt(0) wait would drain every DMA piece in flight (the whole prefetch depth of the ring).
    int phys;
    {
      const int32_t* pt = table + (key_c >> a.page_shift);
      asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(phys) : "s"(pt) : "memory");
    }
    phys = max(phys, 0);
    lds_m* ta = smem + (kt & (MLAPS_SLOTS - 1)) * SLOT;
    const char* page_c = ckv_b + 2 * static_cast<int64_t>(phys) * a.ckv_blk;
    static_for<N>([&](auto IC) {
      constexpr int i = LO + decltype(IC)::value;
      constexpr unsigned fsw = (((i & 3) << 2) | (i >> 2)) * 16;                    // f(row & 15) * 16: row & 15 = i
      const int key_i = min(key_g + i, k_end - 1);                                // rows past the end re-read the last key (masked later)
      const char* src = page_c + 2 * static_cast<int64_t>(key_i & pmask) * a.ckv_tok + (lane16 ^ fsw);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(ta + (16 * li + i) * 1024), 16, 0, 0);
    });
    const char* page_p = kpe_b + 2 * static_cast<int64_t>(phys) * a.kpe_blk;
    for (int jj = pe_first; jj < pe_first + pe_count; ++jj) {
      const int row = 16 * li + 8 * jj + (lane >> 3);
      const int key = min(k_begin + kt * KEYS + row, k_end - 1);
      const char* src = page_p + 2 * static_cast<int64_t>(key & pmask) * a.kpe_tok + (((lane & 7) ^ ((row >> 1) & 7)) * 16);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(ta + MLAPS_A + (2 * li + jj) * 1024), 16, 0, 0);
    }
  };
  // six = the four consumers issue too: a loader 6 rows of its 16-row group, the two consumers of group li 5 rows and one
  // k_pe block each (6 pieces per wave and slot); otherwise (default) a loader issues all 18 pieces of its group.  Measured
  // equal (stream alone, compute ablated: 106.7 vs 106.3 us per op): the 36 pieces of a slot are not bound by the ~100 cycles
  // each costs its issuing wave but by what one CU's fill path delivers (~36 GB/s for a stream that is half L2 hits).
  const bool six = a.ps_issuers == 6;
  using ic0 = std::integral_constant<int, 0>;
  // L2 prefetch issued by a CONSUMER wave (ps_prefetch == 2; round 3, second attempt).  Issued by the loaders (ps_prefetch == 1)
  // the prefetch loads are older than the DMA pieces in the same wave's in-order vmcnt, so the loader's counted wait for a slot
  // also waited for prefetch loads that miss to HBM: slower.  A consumer has no vector-memory instruction in its loop and never
  // waits for these (their destination is an LDS scratch nobody reads).
  constexpr int PF2 = 3;
  auto prefetch_group = [&](int kt, int li) {
    const int key_c = min(k_begin + kt * KEYS + 16 * li, k_end - 1) & ~15;
    int phys;
    {
      const int32_t* pt = table + (key_c >> a.page_shift);
      asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(phys) : "s"(pt) : "memory");
    }
    phys = max(phys, 0);
    const char* pc = ckv_b + 2 * (static_cast<int64_t>(phys) * a.ckv_blk + static_cast<int64_t>(key_c & pmask) * a.ckv_tok);
    const char* pp = kpe_b + 2 * (static_cast<int64_t>(phys) * a.kpe_blk + static_cast<int64_t>(key_c & pmask) * a.kpe_tok);
    lds_m* sink = smem + MLAPS_PF_OFF + li * 256;
    const unsigned c0 = (lane >> 3) * static_cast<unsigned>(2 * a.ckv_tok) + (lane & 7) * 128;     // rows 0-7, one dword per 128-byte line
    const unsigned c1 = c0 + 8u * static_cast<unsigned>(2 * a.ckv_tok);                           // rows 8-15
    const unsigned pe = (lane & 15) * static_cast<unsigned>(2 * a.kpe_tok);                        // 16 rows of 128 B
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pc + c0), (__attribute__((address_space(3))) void*)sink, 4, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pc + c1), (__attribute__((address_space(3))) void*)sink, 4, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pp + pe), (__attribute__((address_space(3))) void*)sink, 4, 0, 0);
  };

  if (is_loader) {
    // ---------------------------------------------------------------------------------------------------------------------
    // LOADER li: c_kv rows 16 li .. 16 li + 15 and k_pe rows 16 li .. + 15 of every slot
    // ---------------------------------------------------------------------------------------------------------------------
    const int li = grp;
    auto issue_slot = [&](int kt) {
      if (six) issue_share(kt, li, ic0{}, std::integral_constant<int, 6>{}, 0, 0);
      else issue_share(kt, li, ic0{}, std::integral_constant<int, 16>{}, 0, 2);
    };
    const char* ckv = ckv_b;
    const char* kpe = kpe_b;
    const int mask = pmask;
    // L2 prefetch, PF slots ahead of the LDS-DMA.  The ring holds two slots in use and two in flight: ~50 KiB per CU on the way,
    // which at the ~1.5-2 us an HBM miss takes under load is ~30 GB/s per CU — the stream, not the matrix pipe, then bounds the
    // kernel (measured with the compute ablated: 65-70 us for the 2 x 302 MB the two head-block workgroups of every token pull).
    // LDS cannot hold more, the XCD's L2 can: one dword per 128-byte line of a later slot brings those lines on chip (3
    // instructions per loader and slot), and the DMA that follows finds them there at L2-hit latency.  The dwords go by LDS-DMA
    // into a 256-byte scratch nobody reads — a load with a REGISTER destination that is never waited for would write into
    // whatever the compiler has put in that register by the time it lands (it did: a memory fault).
    constexpr int PF = 4;
    lds_m* pf_scratch = smem + MLAPS_PF_OFF + li * 256;
    const unsigned pf_c0 = (lane >> 3) * static_cast<unsigned>(2 * a.ckv_tok) + (lane & 7) * 128;   // rows 0-7 of the page slice
    const unsigned pf_c1 = pf_c0 + 8u * static_cast<unsigned>(2 * a.ckv_tok);                       // rows 8-15
    const unsigned pf_p = (lane & 15) * static_cast<unsigned>(2 * a.kpe_tok);                        // 16 rows of 128 B (lanes 16+ repeat)
    const bool pf_on = a.ps_prefetch == 1;       // default OFF: measured 126 -> 136 us per op (stream alone 102 -> 125 us, compute ablated)
    auto prefetch_slot = [&](int kt) {
      const int key_g = k_begin + kt * KEYS + 16 * li;
      const int key_c = min(key_g, k_end - 1) & ~15;
      int phys;
      {
        const int32_t* pt = table + (key_c >> a.page_shift);
        asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(phys) : "s"(pt) : "memory");
      }
      phys = max(phys, 0);
      const char* pc = ckv + 2 * (static_cast<int64_t>(phys) * a.ckv_blk + static_cast<int64_t>(key_c & mask) * a.ckv_tok);
      const char* pp = kpe + 2 * (static_cast<int64_t>(phys) * a.kpe_blk + static_cast<int64_t>(key_c & mask) * a.kpe_tok);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pc + pf_c0),
                                       (__attribute__((address_space(3))) void*)pf_scratch, 4, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pc + pf_c1),
                                       (__attribute__((address_space(3))) void*)pf_scratch, 4, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pp + pf_p),
                                       (__attribute__((address_space(3))) void*)pf_scratch, 4, 0, 0);
    };
    if (n_kt > 0) {
      if (pf_on && !six)
        for (int t = 2; t < 2 + PF && t < n_kt; ++t) prefetch_slot(t);           // (older than every DMA piece: no wait counts them)
      issue_slot(0);
      if (n_kt > 1) {
        issue_slot(1);
        if (six) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();                                                  // slot 0 landed
      for (int k = -1; k < n_kt; ++k) {
        if (k + 3 < n_kt && !((a.ps_debug & 1) && k > 4)) {
          if (pf_on && !six && k + 3 + PF < n_kt) {
            prefetch_slot(k + 3 + PF);
            issue_slot(k + 3);
            asm volatile("s_waitcnt vmcnt(21)" ::: "memory");                        // newest slot + its 3 prefetch loads may be out
          } else {
            issue_slot(k + 3);                                                       // all but the newest slot: slot k + 2 has landed
            if (six) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
          }
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
      }
    }

    // (n_splits == 1: the producers publish the normalisers behind one more barrier)
    if (a.n_splits == 1) __builtin_amdgcn_s_barrier();
    return;
  }

  if (is_producer) {
    // ---------------------------------------------------------------------------------------------------------------------
    // PRODUCER of head group grp
    // ---------------------------------------------------------------------------------------------------------------------
    frag qf[NKS];
    {
      const int64_t qrow = static_cast<int64_t>(tile) * a.heads + head;
      const T* qp = static_cast<const T*>(a.q_lat) + qrow * a.q_stride + h * 8;
      const T* qr = static_cast<const T*>(a.q_rope) + qrow * a.q_rope_stride + h * 8;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) qf[ks] = *reinterpret_cast<const frag*>(ks < 32 ? qp + ks * 16 : qr + (ks - 32) * 16);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0): retire the query loads where the wait-count pass sees it
    // per-lane read offsets inside a slot: row r31, chunk (2 ks + h) ^ f(r31 & 15)  ->  (A0 ^ 32 (ks & 7)) + 256 (ks >> 3)
    const int fr = ((r31 & 3) << 2) | ((r31 >> 2) & 3);
    const unsigned a0 = r31 * 1024 + ((h ^ fr) << 4);
    const unsigned b0 = MLAPS_A + r31 * 128 + ((h ^ ((r31 >> 1) & 7)) << 4);
    float m = -INFINITY, lsum = 0.f;

    if (n_kt > 0) __builtin_amdgcn_s_barrier();                                      // slot 0 landed
    for (int k = -1; k < n_kt && n_kt > 0; ++k) {
      const int kt = k + 1;                                                          // the slot of this step
      if (kt < n_kt && active) {
        const unsigned vt = smem_u32 + (kt & (MLAPS_SLOTS - 1)) * SLOT;
        unsigned kav[8], kbv[4];
#pragma unroll
        for (int v = 0; v < 8; ++v) kav[v] = vt + (a0 ^ (32u * v));
#pragma unroll
        for (int v = 0; v < 4; ++v) kbv[v] = vt + (b0 ^ (32u * v));
        f32x16 s;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.f;
        if (!(a.ps_debug & 2)) {
          u32x4 kr[2][4];
          mlaps_k_issue<0>(kr[0], kav, kbv);
          static_for<9>([&](auto BC) {
            constexpr int B = decltype(BC)::value;
            if constexpr (B + 1 < 9) mlaps_k_issue<B + 1>(kr[(B + 1) & 1], kav, kbv);
            u32x4 (&cur)[4] = kr[B & 1];
            if constexpr (B + 1 < 9)
              asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]) : : "memory");
            else
              asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]) : : "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i) s = mla_mfma32<T>::run(__builtin_bit_cast(frag, cur[i]), qf[4 * B + i], s);
          });
        }
        // wait states between the last MFMA and the first vector read of the scores (no key to mask: hipcc places none)
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(s));
        // lane: head r31, keys k_first + (i & 3) + 8 (i >> 2) + 4 h
        const int key0 = k_begin + kt * KEYS + 4 * h;
        if (k_begin + (kt + 1) * KEYS > k_end) {
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if (key0 + (i & 3) + 8 * (i >> 2) >= k_end) s[i] = -INFINITY;
        }
        if (a.ps_debug & 4) {                                                         // ablation: no softmax
          __builtin_amdgcn_s_barrier();
          continue;
        }
        float mx = s[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s[i]);
        {
          float p = mx, q = mx;
          asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(p), "+v"(q));
          mx = fmaxf(p, q);                                                            // both key halves of the head
        }
        float ref = m;
        if ((mx - m) * a.scale_log2 > 8.0f) ref = mx;                                 // m = -inf: any finite score; NaN (-inf - -inf): keep
        const float alpha = m == ref ? 1.f : fast_exp2((m - ref) * a.scale_log2);    // m = -inf: 0 (O and the sum are 0)
        m = ref;
        const float ms = (ref == -INFINITY ? 0.f : ref) * a.scale_log2;
        float ps = 0.f;
        frag pf[2];
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          frag f;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float p = fast_exp2(fmaf(s[8 * st + j], a.scale_log2, -ms));
            ps += p;
            f[j] = static_cast<T>(p);
          }
          pf[st] = f;
        }
        lsum = lsum * alpha + ps;
        {
          const unsigned pw = p_base + (kt & 1) * 2048;
          const u32x4 w0 = __builtin_bit_cast(u32x4, pf[0]), w1 = __builtin_bit_cast(u32x4, pf[1]);
          asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:1024\n\tds_write_b32 %3, %4\n\ts_waitcnt lgkmcnt(0)"
                       : : "v"(pw), "v"(w0), "v"(w1), "v"(al_base + (kt & 1) * 256), "v"(alpha) : "memory");
        }
      }
      __builtin_amdgcn_s_barrier();
    }
    // ---- the group's final state: row sums of the two key halves, reference maximum ---------------------------------------
    float lt = lsum;
    {
      float p = lt, q = lt;
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(p), "+v"(q));
      lt = p + q;
    }
    if (a.n_splits == 1) {
      float den = lt, w = 1.f;
      const float ml2 = m * a.scale_log2;
      if (a.sink) {
        const float sk = a.sink[head] * 1.4426950408889634f;
        const float M = fmaxf(ml2, sk);
        w = (m == -INFINITY) ? 0.f : fast_exp2(ml2 - M);
        den = lt * w + fast_exp2(sk - M);
      }
      const float inv = den > 0.f ? w / den : 0.f;
      asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : : "v"(al_base), "v"(inv) : "memory");
      __builtin_amdgcn_s_barrier();
    } else if (active && h == 0 && head0 + r31 < a.heads) {
      const int64_t slot = (static_cast<int64_t>(tile) * a.n_splits + split) * a.heads + head;
      a.part_ml[slot * 2] = m * a.scale_log2;
      a.part_ml[slot * 2 + 1] = lt;
    }
    return;
  }

  // -----------------------------------------------------------------------------------------------------------------------
  // CONSUMER of head group grp, latent dims 256 dhalf .. + 255
  // -----------------------------------------------------------------------------------------------------------------------
  f32x16 o[8];
#pragma unroll
  for (int dt = 0; dt < 8; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
  // transposed reads: lane (h, blk, q, p) supplies row 16 st + 4 h + q (+ 8), columns 32 dt + 16 blk + 4 p .. + 3
  const int blk = r31 >> 4, tq = (r31 & 15) >> 2, tp = r31 & 3;
  const unsigned base1 = (4 * h + tq) * 1024 + (tq << 6) + (((2 * blk + (tp >> 1)) ^ h) << 4) + (tp & 1) * 8 + dhalf * 512;
  unsigned t1o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) t1o[e] = base1 ^ (e << 6);

  auto issue_mine = [&](int kt) {                                                     // six: rows 6 + 5 dhalf .. + 4 and k_pe block dhalf of group grp
    if (dhalf) issue_share(kt, grp, std::integral_constant<int, 11>{}, std::integral_constant<int, 5>{}, 1, 1);
    else issue_share(kt, grp, std::integral_constant<int, 6>{}, std::integral_constant<int, 5>{}, 0, 1);
  };
  if (n_kt > 0) {
    if (six) {
      issue_mine(0);
      if (n_kt > 1) {
        issue_mine(1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    __builtin_amdgcn_s_barrier();                                                    // slot 0 landed
  }
  for (int kt = -1; kt < n_kt && n_kt > 0; ++kt) {                                   // step kt: slot kt (step -1: the producers' first slot)
    const bool more = six && kt + 3 < n_kt && !((a.ps_debug & 1) && kt > 4);
    if (more) issue_mine(kt + 3);
    if (a.ps_prefetch == 2 && dhalf == 0 && kt + 3 + PF2 < n_kt) prefetch_group(kt + 3 + PF2, grp);                                                    // (its ring position was freed by the barrier of step kt - 1)
    if (kt >= 0 && active && !(a.ps_debug & 8)) {
      const unsigned vt = smem_u32 + (kt & (MLAPS_SLOTS - 1)) * SLOT;
      u32x4 pw0, pw1;
      float alpha;
      asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:1024\n\tds_read_b32 %2, %4\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(pw0), "=&v"(pw1), "=&v"(alpha) : "v"(p_base + (kt & 1) * 2048), "v"(al_base + (kt & 1) * 256) : "memory");
      frag pf[2] = {__builtin_bit_cast(frag, pw0), __builtin_bit_cast(frag, pw1)};
      if (!__all(alpha == 1.0f)) {
#pragma unroll
        for (int dt = 0; dt < 8; ++dt)
#pragma unroll
          for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
      }
      unsigned t1[4], t2[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) { t1[e] = vt + t1o[e]; t2[e] = t1[e] ^ 32u; }
      // per d tile: 4 transposed reads (k-steps 0 / 1, rows q / q + 8), two MFMAs; two d tiles in flight
      s16x4 va[4], vb[4];
#define MLAPS_ISSUE(dst, DT)                                                                                           \
      asm volatile(                                                                                                   \
          "ds_read_b64_tr_b16 %0, %4 offset:%6\n\tds_read_b64_tr_b16 %1, %5 offset:%7\n\t"                            \
          "ds_read_b64_tr_b16 %2, %4 offset:%8\n\tds_read_b64_tr_b16 %3, %5 offset:%9"                                \
          : "=&v"(dst[0]), "=&v"(dst[1]), "=&v"(dst[2]), "=&v"(dst[3])                                                \
          : "v"(t1[(DT) & 3]), "v"(t2[(DT) & 3]), "i"(((DT) >> 2) * 256), "i"(((DT) >> 2) * 256 + 8192),              \
            "i"(((DT) >> 2) * 256 + 16384), "i"(((DT) >> 2) * 256 + 16384 + 8192)                                     \
          : "memory")
#define MLAPS_RETIRE(dst, N)                                                                                           \
      asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(dst[0]), "+v"(dst[1]), "+v"(dst[2]), "+v"(dst[3]) : : "memory")
#define MLAPS_PV(src, DT)                                                                                              \
      _Pragma("unroll") for (int st = 0; st < 2; ++st) {                                                              \
        const s16x4 lo = src[st * 2], hi = src[st * 2 + 1];                                                           \
        const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};                                  \
        o[(DT)] = mla_mfma32<T>::run(__builtin_bit_cast(frag, both), pf[st], o[(DT)]);                                \
      }
      MLAPS_ISSUE(va, 0);
      MLAPS_ISSUE(vb, 1); MLAPS_RETIRE(va, 4); MLAPS_PV(va, 0);
      MLAPS_ISSUE(va, 2); MLAPS_RETIRE(vb, 4); MLAPS_PV(vb, 1);
      MLAPS_ISSUE(vb, 3); MLAPS_RETIRE(va, 4); MLAPS_PV(va, 2);
      MLAPS_ISSUE(va, 4); MLAPS_RETIRE(vb, 4); MLAPS_PV(vb, 3);
      MLAPS_ISSUE(vb, 5); MLAPS_RETIRE(va, 4); MLAPS_PV(va, 4);
      MLAPS_ISSUE(va, 6); MLAPS_RETIRE(vb, 4); MLAPS_PV(vb, 5);
      MLAPS_ISSUE(vb, 7); MLAPS_RETIRE(va, 4); MLAPS_PV(va, 6);
      MLAPS_RETIRE(vb, 0); MLAPS_PV(vb, 7);
#undef MLAPS_ISSUE
#undef MLAPS_RETIRE
#undef MLAPS_PV
    }
    if (six) {                                                                       // slot kt + 2 has landed (all but the newest share)
      if (more) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
  }
  if (a.ps_prefetch == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // no prefetch load may land in LDS behind this workgroup
  // ---- epilogue: lane holds head r31, dims 256 dhalf + 32 dt + 8 (i >> 2) + 4 h + (i & 3) -----------------------------------
  const bool store = active && head0 + r31 < a.heads;
  if (a.n_splits == 1) {
    __builtin_amdgcn_s_barrier();                                                    // the producers' normalisers
    float inv;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(inv) : "v"(al_base) : "memory");
    if (!store) return;
    typedef typename vec_of<T, 4>::type V4;
    T* dst = static_cast<T*>(a.o_lat) + (static_cast<int64_t>(tile) * a.heads + head) * R + dhalf * 256 + 4 * h;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        V4 ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) ov[r] = static_cast<T>(o[dt][4 * g4 + r] * inv);
        *reinterpret_cast<V4*>(dst + dt * 32 + g4 * 8) = ov;
      }
  } else {
    if (!store || (a.ps_debug & 32)) return;
    const int64_t slot = (static_cast<int64_t>(tile) * a.n_splits + split) * a.heads + head;
    float* po = a.part_o + slot * R + dhalf * 256 + 4 * h;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
        *reinterpret_cast<f32x4*>(po + dt * 32 + g4 * 8) = f32x4{o[dt][4 * g4], o[dt][4 * g4 + 1], o[dt][4 * g4 + 2], o[dt][4 * g4 + 3]};
  }
}

}  // namespace mojo
