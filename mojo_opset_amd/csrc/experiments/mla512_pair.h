// mla512_pair_kernel: the DeepSeek-V3 geometry (kv_lora_rank r = 512, rope = 64) latent-attention kernel.
// Included by mla_attn.hip (needs MlaArgs, mla_mfma, lds_m, MLA_KEYS).
//
//   * 4 waves, ONE per SIMD, each with the whole 512-entry register file; 64 heads per workgroup (two workgroups per
//     token for H = 128, placed on one XCD so the second finds the tile in L2 — measured TCC hit rate 51 %, HBM
//     traffic = algorithmic).  wave = (hg, half):
//         hg   = wave & 1 : heads 32*hg .. 32*hg+31 of the workgroup's 64 (two 16-head MFMA tiles)
//         half = wave >> 1: QK^T over keys 32*half .. +31 of the 64-key tile   (reads half of K from LDS)
//                           PV   over d    256*half .. +255 of the latent       (reads half of V from LDS)
//     Every K / V fragment read feeds two MFMAs (both head tiles), so a tile costs 4 x (36 + 32) KiB of LDS reads.
//     The two waves of a pair (same hg) exchange, per tile, their row maxima (128 B) and their probabilities (bf16,
//     2 KiB) through LDS: two extra workgroup barriers.
//   * Registers by hand: O^T [256 d x 32 heads] = 128 AGPRs, the query = 144 registers split between AGPRs and VGPRs
//     with asm ties (left alone hipcc overflows the AGPR half and spills; a scratch reload inside the loop drains
//     vmcnt(0), i.e. waits for the LDS-DMA of the next tile).
//   * LDS tile = region A [64 keys][64 chunks] (c_kv, 1 KiB rows) + region B [64 keys][8 chunks] (k_pe, 128 B rows),
//     double-buffered.  Chunk c of key s sits at c ^ ((s & 7) << 1) in A and at c ^ (s & 7) in B — conflict-free
//     (SQ_LDS_BANK_CONFLICT = 0) for the ds_read_b128 row reads and the ds_read_b64_tr_b16 transposed reads.  The
//     swizzle is applied on the global SOURCE address of the LDS-DMA (the LDS side is lane-linear by construction).
//   * All LDS reads of the main loop are inline asm with counted lgkmcnt waits (batches in flight while the previous
//     batch feeds the MFMAs).  Plain C++ LDS loads make hipcc drain vmcnt(0) in front of the first one whenever an
//     LDS-DMA may be in flight.  The loop body contains no scalar loads, so the counted waits are safe.
//   * Power-of-two page sizes only (other sizes take the generic kernel in mla_attn.hip).
//
// Measured (B = 64, H = 128, ctx 4096, page 16, MI355X): 136 us; phase split from s_memtime probes: QK^T ~22 %,
// softmax + exchange ~20 %, PV ~23 %, staging issue/wait ~30 %.  MFMA busy 20 %, LDS busy well under half: with one
// wave per SIMD the kernel is bound by exposed latencies (LDS round trips, the ~160-clock issue stall of each 1 KiB
// vector-memory instruction), not by a throughput limit.  A variant that staged through registers (global_load into
// AGPRs, ds_write at the end of the tile) stalled just as long on issue, which is how the stall was pinned on the CU's
// 64 B/clk vector-memory path rather than on LDS-DMA.
#pragma once
#ifndef MLAP_Q1_AGPR
#define MLAP_Q1_AGPR 10
#endif

namespace mojo {

// (mlap_k_issue, the K-fragment batch reader, lives in mla512_oct.h, which is included first)

template <typename T>
__global__ __launch_bounds__(256, 1) void mla512_pair_kernel(MlaArgs a) {
  typedef typename mla_mfma<T>::frag frag;
  constexpr int R = 512, NK = 18, WAVES = 4, HPB = 64;
  constexpr int A_BYTES = MLA_KEYS * 1024, B_BYTES = MLA_KEYS * 128, TILE = A_BYTES + B_BYTES;   // 72 KiB
  constexpr int TABLE_ENTRIES = 1024;
  constexpr int TABLE_OFF = 2 * TILE, MAX_OFF = TABLE_OFF + TABLE_ENTRIES * 4, P_OFF = MAX_OFF + WAVES * 32 * 4;
  extern __shared__ __attribute__((aligned(1024))) char smem_generic[];
  lds_m* smem = (lds_m*)smem_generic;
  const unsigned smem_u32 = static_cast<unsigned>(reinterpret_cast<size_t>(smem));

  const int tile = blockIdx.x % a.n_tiles, hb = blockIdx.x / a.n_tiles, split = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hg = wave & 1, half = wave >> 1;
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
    for (int i = threadIdx.x; i < TABLE_ENTRIES; i += 256) s_table[i] = (p0 + i < a.max_pages) ? table[p0 + i] : -1;
    win_base = p0;
    __syncthreads();
  };
  auto page_of = [&](int key) { return key >> a.page_shift; };
  fill_window(page_of(k_begin));

  const int head0 = hb * HPB + hg * 32;                 // first head of this wave
  const bool active = head0 < a.heads;                  // identical for the two waves of a pair
  int head[2];
#pragma unroll
  for (int ht = 0; ht < 2; ++ht) head[ht] = min(head0 + ht * 16 + l15, a.heads - 1);

  frag qf[2][NK];
#pragma unroll
  for (int ht = 0; ht < 2; ++ht) {
    const int64_t qrow = static_cast<int64_t>(tile) * a.heads + head[ht];
    const T* qp = static_cast<const T*>(a.q_lat) + qrow * a.q_stride + grp * 8;
    const T* qr = static_cast<const T*>(a.q_rope) + qrow * a.q_rope_stride + grp * 8;
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) qf[ht][ks] = *reinterpret_cast<const frag*>(ks * 32 < R ? qp + ks * 32 : qr + (ks * 32 - R));
  }
  // retire the query loads where the compiler's wait-count pass can see it (see mla512_kernel)
  __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0), as an instruction the wait-count pass models
  // Register classes by hand: O^T (128) + the first head tile's query (72) in AGPRs, the second head tile's query (72)
  // in VGPRs next to ~110 working registers.  Left to itself hipcc tries to keep all 144 query registers in AGPRs,
  // runs out at 256 and spills twenty fragments to scratch — and every scratch reload drains vmcnt(0).
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) {
    asm volatile("" : "+a"(qf[0][ks]));
    if (ks < MLAP_Q1_AGPR) asm volatile("" : "+a"(qf[1][ks]));
    else asm volatile("" : "+v"(qf[1][ks]));
  }

  // ---- staging (as mla512_kernel) ---------------------------------------------------------------------------
  const T* ckv = static_cast<const T*>(a.ckv);
  const T* kpe = static_cast<const T*>(a.kpe);
  // stage_prep: page lookups and row addresses of one tile (lane i < 16 owns c_kv row wave + 4i: the 64-bit address
  // arithmetic runs once in the vector unit, each row's address is later pulled out with a v_readlane pair).
  // stage_piece(i): ONE LDS-DMA instruction (16 c_kv rows of 1 KiB + 2 k_pe blocks of 8 rows per wave).  An LDS-DMA
  // issue blocks the wave ~160 clocks (the CU's vector-memory path takes 64 B/clk); the main loop therefore puts two
  // pieces behind each of the nine QK^T batches, where the stall overlaps the batch's eight MFMAs.
  struct StagePlan { int addr_lo, addr_hi, buf; const T* pe[2]; };
  constexpr int PIECES = MLA_KEYS / WAVES + 2;
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
    int my_phys, phys_b0, phys_b1;
    const int key_l = min(k_first + wave + 4 * (lane & 15), k_end - 1);
    const int row_b0 = (wave * 2 + 0) * 8 + (lane >> 3), row_b1 = (wave * 2 + 1) * 8 + (lane >> 3);
    const int key_b0 = min(k_first + row_b0, k_end - 1), key_b1 = min(k_first + row_b1, k_end - 1);
    asm volatile("ds_read_b32 %0, %3\n\tds_read_b32 %1, %4\n\tds_read_b32 %2, %5\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(my_phys), "=&v"(phys_b0), "=&v"(phys_b1)
                 : "v"(table_u32 + 4 * ((key_l >> a.page_shift) - win_base)), "v"(table_u32 + 4 * ((key_b0 >> a.page_shift) - win_base)),
                   "v"(table_u32 + 4 * ((key_b1 >> a.page_shift) - win_base))
                 : "memory");
    my_phys = max(my_phys, 0);
    phys_b0 = max(phys_b0, 0);
    phys_b1 = max(phys_b1, 0);
    const uint64_t row_addr = reinterpret_cast<uint64_t>(ckv) +
                              2 * (static_cast<uint64_t>(static_cast<unsigned>(my_phys)) * static_cast<uint64_t>(a.ckv_blk) +
                                   static_cast<uint64_t>(static_cast<unsigned>(key_l & mask)) * static_cast<uint64_t>(a.ckv_tok));
    sp.addr_lo = static_cast<int>(row_addr);
    sp.addr_hi = static_cast<int>(row_addr >> 32);
    sp.pe[0] = kpe + static_cast<int64_t>(phys_b0) * a.kpe_blk + static_cast<int64_t>(key_b0 & mask) * a.kpe_tok + ((lane & 7) ^ (row_b0 & 7)) * 8;
    sp.pe[1] = kpe + static_cast<int64_t>(phys_b1) * a.kpe_blk + static_cast<int64_t>(key_b1 & mask) * a.kpe_tok + ((lane & 7) ^ (row_b1 & 7)) * 8;
    return sp;
  };
  const int cs_even = (lane ^ (wave << 1)) * 16, cs_odd = (lane ^ ((4 + wave) << 1)) * 16;   // (row & 7) = 4 (i & 1) + wave
  auto stage_piece = [&](const StagePlan& sp, int i) {      // i: compile-time constant after unrolling
    lds_m* ta = smem + sp.buf * TILE;
    if (i < MLA_KEYS / WAVES) {
      const int row = i * WAVES + wave;
      const unsigned lo = __builtin_amdgcn_readlane(sp.addr_lo, i), hi = __builtin_amdgcn_readlane(sp.addr_hi, i);
      const char* src = reinterpret_cast<const char*>((static_cast<uint64_t>(hi) << 32) | lo);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ((i & 1) ? cs_odd : cs_even)),
                                       (__attribute__((address_space(3))) void*)(ta + row * 1024), 16, 0, 0);
    } else {
      const int j = i - MLA_KEYS / WAVES;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sp.pe[j],
                                       (__attribute__((address_space(3))) void*)(ta + A_BYTES + (wave * 2 + j) * 1024), 16, 0, 0);
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

  // exchange slots
  const unsigned max_mine = smem_u32 + MAX_OFF + (wave * 32 + l15) * 4;            // + ht*64
  const unsigned max_other = smem_u32 + MAX_OFF + ((wave ^ 2) * 32 + l15) * 4;
  const unsigned p_mine = smem_u32 + P_OFF + (wave * 128 + lane) * 16;             // + ht*1024
  const unsigned p_other = smem_u32 + P_OFF + ((wave ^ 2) * 128 + lane) * 16;

  f32x4 o[2][16];
#pragma unroll
  for (int ht = 0; ht < 2; ++ht)
#pragma unroll
    for (int dt = 0; dt < 16; ++dt) o[ht][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m[2] = {-INFINITY, -INFINITY}, lsum[2] = {0.f, 0.f};

  if (n_kt > 0) {
    const StagePlan sp0 = stage_prep(0, 0);
#pragma unroll
    for (int i = 0; i < PIECES; ++i) stage_piece(sp0, i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  for (int kt = 0; kt < n_kt; ++kt) {
    const int buf = kt & 1;
    const bool prefetch = kt + 1 < n_kt;
    StagePlan sp{};
    if (prefetch) sp = stage_prep(kt + 1, buf ^ 1);
    if (prefetch && !active) {
#pragma unroll
      for (int i = 0; i < PIECES; ++i) stage_piece(sp, i);
    }
    const unsigned vt = smem_u32 + buf * TILE;
    frag pf[2][2];                                       // [own / partner][head tile]
    float alpha[2] = {1.f, 1.f}, ref_own[2] = {-INFINITY, -INFINITY}, ps[2] = {0.f, 0.f};
    if (active) {
      // ---- S^T (own 32 keys) = K_lat Q_lat^T: 36 fragment reads in 9 batches, each fragment feeds both head tiles ----
      f32x4 s[2][2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) s[tt][ht] = f32x4{0.f, 0.f, 0.f, 0.f};
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
            const frag kf = __builtin_bit_cast(frag, cur[i]);
            s[tt][0] = mla_mfma<T>::run(kf, qf[0][ks], s[tt][0]);
            s[tt][1] = mla_mfma<T>::run(kf, qf[1][ks], s[tt][1]);
          }
          if (prefetch) { stage_piece(sp, 2 * B); stage_piece(sp, 2 * B + 1); }
        });
      }
      asm volatile("s_nop 7\n\ts_nop 7" : "+v"(s[0][0]), "+v"(s[0][1]), "+v"(s[1][0]), "+v"(s[1][1]));   // see mla512_oct.h
      // ---- row maxima of the own half, exchanged with the partner ----------------------------------------------------
      const int key0 = k_begin + kt * MLA_KEYS + 32 * half + 4 * grp;
      if (k_begin + (kt + 1) * MLA_KEYS > k_end) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (key0 + 16 * tt + r >= k_end) { s[tt][0][r] = -INFINITY; s[tt][1][r] = -INFINITY; }
      }
      // One exchange barrier per tile (see mla512_oct_kernel): each wave keeps its own lazy reference per head tile, taken
      // from its own 32 keys in the vector unit, and publishes it with its probabilities; behind the barrier both waves
      // form the same joint reference and the side that was lower rescales its bf16 probabilities.
#pragma unroll
      for (int ht = 0; ht < 2; ++ht) {
        float v = -INFINITY;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int r = 0; r < 4; ++r) v = fmaxf(v, s[tt][ht][r]);
        v = xor_max_16_32(v);
        ref_own[ht] = m[ht];
        if ((v - m[ht]) * a.scale_log2 > 8.0f) ref_own[ht] = v;
        const float ms = (ref_own[ht] == -INFINITY ? 0.f : ref_own[ht]) * a.scale_log2;
        float sum = 0.f;
        frag f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p0 = fast_exp2(fmaf(s[0][ht][r], a.scale_log2, -ms));
          const float p1 = fast_exp2(fmaf(s[1][ht][r], a.scale_log2, -ms));
          sum += p0 + p1;
          f[r] = static_cast<T>(p0);
          f[4 + r] = static_cast<T>(p1);
        }
        pf[0][ht] = f;
        ps[ht] = sum;
      }
      {
        const u32x4 w0 = __builtin_bit_cast(u32x4, pf[0][0]), w1 = __builtin_bit_cast(u32x4, pf[0][1]);
        asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:1024\n\t"
                     "ds_write_b32 %3, %4\n\tds_write_b32 %3, %5 offset:64\n\ts_waitcnt lgkmcnt(0)"
                     : : "v"(p_mine), "v"(w0), "v"(w1), "v"(max_mine), "v"(ref_own[0]), "v"(ref_own[1]) : "memory");
      }
    }
    __builtin_amdgcn_s_barrier();                                                          // (2) references and probabilities visible
    if (active) {
      {
        u32x4 r0, r1;
        float ro[2];
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\t"
                     "ds_read_b32 %2, %5\n\tds_read_b32 %3, %5 offset:64\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(r0), "=&v"(r1), "=&v"(ro[0]), "=&v"(ro[1]) : "v"(p_other), "v"(max_other) : "memory");
        pf[1][0] = __builtin_bit_cast(frag, r0);
        pf[1][1] = __builtin_bit_cast(frag, r1);
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) {
          const float ref = fmaxf(ref_own[ht], ro[ht]);
          const float f_own = ref_own[ht] == ref ? 1.f : fast_exp2((ref_own[ht] - ref) * a.scale_log2);
          const float f_oth = ro[ht] == ref ? 1.f : fast_exp2((ro[ht] - ref) * a.scale_log2);
          if (!__all(f_own == 1.f)) {
            ps[ht] *= f_own;
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[0][ht][e] = static_cast<T>(static_cast<float>(pf[0][ht][e]) * f_own);
          }
          if (!__all(f_oth == 1.f)) {
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[1][ht][e] = static_cast<T>(static_cast<float>(pf[1][ht][e]) * f_oth);
          }
          alpha[ht] = m[ht] == ref ? 1.f : fast_exp2((m[ht] - ref) * a.scale_log2);
          m[ht] = ref;
          lsum[ht] = lsum[ht] * alpha[ht] + ps[ht];
        }
      }
      if (!__all(alpha[0] == 1.0f && alpha[1] == 1.0f)) {
        // (fenced in groups of four fragments: left alone, the scheduler hoists all 128 accumulator reads to the top
        // and the 128 temporaries push the query fragments out to scratch)
#pragma unroll
        for (int ht = 0; ht < 2; ++ht)
#pragma unroll
          for (int dt = 0; dt < 16; ++dt) {
            o[ht][dt] *= alpha[ht];
            if ((dt & 3) == 3) __builtin_amdgcn_sched_barrier(0);
          }
      }
      // ---- O^T (own 256 d) += C_kv^T P^T over all 64 keys: 8 batches of 8 transposed reads, double-buffered ----------
      // batch J = d tiles 2J, 2J+1;   regs [i*4 + which*2 + {lo,hi}], which = 0: own key half, 1: partner's
      const unsigned v_own = vt + own_off, v_oth = vt + oth_off;
      s16x4 va[8], vb[8];
#define MLAP_ISSUE(dst, J)                                                                                             \
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
#define MLAP_RETIRE(dst, N)                                                                                            \
      asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                        \
                   : "+v"(dst[0]), "+v"(dst[1]), "+v"(dst[2]), "+v"(dst[3]), "+v"(dst[4]), "+v"(dst[5]), "+v"(dst[6]),  \
                     "+v"(dst[7])                                                                                       \
                   : : "memory")
#define MLAP_PV(src, J)                                                                                                \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int w = 0; w < 2; ++w) {                   \
        const s16x4 lo = src[i * 4 + w * 2], hi = src[i * 4 + w * 2 + 1];                                              \
        const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};                                   \
        const frag vf = __builtin_bit_cast(frag, both);                                                                \
        o[0][(J) * 2 + i] = mla_mfma<T>::run(vf, pf[w][0], o[0][(J) * 2 + i]);                                         \
        o[1][(J) * 2 + i] = mla_mfma<T>::run(vf, pf[w][1], o[1][(J) * 2 + i]);                                         \
      }
#define MLAP_STEP2(J)                                                                                                  \
      MLAP_ISSUE(vb, (J) + 1); MLAP_RETIRE(va, 8); MLAP_PV(va, (J));                                                   \
      MLAP_ISSUE(va, (J) + 2); MLAP_RETIRE(vb, 8); MLAP_PV(vb, (J) + 1);
      MLAP_ISSUE(va, 0);
      MLAP_STEP2(0) MLAP_STEP2(2) MLAP_STEP2(4)
      MLAP_ISSUE(vb, 7); MLAP_RETIRE(va, 8); MLAP_PV(va, 6);
      MLAP_RETIRE(vb, 0); MLAP_PV(vb, 7);
#undef MLAP_STEP2
#undef MLAP_ISSUE
#undef MLAP_RETIRE
#undef MLAP_PV
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                                          // (3) next tile staged, this one free
  }

  // ---- epilogue: the pair's row sums are added through the maxima slots, each wave stores its 256 d ------------------
  float lt[2];
#pragma unroll
  for (int ht = 0; ht < 2; ++ht) {
    float v = lsum[ht];
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    lt[ht] = v;
  }
  asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:64\n\ts_waitcnt lgkmcnt(0)" : : "v"(max_mine), "v"(lt[0]), "v"(lt[1]) : "memory");
  __builtin_amdgcn_s_barrier();
  {
    float l0, l1;
    asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:64\n\ts_waitcnt lgkmcnt(0)" : "=&v"(l0), "=&v"(l1) : "v"(max_other) : "memory");
    lt[0] += l0;
    lt[1] += l1;
  }
  if (!active) return;
  typedef typename vec_of<T, 4>::type V4;
#pragma unroll
  for (int ht = 0; ht < 2; ++ht) {
    if (head0 + ht * 16 + l15 >= a.heads) continue;
    const int hd = head[ht];
    const int d0 = half * 256 + grp * 4;
    if (a.n_splits == 1) {
      float den = lt[ht];
      float w = 1.f;
      const float ml2 = m[ht] * a.scale_log2;
      if (a.sink) {
        const float sk = a.sink[hd] * 1.4426950408889634f;
        const float M = fmaxf(ml2, sk);
        w = (m[ht] == -INFINITY) ? 0.f : fast_exp2(ml2 - M);
        den = lt[ht] * w + fast_exp2(sk - M);
      }
      const float inv = den > 0.f ? w / den : 0.f;
      T* dst = static_cast<T*>(a.o_lat) + (static_cast<int64_t>(tile) * a.heads + hd) * R + d0;
#pragma unroll
      for (int dt = 0; dt < 16; ++dt) {
        V4 ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) ov[r] = static_cast<T>(o[ht][dt][r] * inv);
        *reinterpret_cast<V4*>(dst + dt * 16) = ov;
      }
    } else {
      const int64_t slot = (static_cast<int64_t>(tile) * a.n_splits + split) * a.heads + hd;
      float* po = a.part_o + slot * R + d0;
#pragma unroll
      for (int dt = 0; dt < 16; ++dt) *reinterpret_cast<f32x4*>(po + dt * 16) = o[ht][dt];
      if (grp == 0 && half == 0) {
        a.part_ml[slot * 2] = m[ht] * a.scale_log2;
        a.part_ml[slot * 2 + 1] = lt[ht];
      }
    }
  }
}

constexpr int MLA512_PAIR_LDS = 2 * (MLA_KEYS * 1024 + MLA_KEYS * 128) + 1024 * 4 + 4 * 32 * 4 + 4 * 128 * 16;

}  // namespace mojo
