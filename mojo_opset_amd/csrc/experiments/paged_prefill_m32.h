// prefill_m32_kernel: prefill_kernel's decomposition (4 waves x 32 rows per workgroup, two workgroups per CU, 64-key tiles
// double-buffered by LDS-DMA, lazy reference maximum, key split) on v_mfma_f32_32x32x16 instead of v_mfma_f32_16x16x32,
// head_dim 128.  Included by paged_prefill_gqa.hip behind prefill_kernel in an EXPERIMENTS build (MOJO_HIP_BUILD_EXPERIMENTS=1;
// MOJO_HIP_PREFILL_M32=1 selects it per call).  Parity-green on the whole prefill suite, measured 4-5 % SLOWER than prefill_kernel
// on every bench case (round 4, A/B in one process: 4x2048 867 vs 906 TF, 1x16384 1 053 vs 1 103, 16 ragged 642 vs 663).
//
// The hypothesis (round 4): both prefill kernels are ISSUE-bound, not matrix-pipe-bound — a wave issues ~4 cycles per instruction and an
// MFMA holds the SIMD's issue port for 8 cycles whatever its size (scripts/probes/mfma_gap_probe.hip) — so the lever is fewer
// instructions per FLOP.  A 32x32x16 MFMA does twice the FLOPs of a 16x16x32 one: 32 MFMAs per wave-tile instead of 64, the
// same LDS reads (16 K fragments, 32 transposed V^T reads), the same registers (S 32, O 64, Q 32, P 16), and a row's scores sit
// in TWO lanes (l, l ^ 32) instead of four, so the cross-lane maximum is one v_permlane32_swap.
//
// Fragment layouts (32x32x16): A[32 x 16] lane (l31, hh) = row l31, k 8 hh .. + 7; B[16 x 32] lane = column l31, k 8 hh .. + 7;
// C[32 x 32] lane = column l31, element i = row (i & 3) + 8 (i >> 2) + 4 hh.  S^T = K Q^T (A = K rows, B = Q) leaves a lane with 16
// keys of one query row per 32-key block; P^T feeds the second product from the same registers (key order inside a 16-key step =
// the accumulator's own order, the transposed V^T reads pick their rows accordingly).  V image: 16-byte chunk c of key r at
// c ^ ((r & 3) << 2) (a half-wave's transposed read covers 4 rows x 64 B: 8 distinct 32-byte bank slots); K image as prefill_kernel.
#pragma once

namespace mojo {

template <typename T> struct pf_mfma32;
template <> struct pf_mfma32<bf16_t> {
  typedef bf16x8 frag;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct pf_mfma32<f16_t> {
  typedef f16x8 frag;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

template <typename T, int G /* q heads per kv head */, bool SPLIT = false /* key split, see PrefillArgs */>
__global__ __launch_bounds__(256, 2) void prefill_m32_kernel(PrefillArgs a) {
  typedef typename pf_mfma32<T>::frag frag;
  constexpr int QPB = 128 / G;               // query positions per workgroup
  extern __shared__ __attribute__((aligned(1024))) char smem_generic[];
  lds_c* smem = (lds_c*)smem_generic;

  // Longest first over the WHOLE grid: workgroups are dispatched in blockIdx order and a query block's key count grows
  // with its position, so the block index is the slow coordinate, descending, and (kv-head, sequence) the fast one.
  // A sawtooth order (one descending ramp per sequence and head) left 30-40 % of the wave slots idle behind the long
  // blocks of the last ramp.  Consecutive ids also land on consecutive XCDs, so with 8 kv-heads each XCD's L2 holds
  // the K/V of one head.
  const int inner = a.hkv * a.batch;
  // (key split: the slices of a block are consecutive workgroups; `wg` is the block's index in the unsplit order)
  const int ks = SPLIT ? static_cast<int>(blockIdx.x % a.ksplit) : 0;
  const int wg = SPLIT ? static_cast<int>(blockIdx.x / a.ksplit) : static_cast<int>(blockIdx.x);
  if (wg >= a.n_qb * inner) {
    if (SPLIT && ks != 0) return;
    // Trailing workgroups zero the padding tokens behind the last sequence (rows no sequence owns must read as zeros);
    // they sit at the end of the grid, i.e. in the tail of the launch, and replace a memset of the whole output.
    const int64_t t0 = max(static_cast<int64_t>(a.cu_q[a.batch]), (static_cast<int64_t>(wg) - a.n_qb * inner) * PF_ZERO_TOKENS);
    const int64_t t1 = min(a.total_tokens, (static_cast<int64_t>(wg) - a.n_qb * inner + 1) * PF_ZERO_TOKENS);
    const int64_t row_elems = static_cast<int64_t>(a.hq) * a.dim;            // dim % 8 == 0: 16-byte pieces
    typedef typename vec_of<T, 8>::type V8;
    V8 z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = static_cast<T>(0.f);
    for (int64_t i = t0 * row_elems + threadIdx.x * 8; i < t1 * row_elems; i += 256 * 8)
      *reinterpret_cast<V8*>(static_cast<T*>(a.out) + i) = z;
    return;
  }
  const int qb = a.n_qb - 1 - wg / inner;
  const int rem = wg % inner;
  // Blocks are dealt to XCDs, and inside an XCD to its shader engines, in strict rotation and in order: with the
  // sequence as a fixed coordinate every block of a long sequence lands on ONE engine, and while its two slots per CU are
  // full the blocks behind it wait although other engines are empty (measured on 16 ragged sequences: 272 workgroups
  // resident for the first 25 us of a 150 us launch, 91 CUs idle).  The sequence coordinate is therefore rotated by
  // one per query-block level, so a sequence's blocks walk over the engines.
  const int kvh = rem % a.hkv, b = (rem / a.hkv + a.skew * (wg / inner)) % a.batch;
  const int q_start = a.cu_q[b];
  const int q_len = a.cu_q[b + 1] - q_start;
  const int kv_len = a.cu_kv ? a.cu_kv[b + 1] - a.cu_kv[b] : q_len;
  // rows [pos0, pos1) of this sequence, the G heads of this kv-head, written as zeros
  auto zero_rows = [&](int pos0, int pos1) {
    typedef typename vec_of<T, 8>::type V8;
    V8 z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = static_cast<T>(0.f);
    const int chunks8 = a.dim / 8;
    for (int i = threadIdx.x; i < (pos1 - pos0) * G * chunks8; i += 256) {
      const int c = i % chunks8, g = (i / chunks8) % G, pos = pos0 + i / (chunks8 * G);
      const int head = a.abab ? g * a.hkv + kvh : kvh * G + g;
      *reinterpret_cast<V8*>(static_cast<T*>(a.out) + (static_cast<int64_t>(q_start + pos) * a.hq + head) * a.dim + c * 8) = z;
    }
  };
  // A sequence longer than the caller's max_q_len hint has rows no query block of this launch covers: they are written
  // as zeros (never left uninitialised) by the workgroup of the sequence's last covered block.
  if (qb == a.n_qb - 1 && q_len > a.n_qb * QPB && ks == 0) zero_rows(a.n_qb * QPB, q_len);
  if (qb * QPB >= q_len) return;
  if (kv_len <= 0) {                                     // a sequence without keys: its rows read as zeros
    if (ks == 0) zero_rows(qb * QPB, min(q_len, (qb + 1) * QPB));
    return;
  }
  const int offset = kv_len - q_len;                     // query i sees keys 0 .. offset + i
#ifdef PF_WG_STAMPS
  const unsigned long long wg_t0 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x < 8192 * 4) {
    g_pf_stamps[blockIdx.x * 16 + 6] = static_cast<unsigned>(__builtin_amdgcn_s_memrealtime());   // 100 MHz, chip-wide
    g_pf_stamps[blockIdx.x * 16 + 8] = __builtin_amdgcn_s_getreg((31 << 11) | 4);                 // HW_ID: where it runs
    g_pf_stamps[blockIdx.x * 16 + 9] = __builtin_amdgcn_s_getreg((31 << 11) | 20);                // XCC_ID
  }
#endif

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int32_t* table = a.tables + static_cast<int64_t>(b) * a.table_stride;

  const int pos_hi = min(q_len, (qb + 1) * QPB) - 1;    // last query position of this block
  int kv_hi = min(kv_len, offset + pos_hi + 1);          // keys [0, kv_hi) are visible to some row
  if (kv_hi < 1) kv_hi = 1;
  const int n_kb_all = (kv_hi + PF_KEYS - 1) / PF_KEYS;
  // key tiles [kb_lo, n_kb) of the block belong to this workgroup (unsplit: all of them).  The host chose `ksplit` from the
  // CAPACITY of a sequence (it reads no length); the block itself knows how many tiles it walks, and cuts them into no more
  // slices than leave 8 tiles (512 keys) each: a short block in a split launch would otherwise pay partial writes and a merge
  // over slices of one or two tiles.  Surplus slices publish "no keys seen" and leave.
  int eff = 1;
  if constexpr (SPLIT) {
    eff = n_kb_all / 8;
    eff = eff < 1 ? 1 : (eff > a.ksplit ? a.ksplit : eff);
    if (ks >= eff) {
      if (threadIdx.x < 128) {
        a.ws_ml[(static_cast<int64_t>(blockIdx.x) * 128 + threadIdx.x) * 2 + 0] = -INFINITY;
        a.ws_ml[(static_cast<int64_t>(blockIdx.x) * 128 + threadIdx.x) * 2 + 1] = 0.f;
      }
      return;
    }
  }
  const int kb_lo = SPLIT ? static_cast<int>(static_cast<int64_t>(n_kb_all) * ks / eff) : 0;
  const int n_kb = SPLIT ? static_cast<int>(static_cast<int64_t>(n_kb_all) * (ks + 1) / eff) : n_kb_all;

  // The prologue is a chain of dependent memory round trips (1.5-2 us each on a busy chip) in front of a workgroup that
  // may own only a handful of tiles, so it is kept to two: {Q fragments, page-id window} together, then the first tile.
  // ---- this wave's rows: ONE 32-row block; lane (l31, h) = row l31, k-half h; row -> (head g, query position) ----------
  const int l31 = lane & 31, hh = lane >> 5;
  int row_pos, row_head;
  frag qf[8];                                            // B operand of S^T = K Q^T: 8 query dims 16 ks + 8 hh .. + 7 of row l31
  {
    const int r = wave * 32 + l31;
    const int g = r / QPB;
    int pos = qb * QPB + (r % QPB);
    row_head = a.abab ? g * a.hkv + kvh : kvh * G + g;
    row_pos = pos;
    if (pos >= q_len) pos = q_len - 1;                   // clamp: computed, never stored
    const T* qptr = static_cast<const T*>(a.q) + (static_cast<int64_t>(q_start + pos) * a.hq + row_head) * a.dim + hh * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const frag*>(qptr + ks * 16);
  }

  // A window of PF_TABLE page ids of this sequence lives in LDS (refilled when the key loop walks past it), so the
  // staging code never issues a dependent global load in front of its LDS-DMA — and never a FLAT load, which hipcc
  // emits for "LDS or global" pointer selects and guards with vmcnt(0)/lgkmcnt(0), draining the whole pipeline.
  // The same pass finds the first negative page id (golden: rows behind it read as zero K/V).
  int* s_table = reinterpret_cast<int*>(smem_generic + 4 * PF_TILE_BYTES);
  int win_base = 0;
  auto fill_window = [&](int p0) {
    for (int i = threadIdx.x; i < PF_TABLE; i += 256) s_table[i] = (p0 + i < a.max_pages) ? table[p0 + i] : -1;
    win_base = p0;
    __syncthreads();
  };
  int first_neg_key = 0x7fffffff;
  {
    int p1 = (kv_hi + a.page - 1) / a.page;
    int fn = 0x7fffffff;
    if (p1 > a.max_pages) { fn = a.max_pages; p1 = a.max_pages; }
    int* s_fn = s_table + PF_TABLE;
    if (threadIdx.x == 0) *s_fn = 0x7fffffff;
    int ids[PF_TABLE / 256];
#pragma unroll
    for (int j = 0; j < PF_TABLE / 256; ++j) {
      const int i = threadIdx.x + j * 256;
      ids[j] = i < a.max_pages ? table[i] : -1;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PF_TABLE / 256; ++j) {
      const int i = threadIdx.x + j * 256;
      s_table[i] = ids[j];
      if (ids[j] < 0 && i < p1) atomicMin(s_fn, i);
    }
    __syncthreads();
    const int wfn = *s_fn;
    if (wfn != 0x7fffffff) {
      fn = wfn;
    } else {
      for (int base = PF_TABLE; base < p1; base += 64) {   // contexts past the first window (rare, and long enough to amortise it)
        const int idx = base + lane;
        const int v = idx < p1 ? table[idx] : 0;
        const unsigned long long neg = __ballot(v < 0);
        if (neg) { fn = base + __builtin_ctzll(neg); break; }
      }
    }
    if (fn != 0x7fffffff) first_neg_key = fn * a.page;
  }

  // ---- staging ------------------------------------------------------------------------------------------
  // wave w fills keys [16w, 16w+16) of a tile with 4 LDS-DMA instructions per tensor (4 keys x 256 B each);
  // lane l: key l/16 of the four, LDS chunk position l%16
  const T* kbase = static_cast<const T*>(a.kc) + kvh * a.c_head;
  const T* vbase = static_cast<const T*>(a.vc) + kvh * a.c_head;
  const int chunks = a.dim / 8;
  auto stage = [&](int kb, int buf) {
    {  // wave-uniform for the whole workgroup: all waves stage the same key block
      const int k_last = min(kb * PF_KEYS + PF_KEYS - 1, kv_hi - 1);
      const int p_last = a.page_shift >= 0 ? (k_last >> a.page_shift) : k_last / a.page;
      if (p_last >= win_base + PF_TABLE) {
        __syncthreads();
        fill_window(a.page_shift >= 0 ? ((kb * PF_KEYS) >> a.page_shift) : (kb * PF_KEYS) / a.page);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kl = wave * 16 + i * 4 + (lane >> 4);                       // key inside the tile
      int key = kb * PF_KEYS + kl;
      if (key >= kv_hi) key = kv_hi - 1;
      const int lp = a.page_shift >= 0 ? (key >> a.page_shift) : key / a.page;
      int phys = s_table[lp - win_base];
      if (phys < 0) phys = 0;                                               // value is masked later
      const int64_t row = static_cast<int64_t>(phys) * a.c_blk + static_cast<int64_t>(key - lp * a.page) * a.c_tok;
      const int cp = lane & 15;
      int ck = cp ^ (kl & 15);
      int cv = cp ^ ((kl & 3) << 2);                      // (V image of the 32x32 formulation: see the header)
      if (ck >= chunks) ck = chunks - 1;
      if (cv >= chunks) cv = chunks - 1;
      lds_c* dk = smem + buf * 2 * PF_TILE_BYTES + (wave * 16 + i * 4) * 256;
      lds_c* dv = dk + PF_TILE_BYTES;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kbase + row + ck * 8),
                                       (__attribute__((address_space(3))) void*)dk, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vbase + row + cv * 8),
                                       (__attribute__((address_space(3))) void*)dv, 16, 0, 0);
    }
  };

  // Fast staging for tiles whose 64 keys all exist (every tile in front of the diagonal): with pages of >= 16 keys the 16
  // keys a wave stages share one page, so the page id is a SCALAR load issued a whole tile ahead, the row base is scalar
  // arithmetic and the per-lane byte offsets (key inside the 16, swizzled chunk) are loop invariants: no vector integer
  // multiplies or LDS table reads per tile.
  unsigned voff_k[4], voff_v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int kl = wave * 16 + i * 4 + (lane >> 4);
    const int cp = lane & 15;
    int ck = cp ^ (kl & 15);
    int cv = cp ^ ((kl & 3) << 2);                      // (V image of the 32x32 formulation: see the header)
    if (ck >= chunks) ck = chunks - 1;
    if (cv >= chunks) cv = chunks - 1;
    const unsigned rowb = static_cast<unsigned>((i * 4 + (lane >> 4)) * static_cast<int>(a.c_tok)) * sizeof(T);
    voff_k[i] = rowb + ck * 16;
    voff_v[i] = rowb + cv * 16;
  }
  // The page id is requested by an asm scalar load (hipcc sinks a load it can see to the point of use, i.e. behind the
  // DMA instructions, and then waits for it in front of the MFMAs) and retired by page_ready() at the top of the next tile.
  auto page_of_tile = [&](int kb) -> int {               // scalar: page id of this wave's 16 keys of tile kb
    int lp = (kb * PF_KEYS + wave * 16) >> a.page_shift;
    lp = min(lp, a.max_pages - 1);
    const int32_t* p = table + lp;
    int v;
    asm volatile("s_load_dword %0, %1, 0x0" : "=s"(v) : "s"(p) : "memory");
    return v;
  };
  auto page_ready = [&](int& v) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v) : : "memory"); };
  // split in two so that the scalar address arithmetic (which waits for the page id's scalar load) runs before the K
  // fragment reads are requested and the DMA instructions behind them
  auto stage_fast_base = [&](int kb, int phys) -> int64_t {
    const int key_w = kb * PF_KEYS + wave * 16;
    return (static_cast<int64_t>(phys) * a.c_blk + static_cast<int64_t>(key_w & (a.page - 1)) * a.c_tok) * static_cast<int64_t>(sizeof(T));
  };
  // piece i of the 8 DMA instructions of a tile: K (even i) or V (odd i) of the 4 keys i/2 of this wave's 16
  auto stage_fast_piece = [&](int buf, int64_t sb, int i) {
    const char* src = reinterpret_cast<const char*>((i & 1) ? vbase : kbase) + sb + ((i & 1) ? voff_v[i >> 1] : voff_k[i >> 1]);
    lds_c* dst = smem + buf * 2 * PF_TILE_BYTES + (i & 1) * PF_TILE_BYTES + (wave * 16 + (i >> 1) * 4) * 256;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };

  // ---- state ----------------------------------------------------------------------------------------------
  f32x16 o[4];                                           // O^T[32-d block]: element i of lane (l31, hh) = d 32 db + (i & 3) + 8 (i >> 2) + 4 hh of row l31
  float m = -INFINITY, lsum = 0.f;
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[db][e] = 0.f;

  const float lazy_raw = PF_LAZY_LOG2 / a.scale_log2;    // the lag in raw-score units (scale > 0)
  const unsigned smem_u32 = static_cast<unsigned>(reinterpret_cast<size_t>(smem));
  // K fragment (A operand, 32 keys x 16 d): key row l31 (+ 32 per key block), chunk 2 ks + hh, swizzled with the row
  unsigned koff[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) koff[ks] = l31 * 256 + (((2 * ks + hh) ^ (l31 & 15)) * 16);
  // V^T fragment (A operand, 32 d x 16 keys) of d block db: two transposed reads, keys 16 f + 4 hh + tq and + 8
  unsigned voff[4];
  {
    const int l15 = lane & 15, tq = l15 >> 2, tp = l15 & 3, gi = (lane >> 4) & 1;
#pragma unroll
    for (int db = 0; db < 4; ++db) voff[db] = (4 * hh + tq) * 256 + ((4 * (db ^ tq) + 2 * gi + (tp >> 1)) * 16) + (tp & 1) * 8;
  }

  // leading key blocks that every row of this workgroup sees completely need no masking at all; the diagonal /
  // tail / hole blocks run the masked variant.  Two loops, so neither carries the other's state.
  const int n_full = min(min(min(kv_len, offset + qb * QPB + 1), first_neg_key) / PF_KEYS, n_kb);
  const int n_fast = a.fast_stage ? n_full : 0;          // tiles [0, n_fast) may be staged the fast way

  if (kb_lo < n_kb) stage(kb_lo, kb_lo & 1);
  int phys_next = a.fast_stage ? page_of_tile(kb_lo + 1) : 0;    // page id for the NEXT stage, loaded a tile ahead
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  auto key_block = [&](auto masked_tag, auto fast_tag, int kb) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    constexpr bool FAST = decltype(fast_tag)::value;     // the NEXT tile is complete and staged the fast way
    const int buf = kb & 1;
    int64_t stage_base = 0;
    if constexpr (FAST) {
      page_ready(phys_next);
      stage_base = stage_fast_base(kb + 1, phys_next);
      asm volatile("" : "+s"(stage_base));               // computed here, not sunk behind the reads
      phys_next = page_of_tile(kb + 2);                  // lands long before the next tile asks for it
    }
    const lds_c* kt = smem + buf * 2 * PF_TILE_BYTES;
    const unsigned vt = smem_u32 + buf * 2 * PF_TILE_BYTES + PF_TILE_BYTES;

    // ---- S^T = K Q^T : two 32-key blocks, 8 k-steps each; all K fragments requested before the first MFMA ----------------
    frag kf[2][8];
#pragma unroll
    for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
        kf[kbk][ks] = *reinterpret_cast<const __attribute__((address_space(3))) frag*>(kt + kbk * 8192 + koff[ks]);
    if constexpr (!FAST) {
      if (kb + 1 < n_kb) stage(kb + 1, buf ^ 1);
    }
    auto dma_piece = [&](int i) {
      if constexpr (FAST) {
        stage_fast_piece(buf ^ 1, stage_base, i);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    __builtin_amdgcn_sched_barrier(0);
    f32x16 s[2];
#pragma unroll
    for (int kbk = 0; kbk < 2; ++kbk) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s[kbk][e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) s[kbk] = pf_mfma32<T>::run(kf[kbk][ks], qf[ks], s[kbk]);
    }
    // ---- V^T fragments: transposed reads in two batches of two d blocks (8 reads per d block: read n = keys 8 n + 4 hh .. + 3).
    // Each batch is one asm statement (issue) + one wait statement naming every destination.
    auto issue_v = [&](s16x4 (&dst)[16], int db0) {
      const unsigned a0 = vt + voff[db0], a1 = vt + voff[db0 + 1];
      asm volatile(
          "ds_read_b64_tr_b16 %0, %16\n\tds_read_b64_tr_b16 %1, %16 offset:2048\n\tds_read_b64_tr_b16 %2, %16 offset:4096\n\tds_read_b64_tr_b16 %3, %16 offset:6144\n\t"
          "ds_read_b64_tr_b16 %4, %16 offset:8192\n\tds_read_b64_tr_b16 %5, %16 offset:10240\n\tds_read_b64_tr_b16 %6, %16 offset:12288\n\tds_read_b64_tr_b16 %7, %16 offset:14336\n\t"
          "ds_read_b64_tr_b16 %8, %17\n\tds_read_b64_tr_b16 %9, %17 offset:2048\n\tds_read_b64_tr_b16 %10, %17 offset:4096\n\tds_read_b64_tr_b16 %11, %17 offset:6144\n\t"
          "ds_read_b64_tr_b16 %12, %17 offset:8192\n\tds_read_b64_tr_b16 %13, %17 offset:10240\n\tds_read_b64_tr_b16 %14, %17 offset:12288\n\tds_read_b64_tr_b16 %15, %17 offset:14336"
          : "=&v"(dst[0]), "=&v"(dst[1]), "=&v"(dst[2]), "=&v"(dst[3]), "=&v"(dst[4]), "=&v"(dst[5]), "=&v"(dst[6]), "=&v"(dst[7]),
            "=&v"(dst[8]), "=&v"(dst[9]), "=&v"(dst[10]), "=&v"(dst[11]), "=&v"(dst[12]), "=&v"(dst[13]), "=&v"(dst[14]), "=&v"(dst[15])
          : "v"(a0), "v"(a1)
          : "memory");
    };
    auto retire_v = [&](s16x4 (&dst)[16]) {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(dst[0]), "+v"(dst[1]), "+v"(dst[2]), "+v"(dst[3]), "+v"(dst[4]), "+v"(dst[5]), "+v"(dst[6]), "+v"(dst[7]),
                     "+v"(dst[8]), "+v"(dst[9]), "+v"(dst[10]), "+v"(dst[11]), "+v"(dst[12]), "+v"(dst[13]), "+v"(dst[14]), "+v"(dst[15])
                   : : "memory");
    };
    // lane (l31, hh) holds, for query row l31, keys  kb*64 + 32 kbk + (e & 3) + 8 (e >> 2) + 4 hh
    const int key0 = kb * PF_KEYS + 4 * hh;
    const bool has_hole = MASKED && (kb + 1) * PF_KEYS > first_neg_key;
    if constexpr (MASKED) {
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = key0 + 32 * kbk + (e & 3) + 8 * (e >> 2);
          if (has_hole && key >= first_neg_key) s[kbk][e] = 0.f;                     // zero K rows: score 0
          if (key > offset + row_pos || key >= kv_len) s[kbk][e] = -INFINITY;
        }
    }
    s16x4 vb0[16], vb1[16];
    issue_v(vb0, 0);
    // lazy reference maximum (as prefill_kernel): lane maxima, a wave-uniform branch on a ballot, then one lane-half swap
    float mx = fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[1][0], s[1][1]));
#pragma unroll
    for (int e = 2; e < 16; e += 2) mx = fmaxf(mx, fmaxf(fmaxf(s[0][e], s[0][e + 1]), fmaxf(s[1][e], s[1][e + 1])));
    if (__any(mx > m + lazy_raw)) {                                                  // m = -inf: any finite score triggers
      float p = mx, q = mx;
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(p), "+v"(q));
      mx = fmaxf(fmaxf(p, q), m);
      const float ms_new = (mx == -INFINITY ? 0.f : mx) * a.scale_log2;
      const float alpha = fast_exp2(m * a.scale_log2 - ms_new);
      m = mx;
      lsum *= alpha;
#pragma unroll
      for (int db = 0; db < 4; ++db) o[db] *= alpha;
    }
    const float ms = (m == -INFINITY ? 0.f : m) * a.scale_log2;
    float ps = 0.f;
    frag pf[2][2];                                                                   // [32-key block][16-key step]
#pragma unroll
    for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        frag f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float p = fast_exp2(fmaf(s[kbk][8 * st + j], a.scale_log2, -ms));
          ps += p;
          if constexpr (MASKED) {                                                    // zero V rows: no contribution
            if (has_hole && key0 + 32 * kbk + 16 * st + (j & 3) + 8 * (j >> 2) >= first_neg_key) p = 0.f;
          }
          f[j] = static_cast<T>(p);
          if (j == 3 || j == 7) dma_piece((kbk * 2 + st) * 2 + (j >> 2));
        }
        pf[kbk][st] = f;
      }
    lsum += ps;

    // ---- O^T += V^T P^T : d blocks 0-1 on batch 0, 2-3 on batch 1 -------------------------------------------------------
    auto pv_batch = [&](const s16x4 (&src)[16], int db0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          const s16x4 lo = src[i * 8 + 2 * f], hi = src[i * 8 + 2 * f + 1];
          const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          o[db0 + i] = pf_mfma32<T>::run(__builtin_bit_cast(frag, both), pf[f >> 1][f & 1], o[db0 + i]);
        }
    };
    retire_v(vb0);
    issue_v(vb1, 2);                                     // into the registers the scores just vacated
    pv_batch(vb0, 0);
    retire_v(vb1);
    pv_batch(vb1, 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // next tile landed
    __builtin_amdgcn_s_barrier();                         // ... and everyone is done reading this one
  };
  int kb_i = kb_lo;
  for (; kb_i + 1 < n_fast; ++kb_i) key_block(std::false_type{}, std::true_type{}, kb_i);     // the hot loop
  if (a.fast_stage) page_ready(phys_next);               // retire the last request before its register is reused
  for (; kb_i < n_full; ++kb_i) key_block(std::false_type{}, std::false_type{}, kb_i);
  for (; kb_i < n_kb; ++kb_i) key_block(std::true_type{}, std::false_type{}, kb_i);

  // ---- finish: row sums over the two lane halves, normalise, store ------------------------------------------------------
  float ls;
  {
    float p = lsum, q = lsum;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(p), "+v"(q));
    ls = p + q;
  }
  if constexpr (SPLIT) {
    // un-normalised partials of this key slice: row r of the block (= wave * 32 + l31), dims 32 db + 8 j + 4 hh .. + 3
    const int64_t item = static_cast<int64_t>(blockIdx.x);
    const int r = wave * 32 + l31;
    float* po = a.ws_o + (item * 128 + r) * a.dim + 4 * hh;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<f32x4*>(po + 32 * db + 8 * j) = f32x4{o[db][4 * j], o[db][4 * j + 1], o[db][4 * j + 2], o[db][4 * j + 3]};
    if (hh == 0) {
      a.ws_ml[(item * 128 + r) * 2 + 0] = m * a.scale_log2;         // (-inf for a slice that saw no visible key)
      a.ws_ml[(item * 128 + r) * 2 + 1] = ls;
    }
    return;
  }
  constexpr int OROW = 272;
  lds_c* stage_o = smem + wave * (32 * OROW);
  typedef typename vec_of<T, 4>::type V4;
  {
    const float inv = 1.0f / ls;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        V4 ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) ov[r] = static_cast<T>(o[db][4 * j + r] * inv);
        *reinterpret_cast<__attribute__((address_space(3))) V4*>(stage_o + l31 * OROW + (32 * db + 8 * j + 4 * hh) * 2) = ov;
      }
  }
  {
    typedef typename vec_of<T, 8>::type V8;
    const int sub = lane >> 4, ch = lane & 15;           // 4 rows per store instruction, 16 bytes per lane
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = i * 4 + sub;
      const int r = wave * 32 + row;
      const int pos = qb * QPB + (r % QPB);
      if (pos >= q_len) continue;
      const int g = r / QPB;
      const int head = a.abab ? g * a.hkv + kvh : kvh * G + g;
      const V8 v = *reinterpret_cast<const __attribute__((address_space(3))) V8*>(stage_o + row * OROW + ch * 16);
      *reinterpret_cast<V8*>(static_cast<T*>(a.out) + (static_cast<int64_t>(q_start + pos) * a.hq + head) * a.dim + ch * 8) = v;
    }
  }
}

}  // namespace mojo
