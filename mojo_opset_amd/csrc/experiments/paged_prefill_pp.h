// prefill_pp_kernel: MojoPagedPrefillGQA with the two waves of every SIMD held in OPPOSITE phases (round 3).
// Included by paged_prefill_gqa.hip behind prefill_kernel (shares PrefillArgs, pf_mfma, the tile constants and the formulation).
//
// prefill_kernel runs two independent 4-wave workgroups per CU.  A wave's tile is QK^T (matrix pipe) -> softmax (vector
// unit, a long dependent chain) -> PV (matrix pipe), and nothing orders the two workgroups of a CU against each other: the
// matrix pipe is busy 0.51 of the time on long sequences (profiles/r2_attention_counters.json), i.e. about what two
// free-running streams of that shape overlap by chance.  Here ONE 8-wave workgroup owns 256 rows (G heads x 256 / G query
// positions); waves 0-3 (group A) and 4-7 (group B) sit pairwise on the four SIMDs, and a tile is cut into two phases
// delimited by workgroup barriers:
//
//     M(k) = PV(k-1) ; QK^T(k)        matrix pipe + LDS fragment reads
//     V(k) = softmax(k) + this wave's four LDS-DMA pieces of a tile three ahead      vector unit
//
// Group B runs ONE phase behind group A, so in every slot each SIMD has one wave in M and one in V:
//
//     slot      0      1      2      3      4    ...
//     A        M(0)   V(0)   M(1)   V(1)   M(2)
//     B         -     M(0)   V(0)   M(1)   V(1)
//
// K tiles are staged by group A's waves, V tiles by group B's (four 1-KiB pieces per wave and tile instead of eight), into
// rings of four 16-KiB buffers each: K(k) is read in slots 2k (A) and 2k+1 (B), V(k) in slots 2k+2 and 2k+3; tile k+3 is
// requested during V(k), into the buffer whose last reader finished a slot earlier, and the counted waits (vmcnt(8): the two
// newest tiles may still be in flight) sit at the end of the phase in front of the barrier that publishes the tile.
// Page ids come from scalar loads one issue ahead; tiles at or behind the diagonal clamp their source rows per lane to the
// last visible key (rows past the sequence's end may hold anything, and 0 x NaN is NaN).  Pages must be powers of two of
// >= 16 keys; everything else (and the key-split form) stays on prefill_kernel.
#pragma once

namespace mojo {

constexpr int PP_RING = 4;
constexpr int PP_TABLE = 4096;               // page ids of the sequence kept in LDS: the launch needs max_blocks_per_seq <= PP_TABLE
constexpr int PP_LDS = 2 * PP_RING * PF_TILE_BYTES + PP_TABLE * 4 + 16;      // K ring | V ring | page ids | first negative page

template <typename T, int G /* q heads per kv head */, int DK /* head_dim / 32 */>
__global__ __launch_bounds__(512) void prefill_pp_kernel(PrefillArgs a) {
  typedef typename pf_mfma<T>::frag frag;
  constexpr int QPB = 256 / G;               // query positions per workgroup
  constexpr int DT = DK * 2;                 // 16-wide d tiles
  extern __shared__ __attribute__((aligned(1024))) char smem_generic[];
  lds_c* smem = (lds_c*)smem_generic;

  const int inner = a.hkv * a.batch;
  const int wg = static_cast<int>(blockIdx.x);
  if (wg >= a.n_qb * inner) {                // trailing workgroups zero the padding tokens behind the last sequence
    const int64_t t0 = max(static_cast<int64_t>(a.cu_q[a.batch]), (static_cast<int64_t>(wg) - a.n_qb * inner) * PF_ZERO_TOKENS);
    const int64_t t1 = min(a.total_tokens, (static_cast<int64_t>(wg) - a.n_qb * inner + 1) * PF_ZERO_TOKENS);
    const int64_t row_elems = static_cast<int64_t>(a.hq) * a.dim;
    typedef typename vec_of<T, 8>::type V8;
    V8 z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = static_cast<T>(0.f);
    for (int64_t i = t0 * row_elems + threadIdx.x * 8; i < t1 * row_elems; i += 512 * 8)
      *reinterpret_cast<V8*>(static_cast<T*>(a.out) + i) = z;
    return;
  }
  const int qb = a.n_qb - 1 - wg / inner;    // longest blocks first over the whole launch (see prefill_kernel)
  const int rem = wg % inner;
  const int kvh = rem % a.hkv, b = (rem / a.hkv + a.skew * (wg / inner)) % a.batch;
  const int q_start = a.cu_q[b];
  const int q_len = a.cu_q[b + 1] - q_start;
  const int kv_len = a.cu_kv ? a.cu_kv[b + 1] - a.cu_kv[b] : q_len;
  auto zero_rows = [&](int pos0, int pos1) {
    typedef typename vec_of<T, 8>::type V8;
    V8 z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = static_cast<T>(0.f);
    const int chunks8 = a.dim / 8;
    for (int i = threadIdx.x; i < (pos1 - pos0) * G * chunks8; i += 512) {
      const int c = i % chunks8, g = (i / chunks8) % G, pos = pos0 + i / (chunks8 * G);
      const int head = a.abab ? g * a.hkv + kvh : kvh * G + g;
      *reinterpret_cast<V8*>(static_cast<T*>(a.out) + (static_cast<int64_t>(q_start + pos) * a.hq + head) * a.dim + c * 8) = z;
    }
  };
  if (qb == a.n_qb - 1 && q_len > a.n_qb * QPB) zero_rows(a.n_qb * QPB, q_len);
  if (qb * QPB >= q_len) return;
  if (kv_len <= 0) {
    zero_rows(qb * QPB, min(q_len, (qb + 1) * QPB));
    return;
  }
  const int offset = kv_len - q_len;         // query i sees keys 0 .. offset + i

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pg = wave >> 2, w4 = wave & 3;   // phase group (0 = A, 1 = B: one phase behind), wave inside the group
  const int grp = lane >> 4, l15 = lane & 15;
  const int32_t* table = a.tables + static_cast<int64_t>(b) * a.table_stride;

  const int pos_hi = min(q_len, (qb + 1) * QPB) - 1;
  int kv_hi = min(kv_len, offset + pos_hi + 1);          // keys [0, kv_hi) are visible to some row
  if (kv_hi < 1) kv_hi = 1;
  const int n_kb = (kv_hi + PF_KEYS - 1) / PF_KEYS;

  // ---- this wave's rows: two 16-row tiles of the group's 128; row -> (head g, query position) ----------------------
  int row_pos[2];
  const T* qptr[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int r = pg * 128 + w4 * 32 + qt * 16 + l15;
    const int g = r / QPB;
    int pos = qb * QPB + (r % QPB);
    const int head = a.abab ? g * a.hkv + kvh : kvh * G + g;
    row_pos[qt] = pos;
    if (pos >= q_len) pos = q_len - 1;                   // clamp: computed, never stored
    qptr[qt] = static_cast<const T*>(a.q) + (static_cast<int64_t>(q_start + pos) * a.hq + head) * a.dim;
  }
  frag qf[2][DK];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int ks = 0; ks < DK; ++ks) qf[qt][ks] = *reinterpret_cast<const frag*>(qptr[qt] + ks * 32 + grp * 8);

  // ---- the sequence's page ids go to LDS once (max_pages <= PP_TABLE); the same pass finds the first negative one (golden:
  // rows behind it read as zero K/V).  Page ids are NOT fetched by asynchronous scalar loads here: a value that lands in a
  // scalar register a phase after the instruction that requested it does not survive the copies hipcc makes across a loop
  // back-edge (the copy reads the register before the load has written it).
  int* s_table = reinterpret_cast<int*>(smem_generic + 2 * PP_RING * PF_TILE_BYTES);
  int first_neg_key = 0x7fffffff;
  {
    int p1 = (kv_hi + a.page - 1) >> a.page_shift;
    int fn = 0x7fffffff;
    if (p1 > a.max_pages) { fn = a.max_pages; p1 = a.max_pages; }
    int* s_fn = s_table + PP_TABLE;
    if (threadIdx.x == 0) *s_fn = 0x7fffffff;
    __syncthreads();
    for (int i = threadIdx.x; i < p1; i += 512) {
      const int id = table[i];
      s_table[i] = id;
      if (id < 0) atomicMin(s_fn, i);
    }
    __syncthreads();
    const int wfn = *s_fn;
    if (wfn != 0x7fffffff) fn = wfn;
    if (fn != 0x7fffffff) first_neg_key = fn * a.page;
  }

  // ---- staging: group A's waves stage K, group B's stage V; wave w4 owns keys [16 w4, 16 w4 + 16) of every tile ----------
  const char* tbase = reinterpret_cast<const char*>(pg == 0 ? a.kc : a.vc) + static_cast<int64_t>(kvh) * a.c_head * static_cast<int64_t>(sizeof(T));
  const int chunks = a.dim / 8;
  const int tok_bytes = static_cast<int>(a.c_tok) * static_cast<int>(sizeof(T));
  // piece i covers LDS rows kl = 16 w4 + 4 i + (lane >> 4); its swizzled source chunk is  cp ^ (kl & 15)  for K and
  // cp ^ ((kl & 7) << 1)  for V (cp = lane & 15), i.e. a per-lane base XOR a compile-time constant of the piece: two registers
  // (chunk base, row offset of the lane's key inside a piece) instead of eight loop-invariant offsets that hipcc would spill
  // — and a spill reload inside the loop carries a vmcnt(0) that drains the whole DMA ring.
  const unsigned cbase = pg == 0 ? ((lane & 15) ^ (lane >> 4)) : ((lane & 15) ^ ((lane >> 4) << 1));
  const unsigned gtok = static_cast<unsigned>((lane >> 4) * tok_bytes);
  auto piece_chunk = [&](int i) -> unsigned {            // 16-byte chunk index of piece i's source
    unsigned c = cbase ^ static_cast<unsigned>(pg == 0 ? (i * 4) : ((i & 1) * 8));
    if (DK == 3) c = min(c, static_cast<unsigned>(chunks - 1));
    return c;
  };
  const int last_group = (kv_hi - 1) & ~15;              // first key of the 16-key group that holds the last visible key
  auto group_key = [&](int t) -> int { return min(t * PF_KEYS + w4 * 16, last_group); };   // scalar
  auto page_of_tile = [&](int t) -> int {                // page id of this wave's 16 keys of tile t (a uniform LDS read)
    int lp = group_key(t) >> a.page_shift;
    lp = min(lp, a.max_pages - 1);
    return __builtin_amdgcn_readfirstlane(s_table[lp]);
  };
  auto stage_base = [&](int t, int phys) -> int64_t {    // byte offset of the group's first row
    const int key_w = group_key(t);
    return (static_cast<int64_t>(max(phys, 0)) * a.c_blk + static_cast<int64_t>(key_w & (a.page - 1)) * a.c_tok) * static_cast<int64_t>(sizeof(T));
  };
  lds_c* const ring = smem + pg * (PP_RING * PF_TILE_BYTES);
  // piece i of tile t; clamp: the tile may hold rows past the last visible key (source rows clamped per lane)
  auto stage_piece = [&](int t, int64_t sb, int i, bool clamp) {
    unsigned off = static_cast<unsigned>(i * 4 * tok_bytes) + gtok + piece_chunk(i) * 16;
    if (clamp) {
      const int lim = kv_hi - 1 - group_key(t);          // >= 0
      off = static_cast<unsigned>(min(i * 4 + (lane >> 4), lim) * tok_bytes) + piece_chunk(i) * 16;
    }
    lds_c* dst = ring + (t & (PP_RING - 1)) * PF_TILE_BYTES + (w4 * 16 + i * 4) * 256;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tbase + sb + off),
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };

  // ---- state ------------------------------------------------------------------------------------------------------------
  f32x4 o[2][DT];
  float m[2], lsum[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    m[qt] = -INFINITY;
    lsum[qt] = 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float lazy_raw = PF_LAZY_LOG2 / a.scale_log2;
  const unsigned smem_u32 = static_cast<unsigned>(reinterpret_cast<size_t>(smem));
  const int tq = l15 >> 2, tp = l15 & 3;
  const unsigned kbase_lane = smem_u32 + l15 * 256 + ((grp ^ l15) & 15) * 16;
  // V^T read of d tile dt: row 4 grp + tq, chunk (2 dt + (tp >> 1)) ^ ((row & 7) << 1) = ((tp >> 1) ^ ((row & 7) << 1)) ^ 2 dt
  const unsigned vbase_lane = smem_u32 + PP_RING * PF_TILE_BYTES + (4 * grp + tq) * 256 + (((tp >> 1) ^ (((4 * grp + tq) & 7) << 1)) * 16) + (tp & 1) * 8;
  // leading key tiles every row of the workgroup sees completely
  const int n_full = min(min(min(kv_len, offset + qb * QPB + 1), first_neg_key) / PF_KEYS, n_kb);

  // prologue: tiles 0 .. 2 (the ring's fourth buffer is requested during V(0))
#pragma unroll 1
  for (int t = 0; t < min(3, n_kb); ++t) {
    const int64_t sb = stage_base(t, page_of_tile(t));
#pragma unroll
    for (int i = 0; i < 4; ++i) stage_piece(t, sb, i, t >= n_full);
  }
  // K(0) (group A) must have landed before slot 0; the later tiles may stay in flight
  if (n_kb >= 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  f32x4 s[2][4];
  frag pf[2][2];                                          // probabilities of the tile whose PV is pending: [q tile][32-key step]

  // ---- QK^T(k): S^T = K Q^T, 4 key tiles x 2 q tiles; the fragment reads are a step of their own so that a phase can request
  // them under the PV product's MFMAs ------------------------------------------------------------------------------------------
  auto qk_reads = [&](int k, frag (&kf)[4][DK], int t0, int t1) {      // key sub-tiles [t0, t1) of the four
    // fragment (t, ks): row 16 t + l15, chunk (4 ks + grp) ^ l15 = (grp ^ l15) ^ 4 ks: one lane base, XOR 64 ks, + 4096 t
    unsigned kb_lane = kbase_lane + (k & (PP_RING - 1)) * PF_TILE_BYTES;
    asm volatile("" : "+v"(kb_lane));                    // (keeps the derived addresses inside the phase)
#pragma unroll
    for (int ks = 0; ks < DK; ++ks) {
      const unsigned ad = kb_lane ^ static_cast<unsigned>(ks * 64);
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (t >= t0 && t < t1)
          kf[t][ks] = *reinterpret_cast<const __attribute__((address_space(3))) frag*>(static_cast<uintptr_t>(ad + t * 4096));
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto qk_mfma = [&](const frag (&kf)[4][DK]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      s[0][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      s[1][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < DK; ++ks) {
        s[0][t] = pf_mfma<T>::run(kf[t][ks], qf[0][ks], s[0][t]);
        s[1][t] = pf_mfma<T>::run(kf[t][ks], qf[1][ks], s[1][t]);
      }
    }
  };

  // ---- PV(k): O^T += V^T P^T; V^T through transposed reads in two batches of DT / 2 d tiles -------------------------------------
  constexpr int HB = DT / 2 * 4;
  auto issue_v = [&](s16x4 (&dst)[16], int k, int dt0) {
    {
      unsigned vb_lane = vbase_lane + (k & (PP_RING - 1)) * PF_TILE_BYTES;
      asm volatile("" : "+v"(vb_lane));
      unsigned ad[4];
#pragma unroll
      for (int i = 0; i < DT / 2; ++i) ad[i] = vb_lane ^ static_cast<unsigned>((dt0 + i) * 32);
      if constexpr (DT / 2 == 4) {
        asm volatile(
            "ds_read_b64_tr_b16 %0, %16\n\tds_read_b64_tr_b16 %1, %16 offset:4096\n\tds_read_b64_tr_b16 %2, %16 offset:8192\n\tds_read_b64_tr_b16 %3, %16 offset:12288\n\t"
            "ds_read_b64_tr_b16 %4, %17\n\tds_read_b64_tr_b16 %5, %17 offset:4096\n\tds_read_b64_tr_b16 %6, %17 offset:8192\n\tds_read_b64_tr_b16 %7, %17 offset:12288\n\t"
            "ds_read_b64_tr_b16 %8, %18\n\tds_read_b64_tr_b16 %9, %18 offset:4096\n\tds_read_b64_tr_b16 %10, %18 offset:8192\n\tds_read_b64_tr_b16 %11, %18 offset:12288\n\t"
            "ds_read_b64_tr_b16 %12, %19\n\tds_read_b64_tr_b16 %13, %19 offset:4096\n\tds_read_b64_tr_b16 %14, %19 offset:8192\n\tds_read_b64_tr_b16 %15, %19 offset:12288"
            : "=&v"(dst[0]), "=&v"(dst[1]), "=&v"(dst[2]), "=&v"(dst[3]), "=&v"(dst[4]), "=&v"(dst[5]), "=&v"(dst[6]), "=&v"(dst[7]),
              "=&v"(dst[8]), "=&v"(dst[9]), "=&v"(dst[10]), "=&v"(dst[11]), "=&v"(dst[12]), "=&v"(dst[13]), "=&v"(dst[14]), "=&v"(dst[15])
            : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3])
            : "memory");
      } else if constexpr (DT / 2 == 3) {
        asm volatile(
            "ds_read_b64_tr_b16 %0, %12\n\tds_read_b64_tr_b16 %1, %12 offset:4096\n\tds_read_b64_tr_b16 %2, %12 offset:8192\n\tds_read_b64_tr_b16 %3, %12 offset:12288\n\t"
            "ds_read_b64_tr_b16 %4, %13\n\tds_read_b64_tr_b16 %5, %13 offset:4096\n\tds_read_b64_tr_b16 %6, %13 offset:8192\n\tds_read_b64_tr_b16 %7, %13 offset:12288\n\t"
            "ds_read_b64_tr_b16 %8, %14\n\tds_read_b64_tr_b16 %9, %14 offset:4096\n\tds_read_b64_tr_b16 %10, %14 offset:8192\n\tds_read_b64_tr_b16 %11, %14 offset:12288"
            : "=&v"(dst[0]), "=&v"(dst[1]), "=&v"(dst[2]), "=&v"(dst[3]), "=&v"(dst[4]), "=&v"(dst[5]), "=&v"(dst[6]), "=&v"(dst[7]),
              "=&v"(dst[8]), "=&v"(dst[9]), "=&v"(dst[10]), "=&v"(dst[11])
            : "v"(ad[0]), "v"(ad[1]), "v"(ad[2])
            : "memory");
      } else {
        asm volatile(
            "ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:4096\n\tds_read_b64_tr_b16 %2, %8 offset:8192\n\tds_read_b64_tr_b16 %3, %8 offset:12288\n\t"
            "ds_read_b64_tr_b16 %4, %9\n\tds_read_b64_tr_b16 %5, %9 offset:4096\n\tds_read_b64_tr_b16 %6, %9 offset:8192\n\tds_read_b64_tr_b16 %7, %9 offset:12288"
            : "=&v"(dst[0]), "=&v"(dst[1]), "=&v"(dst[2]), "=&v"(dst[3]), "=&v"(dst[4]), "=&v"(dst[5]), "=&v"(dst[6]), "=&v"(dst[7])
            : "v"(ad[0]), "v"(ad[1])
            : "memory");
      }
    }
  };
  // `younger`: LDS reads requested behind this batch that may stay in flight (LDS operations of a wave return in order)
  auto retire_v = [&](s16x4 (&dst)[16], auto younger_tag) {
    {
      constexpr int YOUNGER = decltype(younger_tag)::value;
      if constexpr (HB == 16) {
        asm volatile("s_waitcnt lgkmcnt(%c[n])"
                     : "+v"(dst[0]), "+v"(dst[1]), "+v"(dst[2]), "+v"(dst[3]), "+v"(dst[4]), "+v"(dst[5]), "+v"(dst[6]), "+v"(dst[7]),
                       "+v"(dst[8]), "+v"(dst[9]), "+v"(dst[10]), "+v"(dst[11]), "+v"(dst[12]), "+v"(dst[13]), "+v"(dst[14]), "+v"(dst[15])
                     : [n] "i"(YOUNGER) : "memory");
      } else if constexpr (HB == 12) {
        asm volatile("s_waitcnt lgkmcnt(%c[n])"
                     : "+v"(dst[0]), "+v"(dst[1]), "+v"(dst[2]), "+v"(dst[3]), "+v"(dst[4]), "+v"(dst[5]), "+v"(dst[6]), "+v"(dst[7]),
                       "+v"(dst[8]), "+v"(dst[9]), "+v"(dst[10]), "+v"(dst[11])
                     : [n] "i"(YOUNGER) : "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(%c[n])"
                     : "+v"(dst[0]), "+v"(dst[1]), "+v"(dst[2]), "+v"(dst[3]), "+v"(dst[4]), "+v"(dst[5]), "+v"(dst[6]), "+v"(dst[7])
                     : [n] "i"(YOUNGER) : "memory");
      }
    }
  };
  auto pv_batch = [&](const s16x4 (&src)[16], int dt0) {
    {
#pragma unroll
      for (int i = 0; i < DT / 2; ++i)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const s16x4 lo = src[i * 4 + kk * 2], hi = src[i * 4 + kk * 2 + 1];
          const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          const frag vf = __builtin_bit_cast(frag, both);
          o[0][dt0 + i] = pf_mfma<T>::run(vf, pf[0][kk], o[0][dt0 + i]);
          o[1][dt0 + i] = pf_mfma<T>::run(vf, pf[1][kk], o[1][dt0 + i]);
        }
    }
  };
  s16x4 vb0[16];                                         // first V^T batch of the pending PV: requested at the end of the V phase

  // ---- softmax(k) -> pf, with this wave's four DMA pieces of tile k + 3 issued between its vector instructions ------------------
  auto softmax = [&](auto masked_tag, int k) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    const int t_new = k + 3;
    const bool do_stage = t_new < n_kb;                  // wave-uniform
    int64_t sb = 0;
    if (do_stage) {
      sb = stage_base(t_new, page_of_tile(t_new));
      asm volatile("" : "+s"(sb));
    }
    const bool clamp = t_new >= n_full;
    auto dma_piece = [&](int i) {
      if (do_stage) stage_piece(t_new, sb, i, clamp);
      __builtin_amdgcn_sched_barrier(0);
    };
    const int key0 = k * PF_KEYS + 4 * grp;
    const bool has_hole = MASKED && (k + 1) * PF_KEYS > first_neg_key;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      f32x4 (&sc)[4] = s[qt];
      if constexpr (MASKED) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = key0 + 16 * t + r;
            if (has_hole && key >= first_neg_key) sc[t][r] = 0.f;                   // zero K rows: score 0
            if (key > offset + row_pos[qt] || key >= kv_len) sc[t][r] = -INFINITY;
          }
      }
      float mx = fmaxf(fmaxf(sc[0][0], sc[0][1]), fmaxf(sc[0][2], sc[0][3]));
#pragma unroll
      for (int t = 1; t < 4; ++t) mx = fmaxf(mx, fmaxf(fmaxf(sc[t][0], sc[t][1]), fmaxf(sc[t][2], sc[t][3])));
      if (__any(mx > m[qt] + lazy_raw)) {                                            // m = -inf: any finite score triggers
        mx = xor_max_16_32(mx);
        mx = fmaxf(mx, m[qt]);
        const float ms_new = (mx == -INFINITY ? 0.f : mx) * a.scale_log2;
        const float alpha = fast_exp2(m[qt] * a.scale_log2 - ms_new);
        m[qt] = mx;
        lsum[qt] *= alpha;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[qt][dt] *= alpha;
      }
      const float ms = (m[qt] == -INFINITY ? 0.f : m[qt]) * a.scale_log2;
      float ps = 0.f;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        frag f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float p0 = fast_exp2(fmaf(sc[2 * kk][r], a.scale_log2, -ms));
          float p1 = fast_exp2(fmaf(sc[2 * kk + 1][r], a.scale_log2, -ms));
          ps += p0 + p1;
          if constexpr (MASKED) {                                                    // zero V rows: no contribution
            if (has_hole && key0 + 32 * kk + r >= first_neg_key) p0 = 0.f;
            if (has_hole && key0 + 32 * kk + 16 + r >= first_neg_key) p1 = 0.f;
          }
          f[r] = static_cast<T>(p0);
          f[4 + r] = static_cast<T>(p1);
          if (r == 3) dma_piece(qt * 2 + kk);
        }
        pf[qt][kk] = f;
      }
      lsum[qt] += ps;
    }
  };

  // ---- the phases -----------------------------------------------------------------------------------------------------------
  // Barrier j of group A and barrier j of group B are the same barrier; A: prologue, after M(0), after V(0), after M(1), ...,
  // after M(n), final; B: prologue, leading, after M(0), after V(0), ..., after M(n).
  auto wait_v_landed = [&](int k) {                      // group B, end of V(k - 1): V(k) is read from the end of the next slot on
    if (pg == 1) {
      if (k + 3 <= n_kb) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };
  auto wait_k_landed = [&](int k) {                      // group A, end of V(k): K(k + 1) is read from the next slot on
    if (pg == 0) {
      if (k + 4 <= n_kb) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };
  // V(k) | M(k + 1).  The first V^T batch of PV(k) is requested at the END of V(k), in front of the barrier (tile V(k) was
  // published a slot earlier), the second batch and the K fragments of QK^T(k + 1) under the first batch's MFMAs: the matrix
  // phase starts with its operands in registers instead of two exposed LDS round trips.
#ifdef PF_STAMPS
  unsigned tacc = 0;
  unsigned long long t_prev = __builtin_readcyclecounter();
#endif
  auto tile = [&](auto masked_tag, int k) {
    constexpr bool HOT = !decltype(masked_tag)::value;
    if constexpr (HOT) { PF_STAMP(0); }
    softmax(masked_tag, k);
    if constexpr (HOT) { PF_STAMP(1); }
    issue_v(vb0, k, 0);
    if (pg == 0) wait_k_landed(k); else wait_v_landed(k + 1);
    if constexpr (HOT) { PF_STAMP(2); }
    __builtin_amdgcn_s_barrier();
    if constexpr (HOT) { PF_STAMP(3); }
    s16x4 vb1[16];
    frag kf[4][DK];
    const bool more = k + 1 < n_kb;                      // wave-uniform
    retire_v(vb0, std::integral_constant<int, 0>{});
    issue_v(vb1, k, DT / 2);
    pv_batch(vb0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (HOT) { PF_STAMP(4); }
    retire_v(vb1, std::integral_constant<int, 0>{});
    pv_batch(vb1, DT / 2);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (HOT) { PF_STAMP(5); }
    if (more) {
      qk_reads(k + 1, kf, 0, 4);
      if constexpr (HOT) { PF_STAMP(6); }
      qk_mfma(kf);
    }
    if constexpr (HOT) { PF_STAMP(7); }
    __builtin_amdgcn_s_barrier();
    if constexpr (HOT) { PF_STAMP(8); }
  };
  if (pg == 1) __builtin_amdgcn_s_barrier();             // group B starts one phase behind
  {
    frag kf[4][DK];
    qk_reads(0, kf, 0, 4);
    qk_mfma(kf);
  }
  __builtin_amdgcn_s_barrier();
  int k = 0;
#pragma unroll 1
  for (; k < n_full; ++k) tile(std::false_type{}, k);    // the hot loop: tiles every row sees completely
#ifdef PF_STAMPS
  if (blockIdx.x < 2048) {
    if (lane < 15) g_pf_stamps[(blockIdx.x * 8 + wave) * 16 + lane] = tacc;
    if (lane == 15) g_pf_stamps[(blockIdx.x * 8 + wave) * 16 + 15] = static_cast<unsigned>(k);
  }
#endif
#pragma unroll 1
  for (; k < n_kb; ++k) tile(std::true_type{}, k);       // diagonal / tail / hole tiles
  if (pg == 0) __builtin_amdgcn_s_barrier();             // (group B's M(n)); behind it the rings become the output staging

  // ---- finish: reduce the row sums over the 4 lane groups, normalise, transpose through LDS, store whole rows --------------
  constexpr int OROW = 272;                       // (68 dwords: the 16 rows of a write land 4 banks apart; 288 left rows l and l + 8 on one bank pair)
  lds_c* stage_o = smem + wave * (32 * OROW);
  typedef typename vec_of<T, 4>::type V4;
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const float inv = 1.0f / xor_sum_16_32(lsum[qt]);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      V4 ov;
#pragma unroll
      for (int r = 0; r < 4; ++r) ov[r] = static_cast<T>(o[qt][dt][r] * inv);
      *reinterpret_cast<__attribute__((address_space(3))) V4*>(stage_o + (qt * 16 + l15) * OROW + (dt * 16 + grp * 4) * 2) = ov;
    }
  }
  {
    typedef typename vec_of<T, 8>::type V8;
    constexpr int CPR = DT * 2;
    constexpr int RPI = 64 / CPR;
    const int sub = lane / CPR, ch = lane % CPR;
#pragma unroll
    for (int i = 0; i < (32 + RPI - 1) / RPI; ++i) {
      const int row = i * RPI + sub;
      if (sub >= RPI || row >= 32) continue;
      const int r = pg * 128 + w4 * 32 + row;
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
