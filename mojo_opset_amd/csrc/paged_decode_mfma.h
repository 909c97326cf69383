// decode_mfma_kernel: MojoPagedDecodeGQA with the two contractions on the matrix cores (round 3).
// Included by paged_decode_gqa.hip (shares DecodeArgs, the chunking rules, decode_head, the merge kernel and the launch forms).
//
// The vector-unit kernel (decode_split_kernel) spends ~600 vector instructions per 16-token tile for four query heads and
// ~1 070 for eight (dot products, DPP butterflies, exponentials and the P V sums, all of them per head): with two waves per
// SIMD that is a latency-bound stream — groups of eight query heads (Llama-3-70B: 64 / 8) ran at 3.0 TB/s however the K/V
// bytes were shared.  Here a tile costs ~60 vector instructions whatever the group size:
//
//   S^T[16 tokens x 16 heads] = K Q^T      v_mfma_f32_16x16x32: A = the K registers AS LOADED, and loaded in WHOLE 128-byte lines (a
//                                          load instruction = 8 tokens x one line; 64-byte runs stream at about half the rate):
//                                          lane l holds chunk (l >> 3) of token (l & 7), which the MFMA sees as row
//                                          (token, parity p = chunk & 1) with k-group g = chunk >> 1.  Row (token, p) meets the
//                                          right query dims only in the product whose B operand holds the chunks 2 g + p: every
//                                          register feeds TWO products (p = 0, 1), each valid in 8 of its 16 rows, and
//                                          S[token] = C0[row token] + C1[row token + 8] — one v_permlane32_swap per score
//                                          register puts two 8-token halves side by side in exactly the layout of the 16-row form.
//                                          B = the query slices (head = l & 15; lanes past the group repeat its last head)
//   online softmax                         lane = (head, tokens 4 g .. 4 g + 3 of every tile): in-lane maximum, one cross-group
//                                          maximum (two lane-row swaps), lazy reference (rescale O only when it grows by 2^8)
//   O^T[D x 16 heads] += V^T P^T           v_mfma_f32_16x16x16: B = the four probabilities of the lane (the accumulator layout of
//                                          S^T IS the B layout: no lane movement), A = V^T from a wave-private 4 KiB LDS image of
//                                          the tile read with ds_read_b64_tr_b16 (32-byte pairs XOR-swizzled by the token: the
//                                          eight row blocks of a half-wave land in eight bank slots)
//
// The LDS image belongs to ONE wave (LDS operations of a wave execute in order): no barrier anywhere in the loop.  Pages must
// hold a multiple of 16 tokens (a tile then lies in one page), head_dim 64 or 128, group size <= 16; everything else takes the
// vector-unit kernel.  Same chunking, pairing, in-LDS merge, workspace layout and hole / empty-row semantics as that kernel.
#pragma once

namespace mojo {

template <typename T> struct dec_mma;
template <> struct dec_mma<bf16_t> {
  typedef bf16x8 frag8;
  typedef s16x4 frag4;
  static __device__ __forceinline__ f32x4 qk(frag8 a, frag8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ f32x4 pv(frag4 a, frag4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ frag4 pack(float p0, float p1, float p2, float p3) {
    const bf16x4 v = {static_cast<bf16_t>(p0), static_cast<bf16_t>(p1), static_cast<bf16_t>(p2), static_cast<bf16_t>(p3)};
    return __builtin_bit_cast(frag4, v);
  }
  static __device__ __forceinline__ frag4 from_lds(s16x4 v) { return v; }
};
template <> struct dec_mma<f16_t> {
  typedef f16x8 frag8;
  typedef f16x4 frag4;
  static __device__ __forceinline__ f32x4 qk(frag8 a, frag8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ f32x4 pv(frag4 a, frag4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ frag4 pack(float p0, float p1, float p2, float p3) {
    const f16x4 v = {static_cast<f16_t>(p0), static_cast<f16_t>(p1), static_cast<f16_t>(p2), static_cast<f16_t>(p3)};
    return v;
  }
  static __device__ __forceinline__ frag4 from_lds(s16x4 v) { return __builtin_bit_cast(f16x4, v); }
};

constexpr int DECM_TILE = 16;                 // tokens of a sub-tile (one S^T product set, one V image)

template <typename T, int DK /* head_dim / 32 */, bool NT, int MODE>
__global__ __launch_bounds__(MODE != DEC_SPLIT ? 512 : 64) void decode_mfma_kernel(DecodeArgs a, int G) {
  constexpr bool FUSED = MODE != DEC_SPLIT;
  constexpr bool PAIRED = MODE == DEC_PAIRED;
  constexpr int D = DK * 32, ND = D / 16;               // head_dim, 16-wide d tiles of O^T
  constexpr int ROWB = D * 2;                           // bytes of a token row
  constexpr int NP = D / 16;                            // 32-byte pairs per row
  constexpr int RPB = 8 / NP;                           // rows per 256-byte bank row (1 at D = 128, 2 at D = 64)
  constexpr int NS = DK == 2 ? 2 : 1;                   // sub-tiles per loop step: a step moves 8 KiB of K/V whatever the head_dim (at
                                                        // head_dim 64 one sub-tile per step left the loop's fixed part — maximum, reference,
                                                        // rescale test, branches — on half the bytes: 0.68 of HBM against 0.80 at 128)
  constexpr int STEP = DECM_TILE * NS;
  typedef typename pack8<T>::vec V8;
  typedef dec_mma<T> MM;
  const int lane = threadIdx.x & 63;
  const int tl = lane & 15, g4 = lane >> 4;             // K / V loads: token tl of the tile, dim chunk g4 of each k-step
  const int wave_id = FUSED ? __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6)) : 0;
  // (grouped form, DecodeArgs::fuse_group: workgroup x of a row owns chunks [x * waves, (x + 1) * waves) of it)
  int chunk = FUSED ? static_cast<int>(blockIdx.x) * static_cast<int>(blockDim.x >> 6) + wave_id : static_cast<int>(blockIdx.x);
  int b = blockIdx.y / a.hkv;
  const int kvh = blockIdx.y % a.hkv;

  int seq_len, chunk_tokens;
  int pb[2] = {0, -1}, plen[2] = {0, 0}, pchunk[2] = {DEC_TILE, DEC_TILE}, n_first = 8;
  if constexpr (PAIRED) {                               // (identical to decode_split_kernel: the two launches must agree)
    const int cap = a.n_chunks * a.chunk_tokens;
    int len = -1;
    if (lane < a.batch) len = a.max_pages > 0 ? max(min(a.seq_lens[lane], cap), 0) : 0;
    int rank = lane;
    if (__ballot(lane < a.batch && len != __builtin_amdgcn_readfirstlane(len)) != 0) {
      rank = 0;
      for (int o = 0; o < a.batch; ++o) {
        const int lo = __builtin_amdgcn_readlane(len, o);
        rank += (lo > len || (lo == len && o < lane)) ? 1 : 0;
      }
    }
    const int p = b;
    const unsigned long long first = __ballot(lane < a.batch && rank == p);
    const unsigned long long second = __ballot(lane < a.batch && rank == a.batch - 1 - p && a.batch - 1 - p > p);
    pb[0] = __builtin_ctzll(first);
    plen[0] = __builtin_amdgcn_readlane(len, pb[0]);
    if (second) {
      pb[1] = __builtin_ctzll(second);
      plen[1] = __builtin_amdgcn_readlane(len, pb[1]);
    }
    const int sum = plen[0] + plen[1];
    n_first = plen[1] <= 0 ? 8 : min(max((8 * plen[0] + sum / 2) / sum, 1), 7);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int n_u = u ? 8 - n_first : n_first;
      int c = n_u > 0 ? (plen[u] + n_u - 1) / n_u : DEC_TILE;
      c = max(c, 128);
      pchunk[u] = ((c + DEC_TILE - 1) / DEC_TILE) * DEC_TILE;
    }
    const int u = wave_id < n_first ? 0 : 1;
    b = pb[u];
    chunk = u ? wave_id - n_first : wave_id;
    seq_len = b >= 0 ? plen[u] : 0;
    chunk_tokens = pchunk[u];
    if (b < 0) b = pb[0];
  } else {
    seq_len = a.max_pages > 0 ? decode_seq_len(a, b) : 0;
    chunk_tokens = decode_seq_chunk(a, seq_len);
  }
  const int tok_begin = chunk * chunk_tokens;
  const bool has_work = seq_len > 0 && tok_begin < seq_len;
  if (!FUSED && !has_work) return;
  const int tok_end = has_work ? min(seq_len, tok_begin + chunk_tokens) : tok_begin + 1;

  // query slices: B operand, lane = (head tl, dims 32 s + 8 g4 .. + 7)
  const int hq_l = min(tl, G - 1);                      // lanes past the group repeat its last head (computed, never stored)
  // B operands: line L (128 bytes = 8 chunks) of the row, parity p: lane k-group g4 holds the dims of chunk 8 L + 2 g4 + p
  constexpr int NL = D / 64;                            // 128-byte lines per token row
  typename MM::frag8 qf[NL][2];
  {
    const int h = decode_head(a, kvh, hq_l, G);
    const T* qp = static_cast<const T*>(a.q) + (static_cast<int64_t>(b) * a.hq + h) * a.dim;
#pragma unroll
    for (int L = 0; L < NL; ++L)
#pragma unroll
      for (int p = 0; p < 2; ++p) qf[L][p] = *reinterpret_cast<const typename MM::frag8*>(qp + (8 * L + 2 * g4 + p) * 8);
  }

  f32x4 o[ND];
#pragma unroll
  for (int dt = 0; dt < ND; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, l = 0.f;

  const int32_t* table = a.tables + static_cast<int64_t>(b) * a.table_stride;
  int p1 = (tok_end + a.page - 1) / a.page;
  int first_neg = 0x7fffffff;
  if (p1 > a.max_pages) { first_neg = a.max_pages; p1 = a.max_pages; }
  constexpr int SCAN = 4;
  int scan_v[SCAN];
  auto scan_issue = [&](int base) {
#pragma unroll
    for (int u = 0; u < SCAN; ++u) {
      const int idx = base + u * 64 + lane;
      scan_v[u] = idx < p1 ? table[idx] : 0;
    }
  };
  auto scan_reduce = [&](int base) {
#pragma unroll
    for (int u = 0; u < SCAN; ++u) {
      const unsigned long long neg = __ballot(scan_v[u] < 0);
      if (neg && first_neg == 0x7fffffff) first_neg = base + u * 64 + __builtin_ctzll(neg);
    }
  };
  if (has_work) scan_issue(0);

  // K: instruction (token group j, line L) = tokens 8 j + (l & 7), chunk 8 L + (l >> 3);  V: whole rows, RPI rows per instruction
  constexpr int CPR = D / 8;                            // 16-byte chunks per row
  constexpr int RPI = 64 / CPR;                         // V rows per load instruction (4 at D = 128, 8 at D = 64)
  constexpr int NV = 16 / RPI;                          // V load instructions per tile
  const int kt8 = lane & 7, kc8 = lane >> 3;
  const int vr = lane / CPR, vc = lane % CPR;
  const T* kbase = static_cast<const T*>(a.kc) + (kvh >> a.hshift) * a.c_head + kc8 * 8;
  const T* vbase = static_cast<const T*>(a.vc) + (kvh >> a.hshift) * a.c_head + vc * 8;
  const int last_tile = ((tok_end - 1) / DECM_TILE) * DECM_TILE;      // first token of the last non-empty tile
  const int last_page = a.max_pages - 1;

  struct Tile { V8 k[NS][2][NL]; V8 v[NS][NV]; int lp[NS]; };
  auto ld = [&](const T* p) -> V8 {
    if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const V8*>(p));
    else return *reinterpret_cast<const V8*>(p);
  };
  auto load_tile = [&](Tile& t, int t0) {
#pragma unroll
    for (int ss = 0; ss < NS; ++ss) {
      const int tu = min(t0 + DECM_TILE * ss, last_tile);   // wave-uniform; 16 | page: the sub-tile lies in one page
      const int lp = tu >> a.page_shift;
      t.lp[ss] = lp;
      const int phys = max(table[min(lp, last_page)], 0);
      const int64_t pg = static_cast<int64_t>(phys) * a.c_blk + static_cast<int64_t>(tu - (lp << a.page_shift)) * a.c_tok;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int L = 0; L < NL; ++L) t.k[ss][j][L] = ld(kbase + pg + static_cast<int64_t>(8 * j + kt8) * a.c_tok + L * 64);
#pragma unroll
      for (int u = 0; u < NV; ++u) t.v[ss][u] = ld(vbase + pg + static_cast<int64_t>(RPI * u + vr) * a.c_tok);
    }
  };

  // wave-private V images, one per sub-tile: [16 tokens][ROWB bytes], 32-byte pair pp of row t at pp ^ ((t / RPB) & (NP - 1))
  extern __shared__ float s_part[];                      // [waves][G][D + 2] partials, then [waves][16 x ROWB] V images
  const int n_waves = FUSED ? static_cast<int>(blockDim.x >> 6) : 1;
  char* const v_img = reinterpret_cast<char*>(s_part + (FUSED ? n_waves * G * (D + 2) : 0)) + wave_id * (NS * 16 * ROWB);
  const unsigned v_u32 = static_cast<unsigned>(reinterpret_cast<size_t>(v_img));
  unsigned w_off[NV];                                    // write of load instruction u: chunk vc of row RPI u + vr
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    const int row = RPI * u + vr;
    const int fw = (row / RPB) & (NP - 1);
    w_off[u] = row * ROWB + (((vc >> 1) ^ fw) << 5) + (vc & 1) * 16;
  }
  // transposed read of d tile dt: lane (group g4, i = tl): row = token 4 g4 + (i >> 2), columns 16 dt + 4 (i & 3) .. + 3
  const int rrow = 4 * g4 + (tl >> 2);
  const int fr = (rrow / RPB) & (NP - 1);
  const unsigned r_base = v_u32 + rrow * ROWB + (tl & 3) * 8;      // + ((dt ^ fr) << 5)

  auto process = [&](Tile& t, int t0) {
#pragma unroll
    for (int ss = 0; ss < NS; ++ss)
      if (t.lp[ss] >= first_neg) {                       // rare: pages behind a hole read as zeros
        V8 z = {};
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int L = 0; L < NL; ++L) t.k[ss][j][L] = z;
#pragma unroll
        for (int u = 0; u < NV; ++u) t.v[ss][u] = z;
      }
    const bool full = t0 + STEP <= tok_end;              // wave-uniform
    float x[NS][4];
#pragma unroll
    for (int ss = 0; ss < NS; ++ss) {
      // c[j][p]: token group j (8 tokens), parity p; valid rows: (token, p)
      f32x4 c[2][2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          c[j][p] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int L = 0; L < NL; ++L) c[j][p] = MM::qk(__builtin_bit_cast(typename MM::frag8, t.k[ss][j][L]), qf[L][p], c[j][p]);
        }
      // stage V while the scores come out of the matrix pipe (rows past the length may hold NaN / Inf: zeros)
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        const bool vrow_ok = full || (t0 + DECM_TILE * ss + RPI * u + vr) < tok_end;
        V8 z = {};
        *reinterpret_cast<V8*>(v_img + ss * (16 * ROWB) + w_off[u]) = vrow_ok ? t.v[ss][u] : z;
      }
      // S[token] = C0[row token] + C1[row token + 8]; rows 0-7 live in lanes 0-31, rows 8-15 in lanes 32-63 (row = 4 (l >> 4) + i).
      // Tokens 0-7 (group 0) end up in lanes 0-31, tokens 8-15 (group 1) in lanes 32-63: token 4 (l >> 4) + i, as in the 16-row form.
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // X = C1 of group 0 (its upper half is needed below), Y = C0 of group 1;  X' = [X.lo, Y.lo], Y' = [X.hi, Y.hi].
        // The builtin, not inline asm: the operands come straight out of the matrix pipe, and only the compiler's hazard
        // recogniser knows how many wait states an MFMA result needs before a lane swap may read it.
        // (floats first: __builtin_bit_cast applied to a vector-element lvalue reads element 0 whatever the index)
        const float xa = c[0][1][i], ya = c[1][0][i];
        const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, xa), __builtin_bit_cast(unsigned, ya), false, false);
        const float own = lane < 32 ? c[0][0][i] : c[1][1][i];
        const float oth = __builtin_bit_cast(float, lane < 32 ? sw[1] : sw[0]);
        x[ss][i] = (own + oth) * a.scale_log2;
      }
    }
    if (!full) {
#pragma unroll
      for (int ss = 0; ss < NS; ++ss)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (t0 + DECM_TILE * ss + 4 * g4 + i >= tok_end) x[ss][i] = -INFINITY;
    }
    float mx = fmaxf(fmaxf(x[0][0], x[0][1]), fmaxf(x[0][2], x[0][3]));
#pragma unroll
    for (int ss = 1; ss < NS; ++ss) mx = fmaxf(mx, fmaxf(fmaxf(x[ss][0], x[ss][1]), fmaxf(x[ss][2], x[ss][3])));
    mx = xor_max_16_32(mx);                              // the head's maximum over the step (all four token groups)
    float ref = m;
    if (mx - m > 8.0f) ref = mx;                         // m = -inf: any finite score; NaN (-inf - -inf): keep
    if (!__all(ref == m)) {
      const float alpha = m == ref ? 1.f : fast_exp2(m - ref);        // m = -inf: 0 (O and the sum are 0)
      l *= alpha;
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) o[dt] *= alpha;
      m = ref;
    }
    const float ms = m == -INFINITY ? 0.f : m;
    typename MM::frag4 pf[NS];
#pragma unroll
    for (int ss = 0; ss < NS; ++ss) {
      float p[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) p[i] = fast_exp2(x[ss][i] - ms);
      l += (p[0] + p[1]) + (p[2] + p[3]);
      pf[ss] = MM::pack(p[0], p[1], p[2], p[3]);
    }
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
      for (int ss = 0; ss < NS; ++ss) {
        const s16x4 vt = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            reinterpret_cast<__attribute__((address_space(3))) s16x4*>(static_cast<uintptr_t>(r_base + ss * (16 * ROWB) + ((dt ^ fr) << 5))));
        o[dt] = MM::pv(MM::from_lds(vt), pf[ss], o[dt]);
      }
  };

  Tile ta, tb, tc;
  if (has_work) {
    load_tile(ta, tok_begin);
    if (tok_begin + STEP < tok_end) load_tile(tb, tok_begin + STEP);
    scan_reduce(0);
    for (int base = 64 * SCAN; base < p1 && first_neg == 0x7fffffff; base += 64 * SCAN) {
      scan_issue(base);
      scan_reduce(base);
    }
    for (int t0 = tok_begin; t0 < tok_end; t0 += 3 * STEP) {
      if (t0 + 2 * STEP < tok_end) load_tile(tc, t0 + 2 * STEP);
      process(ta, t0);
      if (t0 + STEP >= tok_end) break;
      if (t0 + 3 * STEP < tok_end) load_tile(ta, t0 + 3 * STEP);
      process(tb, t0 + STEP);
      if (t0 + 2 * STEP >= tok_end) break;
      if (t0 + 4 * STEP < tok_end) load_tile(tb, t0 + 4 * STEP);
      process(tc, t0 + 2 * STEP);
    }
  }

  // the row sums of the four token groups of a head meet (the reference maximum is already common to them)
  l = xor_sum_16_32(l);
  // lane holds head tl, dims 16 dt + 4 g4 + i
  const bool head_ok = tl < G;
  if constexpr (FUSED) {
    const int stride = D + 2;
    if (head_ok) {
      float* dst = s_part + (wave_id * G + tl) * stride;
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) *reinterpret_cast<f32x4*>(dst + dt * 16 + 4 * g4) = o[dt];
      if (g4 == 0) { dst[D] = m; dst[D + 1] = l; }
    }
    __syncthreads();
    const int per_head = D / 4;
    typedef typename vec_of<T, 4>::type V4;
    constexpr int UNITS = PAIRED ? 2 : 1;
    for (int item = threadIdx.x; item < UNITS * G * per_head; item += blockDim.x) {
      const int u = item / (G * per_head);
      const int rest = item - u * (G * per_head);
      const int g = rest / per_head, d0 = (rest - g * per_head) * 4;
      int ub, ulen, uchunk, slot0, uwaves;
      if constexpr (PAIRED) {
        ub = pb[u]; ulen = plen[u]; uchunk = pchunk[u];
        slot0 = u ? n_first : 0;
        uwaves = u ? 8 - n_first : n_first;
        if (ub < 0) continue;
      } else {
        ub = b; ulen = seq_len; uchunk = chunk_tokens; slot0 = 0; uwaves = static_cast<int>(blockDim.x >> 6);
      }
      int n_chunks_seq = ulen <= 0 ? 0 : min((ulen + uchunk - 1) / uchunk, uwaves);
      bool partial = false;                                // grouped form: this workgroup's chunks are not the whole row
      if constexpr (!PAIRED) {
        if (a.fuse_group > 0) {
          const int total = ulen <= 0 ? 0 : (ulen + uchunk - 1) / uchunk;
          partial = total > uwaves;
          n_chunks_seq = min(max(total - static_cast<int>(blockIdx.x) * uwaves, 0), uwaves);
          if (blockIdx.x > 0 && n_chunks_seq == 0) continue;   // a workgroup past the row's last chunk: nothing to leave
        }
      }
      if (n_chunks_seq == 0 && a.leave_empty) continue;
      const int h = decode_head(a, kvh, g, G);
      float mx = -INFINITY;
      for (int c = 0; c < n_chunks_seq; ++c) mx = fmaxf(mx, s_part[((slot0 + c) * G + g) * stride + D]);
      f32x4 num = {0.f, 0.f, 0.f, 0.f};
      float den = 0.f;
      for (int c = 0; c < n_chunks_seq; ++c) {
        const float* src = s_part + ((slot0 + c) * G + g) * stride;
        const float w = exp2f(src[D] - mx);
        den = fmaf(w, src[D + 1], den);
        num += f32x4{src[d0], src[d0 + 1], src[d0 + 2], src[d0 + 3]} * w;
      }
      if (partial) {                                         // un-normalised sums against this workgroup's maximum, for the merge launch
        const int64_t slot = (static_cast<int64_t>(blockIdx.y) * a.n_chunks + blockIdx.x) * G + g;
        *reinterpret_cast<f32x4*>(a.ws_acc + slot * D + d0) = num;
        if (d0 == 0) { a.ws_ml[slot * 2 + 0] = mx; a.ws_ml[slot * 2 + 1] = den; }
        continue;
      }
      const float inv = n_chunks_seq > 0 ? 1.0f / den : 0.f;
      V4 ov;
#pragma unroll
      for (int e = 0; e < 4; ++e) ov[e] = static_cast<T>(num[e] * inv);
      *reinterpret_cast<V4*>(static_cast<T*>(a.out) + (static_cast<int64_t>(ub) * a.hq + h) * D + d0) = ov;
    }
    return;
  }
  if (!head_ok) return;
  const int n_chunks_seq = (seq_len + chunk_tokens - 1) / chunk_tokens;
  if (n_chunks_seq == 1) {                               // single chunk: finish here, the merge kernel skips this row
    const int h = decode_head(a, kvh, tl, G);
    const float inv = 1.0f / l;
    typedef typename vec_of<T, 4>::type V4;
    T* dst = static_cast<T*>(a.out) + (static_cast<int64_t>(b) * a.hq + h) * D + 4 * g4;
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) {
      V4 ov;
#pragma unroll
      for (int e = 0; e < 4; ++e) ov[e] = static_cast<T>(o[dt][e] * inv);
      *reinterpret_cast<V4*>(dst + dt * 16) = ov;
    }
    return;
  }
  const int64_t slot = (static_cast<int64_t>(blockIdx.y) * a.n_chunks + chunk) * G + tl;
  float* dst = a.ws_acc + slot * D + 4 * g4;
#pragma unroll
  for (int dt = 0; dt < ND; ++dt) *reinterpret_cast<f32x4*>(dst + dt * 16) = o[dt];
  if (g4 == 0) {
    a.ws_ml[slot * 2 + 0] = m;
    a.ws_ml[slot * 2 + 1] = l;
  }
}

}  // namespace mojo
