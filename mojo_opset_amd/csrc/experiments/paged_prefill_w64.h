// prefill_w64_kernel: MojoPagedPrefillGQA with ONE wave per SIMD and 64 query rows per wave (head_dim 128, pages >= 16 keys).
// Included by paged_prefill_gqa.hip behind prefill_kernel in an EXPERIMENTS build (MOJO_HIP_BUILD_EXPERIMENTS=1; shares PrefillArgs,
// lds_c, the block order, the zero-fill tail); selected per call with MOJO_HIP_PREFILL_W64=1.  Parity-green, measured slower than
// prefill_kernel (round 4: 1x16384 1 073 vs 1 100 TF, 4x2048 793 vs 894, 16 ragged 468 vs 662) — why, with the numbers: DESIGN §4.4.
//
// Why a second kernel: prefill_kernel runs two free-running 4-wave workgroups per CU; its matrix pipe is
// idle half of the time because a wave's softmax (vector unit) and its MFMAs overlap only with the OTHER workgroup's, by
// chance, and a 32-row wave-tile re-reads the whole K and V tile from LDS.  Here a workgroup is 4 waves = 256 rows (G heads x
// 256 / G positions), a wave owns 64 rows = two 32-row blocks on v_mfma_f32_32x32x16 (a K or V^T fragment feeds two 32-cycle
// MFMAs: half the LDS bytes per FLOP), has the SIMD's whole register file, and overlaps matrix and vector work INSIDE itself
// by skewing the loop one tile:
//
//     iteration t :  MFMA   PV(t-1)  [32]                         then   QK^T(t+1)  [32]
//                    VALU   softmax(t), rows 0-31                        softmax(t), rows 32-63;  lane maxima of S(t+1)
//                    LDS    V^T(t-1) fragments, first K / Q fragments     K(t+1) / Q fragments;  first V^T(t) fragments (next iteration)
//                    DMA    K(t+2), V(t+1): 8 pieces per wave
//     (this order keeps the fewest registers alive: S(t+1) only exists in the second half, P(t-1) only in the first)
//
// All 64 MFMAs of an iteration are independent of the iteration's vector work (S(t) and P(t-1) were finished one iteration
// earlier), so every MFMA gap carries one softmax element (v_fma + v_exp of element i, v_add + v_cvt_pk of element i - 1: a
// result of the transcendental unit is never consumed by the next instruction): the source is written gap by gap and pinned
// with sched_barrier.  LDS reads are inline asm, one or two groups ahead of their MFMAs, with counted lgkmcnt waits (hipcc puts vmcnt(0) in front of LDS reads it can see while
// LDS-DMA is in flight; page ids therefore come from an LDS table by ds_read, not from a scalar load that would share
// lgkmcnt out of order).  LDS: K ring of two 16-KiB tiles (K(t+2) is written while K(t+1) is read), V ring of three (V(t+1)
// is written while V(t-1) is read and V(t) waits — so the first fragments of the next iteration can be requested BEFORE the
// barrier), the workgroup's 256 query
// rows (64 KiB: Q fragments are re-read per k-step instead of occupying 64 registers — with O at 128 and two S tiles alive
// the register file has no room for them), the sequence's page ids.  One barrier per iteration.
//
// Softmax arithmetic per element is ONE transcendental: Q is stored pre-multiplied by scale * log2(e) (rounded to the storage
// type: a relative 2^-9 per query element, far inside the 2e-2 bound) and the S^T accumulators START at minus the row's
// reference: the first MFMA of a tile multiplies a "ones" A fragment (1 in column 0) with a B fragment holding -m in row 0 —
// the reference is kept exactly representable in the storage type, any value within the lag is a valid reference — so the
// matrix pipe delivers s * scale - m and the vector unit only computes 2^x, the row sum and the rounding.  (A register tuple
// holding -m as srcC of the first MFMA does the same without the extra MFMA per accumulator, but costs 32 arch VGPRs that
// the loop does not have: hipcc then spilled ~250 registers and moved fragment registers while their LDS reads were in flight.)  (Measured on this chip, scripts/probes/mfma_gap_probe.hip:
// one wave per SIMD hides THREE vector instructions under a 32-cycle MFMA — 33.5 cycles with v_fma v_exp v_add — and pays
// ~5 cycles for each further one, ~4 for an LDS read: fma + exp + add + cvt/2 + max/2 + one read ran at 49 cycles per MFMA.)
//
// Lazy reference maximum as in prefill_kernel (may lag by 2^8).  The decision for S(t) is taken at the top of iteration t on
// lane maxima computed in iteration t-1.  When it fires (first tiles, then rarely) the new reference is used by softmax(t)
// at once, while O and l — which still hold, or are still receiving, terms exponentiated against the OLD reference: P(t-1) is
// multiplied into O in this very iteration — are rescaled at the END of the iteration, behind the PV(t-1) MFMAs and
// before anything at the new scale is added (guide T13: scale everything at the old maximum exactly once).  (A separate
// sequential iteration for this case made hipcc spill ~350 registers in the hot loop; a uniform branch inside it does not.)
// Diagonal / tail / hole tiles run a plain masked loop after the pipeline drains.
//
// LDS images (rows of 256 B): K and Q chunk c of row r at c ^ (r & 15) (ds_read_b128 of the 32x32x16 A / B operand:
// conflict-free); V chunk c of key r at c ^ ((r & 3) << 2) (ds_read_b64_tr_b16: a half-wave's 4 rows x 64 B land in 8 distinct
// 32-byte bank slots).  Key order inside a 16-key step of PV: element j of lane half h is key 16 s + 8 (j >> 2) + 4 h + (j & 3)
// on both operands — the S^T accumulator registers 8 s .. 8 s + 7 ARE that order, the transposed reads pick their rows accordingly.
#pragma once

namespace mojo {

// MFMAs as inline asm with EXPLICIT register classes.  hipcc chooses one class for the C/D operands of every MFMA of a
// function; with O (128 registers) and two S tiles (128) alive it moved accumulators between the two halves of the register
// file by the hundred and spilled (first build: 688 spilled registers, 450 v_accvgpr moves per iteration).  Here:
//   S^T accumulate :  C/D in arch VGPRs (the softmax reads them with vector instructions), A = K fragment and B = Q fragment in AGPRs
//   O^T accumulate :  C/D in AGPRs, A = V^T fragment and B = P^T fragment in arch VGPRs
// The compiler does not see an MFMA in an asm statement, so it inserts no wait states between it and a dependent vector
// instruction: w64_settle_s / w64_settle_o (>= 18 wait states) stand wherever vector code reads accumulators right behind
// an MFMA — and they NAME the accumulators ("+v" / "+a" operands): a wait-state statement that does not mention them is
// no barrier for the compiler's own register reads, and hipcc hoisted the v_accvgpr_read of O above it (O read while the
// last MFMAs were still writing it: garbage in d blocks 2-3 whenever a reference update hit the interleaved loop).
// Accumulation chains (vdst = srcC of the previous MFMA on the same registers) need none.
template <typename T> struct w64_mfma;
template <> struct w64_mfma<bf16_t> {
  typedef bf16x8 frag;
  // (s_nop: the operands of the seeding MFMA are materialised by the compiler, typically with v_accvgpr_write / _mov right in
  //  front of the statement, and it knows of no MFMA here to keep them apart from: without the wait states the second seed of
  //  a pair read stale AGPRs — NaN rows 32-63 of every multi-tile block)
  static __device__ __forceinline__ void s_first(f32x16& c, const u32x4& a, const u32x4& b) {
    asm volatile("s_nop 4\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(c) : "a"(a), "a"(b));
  }
  static constexpr unsigned kOne = 0x3F80u;              // 1.0 in the storage type
  static __device__ __forceinline__ unsigned bits_of(float x) { return __builtin_bit_cast(unsigned short, static_cast<bf16_t>(x)); }
  static __device__ __forceinline__ float rounded(float x) { return static_cast<float>(static_cast<bf16_t>(x)); }
  static __device__ __forceinline__ void s_acc(f32x16& c, const u32x4& a, const u32x4& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "a"(a), "a"(b));
  }
  static __device__ __forceinline__ void o_acc(f32x16& c, const u32x4& a, const u32x4& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  }
};
template <> struct w64_mfma<f16_t> {
  typedef f16x8 frag;
  static __device__ __forceinline__ void s_first(f32x16& c, const u32x4& a, const u32x4& b) {
    asm volatile("s_nop 4\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(c) : "a"(a), "a"(b));
  }
  static constexpr unsigned kOne = 0x3C00u;
  static __device__ __forceinline__ unsigned bits_of(float x) { return __builtin_bit_cast(unsigned short, static_cast<f16_t>(x)); }
  static __device__ __forceinline__ float rounded(float x) { return static_cast<float>(static_cast<f16_t>(x)); }
  static __device__ __forceinline__ void s_acc(f32x16& c, const u32x4& a, const u32x4& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "a"(a), "a"(b));
  }
  static __device__ __forceinline__ void o_acc(f32x16& c, const u32x4& a, const u32x4& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  }
};
__device__ __forceinline__ void w64_settle_s(f32x16& s0, f32x16& s1, f32x16& s2, f32x16& s3) {
  asm volatile("s_nop 15\n\ts_nop 7" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : : "memory");
}
__device__ __forceinline__ void w64_settle_o(f32x16 (&o)[4][2]) {
  asm volatile("s_nop 15\n\ts_nop 7"
               : "+a"(o[0][0]), "+a"(o[0][1]), "+a"(o[1][0]), "+a"(o[1][1]), "+a"(o[2][0]), "+a"(o[2][1]), "+a"(o[3][0]), "+a"(o[3][1])
               : : "memory");
}

constexpr int W64_K_SLOTS = 2, W64_V_SLOTS = 3;
constexpr int W64_V_OFF = W64_K_SLOTS * PF_TILE_BYTES;                       // 32 KiB
constexpr int W64_Q_OFF = W64_V_OFF + W64_V_SLOTS * PF_TILE_BYTES;           // 80 KiB: [256 rows][256 B], swizzled like K
constexpr int W64_TABLE_OFF = W64_Q_OFF + 256 * 256;                         // 144 KiB
constexpr int W64_TABLE = 4080;                                              // block-table entries of the sequence held in LDS
constexpr int W64_LDS = W64_TABLE_OFF + W64_TABLE * 4 + 16;                  // 163 792 B of the CU's 163 840
constexpr int W64_OROW = 272;                                                // epilogue staging row pitch (bytes)

// ---- asm LDS reads: issue now, retire later with a counted wait that names every destination ---------------------------
// fragments of TWO k-steps of QK^T for one 32-key block (K addresses ka0 / ka1 + KOFF) and both 32-row blocks (Q addresses
// qa0 / qa1, + 8192): {K ks, K ks+1, Q ks rows 0-31, Q ks rows 32-63, Q ks+1 rows 0-31, Q ks+1 rows 32-63}
template <int KOFF>
__device__ __forceinline__ void w64_kq_issue(u32x4 (&d)[6], unsigned ka0, unsigned ka1, unsigned qa0, unsigned qa1) {
  asm volatile("ds_read_b128 %0, %6 offset:%10\n\tds_read_b128 %1, %7 offset:%10\n\t"
               "ds_read_b128 %2, %8\n\tds_read_b128 %3, %8 offset:8192\n\t"
               "ds_read_b128 %4, %9\n\tds_read_b128 %5, %9 offset:8192"
               : "=&a"(d[0]), "=&a"(d[1]), "=&a"(d[2]), "=&a"(d[3]), "=&a"(d[4]), "=&a"(d[5])
               : "v"(ka0), "v"(ka1), "v"(qa0), "v"(qa1), "i"(KOFF)
               : "memory");
}
template <int N>
__device__ __forceinline__ void w64_kq_retire(u32x4 (&d)[6]) {
  asm volatile("s_waitcnt lgkmcnt(%6)" : "+a"(d[0]), "+a"(d[1]), "+a"(d[2]), "+a"(d[3]), "+a"(d[4]), "+a"(d[5]) : "i"(N) : "memory");
}
// the V^T fragments of 16-key step F (keys 16 F .. 16 F + 15 of the tile) for the four 32-d blocks: two transposed reads each
// (keys 16 F + 4 h .. + 3 and + 8); a0..a3 = per-lane addresses of the d blocks inside the V tile
template <int F>
__device__ __forceinline__ void w64_v_issue(s16x4 (&d)[8], unsigned a0, unsigned a1, unsigned a2, unsigned a3) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %8 offset:%12\n\tds_read_b64_tr_b16 %1, %8 offset:%13\n\t"
      "ds_read_b64_tr_b16 %2, %9 offset:%12\n\tds_read_b64_tr_b16 %3, %9 offset:%13\n\t"
      "ds_read_b64_tr_b16 %4, %10 offset:%12\n\tds_read_b64_tr_b16 %5, %10 offset:%13\n\t"
      "ds_read_b64_tr_b16 %6, %11 offset:%12\n\tds_read_b64_tr_b16 %7, %11 offset:%13"
      : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7])
      : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "i"(F * 4096), "i"(F * 4096 + 2048)
      : "memory");
}
template <int N>
__device__ __forceinline__ void w64_v_retire(s16x4 (&d)[8]) {
  asm volatile("s_waitcnt lgkmcnt(%8)"
               : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])
               : "i"(N) : "memory");
}
// x[l] + x[l ^ 32] / max(x[l], x[l ^ 32]) in the vector unit
__device__ __forceinline__ float w64_xor32_sum(float x) {
  float p = x, q = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(p), "+v"(q));
  return p + q;
}
// max(a, b, c) as ONE instruction placed where it is written (fmaxf on MFMA results gets a canonicalising v_max per operand
// from hipcc, and un-pinned maxima were sunk behind the barrier into the next iteration's head)
__device__ __forceinline__ float w64_max3(float a, float b, float c) {
  float r;
  asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float w64_xor32_max(float x) {
  float p = x, q = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(p), "+v"(q));
  return fmaxf(p, q);
}

template <typename T, int G /* q heads per kv head */>
__global__ __launch_bounds__(256, 1) void prefill_w64_kernel(PrefillArgs a) {
  typedef typename w64_mfma<T>::frag frag;
  constexpr int QPB = 256 / G;               // query positions per workgroup
  extern __shared__ __attribute__((aligned(1024))) char smem_generic[];
  lds_c* smem = (lds_c*)smem_generic;

  // ---- block order, zero-fill tail, empty cases: as prefill_kernel (with 256-row blocks) -------------------------------
  const int inner = a.hkv * a.batch;
  const int wg = static_cast<int>(blockIdx.x);
  if (wg >= a.n_qb * inner) {
    const int64_t t0 = max(static_cast<int64_t>(a.cu_q[a.batch]), (static_cast<int64_t>(wg) - a.n_qb * inner) * PF_ZERO_TOKENS);
    const int64_t t1 = min(a.total_tokens, (static_cast<int64_t>(wg) - a.n_qb * inner + 1) * PF_ZERO_TOKENS);
    const int64_t row_elems = static_cast<int64_t>(a.hq) * a.dim;
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
  const int kvh = rem % a.hkv, b = (rem / a.hkv + a.skew * (wg / inner)) % a.batch;
  const int q_start = a.cu_q[b];
  const int q_len = a.cu_q[b + 1] - q_start;
  const int kv_len = a.cu_kv ? a.cu_kv[b + 1] - a.cu_kv[b] : q_len;
  auto zero_rows = [&](int pos0, int pos1) {
    typedef typename vec_of<T, 8>::type V8;
    V8 z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = static_cast<T>(0.f);
    for (int i = threadIdx.x; i < (pos1 - pos0) * G * 16; i += 256) {
      const int c = i % 16, g = (i / 16) % G, pos = pos0 + i / (16 * G);
      const int head = a.abab ? g * a.hkv + kvh : kvh * G + g;
      *reinterpret_cast<V8*>(static_cast<T*>(a.out) + (static_cast<int64_t>(q_start + pos) * a.hq + head) * 128 + c * 8) = z;
    }
  };
  if (qb == a.n_qb - 1 && q_len > a.n_qb * QPB) zero_rows(a.n_qb * QPB, q_len);
  if (qb * QPB >= q_len) return;
  if (kv_len <= 0) {
    zero_rows(qb * QPB, min(q_len, (qb + 1) * QPB));
    return;
  }
  const int offset = kv_len - q_len;                     // query i sees keys 0 .. offset + i

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int32_t* table = a.tables + static_cast<int64_t>(b) * a.table_stride;

  const int pos_hi = min(q_len, (qb + 1) * QPB) - 1;
  int kv_hi = min(kv_len, offset + pos_hi + 1);
  if (kv_hi < 1) kv_hi = 1;
  const int n_kb = (kv_hi + PF_KEYS - 1) / PF_KEYS;

  // ---- this wave's rows: two 32-row blocks; row -> (head g, query position).  Q goes to LDS in the layout of the B operand
  //      of S^T = K Q^T (16-byte chunk c of row r at c ^ (r & 15), as K): lane (l31, h) owns chunks 2 ks + h of its row -------
  int row_pos[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const int r = wave * 64 + rb * 32 + l31;
    const int g = r / QPB;
    int pos = qb * QPB + (r % QPB);
    row_pos[rb] = pos;
    if (pos >= q_len) pos = q_len - 1;                   // clamp: computed, never stored
    const int head = a.abab ? g * a.hkv + kvh : kvh * G + g;
    const T* qp = static_cast<const T*>(a.q) + (static_cast<int64_t>(q_start + pos) * a.hq + head) * 128 + h * 8;
    frag qv[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qv[ks] = *reinterpret_cast<const frag*>(qp + ks * 16);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      frag qs;                                           // q * (scale * log2 e), rounded to the storage type
#pragma unroll
      for (int e = 0; e < 8; ++e) qs[e] = static_cast<T>(static_cast<float>(qv[ks][e]) * a.scale_log2);
      *reinterpret_cast<__attribute__((address_space(3))) frag*>(smem + W64_Q_OFF + r * 256 + (((2 * ks + h) ^ (l31 & 15)) * 16)) = qs;
    }
  }

  // ---- the sequence's page ids in LDS; first negative id (rows behind it read as zero K / V, the golden's `break`) -------
  int* s_table = reinterpret_cast<int*>(smem_generic + W64_TABLE_OFF);
  int first_neg_key = 0x7fffffff;
  {
    int p1 = (kv_hi + a.page - 1) >> a.page_shift;
    int fn = 0x7fffffff;
    if (p1 > a.max_pages) { fn = a.max_pages; p1 = a.max_pages; }
    int* s_fn = s_table + W64_TABLE;
    if (threadIdx.x == 0) *s_fn = 0x7fffffff;
    __syncthreads();
    int my_fn = 0x7fffffff;
    for (int i = threadIdx.x; i < p1 + 16 && i < W64_TABLE; i += 256) {    // (+16: the page-id prefetch looks a few tiles ahead)
      const int v = i < a.max_pages ? table[i] : -1;
      s_table[i] = v;
      if (v < 0 && i < p1) my_fn = min(my_fn, i);
    }
    if (my_fn != 0x7fffffff) atomicMin(s_fn, my_fn);
    __syncthreads();
    const int wfn = *s_fn;
    if (wfn != 0x7fffffff) fn = wfn;
    if (fn != 0x7fffffff) first_neg_key = fn * a.page;
  }
  const int n_full = min(min(min(kv_len, offset + qb * QPB + 1), first_neg_key) / PF_KEYS, n_kb);

  // ---- staging: wave w fills keys [16 w, 16 w + 16) of a K or V tile, 4 pieces of 1 KiB each ------------------------------
  const T* kbase = static_cast<const T*>(a.kc) + kvh * a.c_head;
  const T* vbase = static_cast<const T*>(a.vc) + kvh * a.c_head;
  unsigned voff_k[4], voff_v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int kl = wave * 16 + i * 4 + (lane >> 4);
    const int cp = lane & 15;
    const unsigned rowb = static_cast<unsigned>((i * 4 + (lane >> 4)) * static_cast<int>(a.c_tok)) * sizeof(T);
    voff_k[i] = rowb + (cp ^ (kl & 15)) * 16;
    voff_v[i] = rowb + (cp ^ ((kl & 3) << 2)) * 16;
  }
  auto k_slot = [](int kb) -> unsigned { return static_cast<unsigned>(kb & 1) * PF_TILE_BYTES; };
  auto v_slot = [](int kb) -> unsigned { return W64_V_OFF + static_cast<unsigned>(kb % W64_V_SLOTS) * PF_TILE_BYTES; };
  // complete tiles: the wave's 16 keys share one page -> scalar row base (`sb`), loop-invariant lane offsets
  auto stage_piece = [&](bool is_v, unsigned slot, int64_t sb, int i) {
    const char* src = reinterpret_cast<const char*>(is_v ? vbase : kbase) + sb + (is_v ? voff_v[i] : voff_k[i]);
    lds_c* dst = smem + slot + (wave * 16 + i * 4) * 256;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };
  auto stage_base_of = [&](int kb, int phys) -> int64_t {
    const int key_w = kb * PF_KEYS + wave * 16;
    return (static_cast<int64_t>(phys) * a.c_blk + static_cast<int64_t>(key_w & (a.page - 1)) * a.c_tok) * static_cast<int64_t>(sizeof(T));
  };
  // any tile (partial tiles, holes): page id per 4 keys from the LDS table
  auto stage_general = [&](bool is_v, int kb, unsigned slot) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kl = wave * 16 + i * 4 + (lane >> 4);
      int key = kb * PF_KEYS + kl;
      if (key >= kv_hi) key = kv_hi - 1;
      const int lp = key >> a.page_shift;
      int phys = s_table[lp];
      if (phys < 0) phys = 0;                                               // value is masked later
      const int64_t row = static_cast<int64_t>(phys) * a.c_blk + static_cast<int64_t>(key - (lp << a.page_shift)) * a.c_tok;
      const int cp = lane & 15;
      const int c = is_v ? (cp ^ ((kl & 3) << 2)) : (cp ^ (kl & 15));
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((is_v ? vbase : kbase) + row + c * 8),
                                       (__attribute__((address_space(3))) void*)(smem + slot + (wave * 16 + i * 4) * 256), 16, 0, 0);
    }
  };
  auto page_of_tile_lds = [&](int kb) -> int {           // page id of this wave's 16 keys of tile kb (plain LDS read)
    int lp = (kb * PF_KEYS + wave * 16) >> a.page_shift;
    lp = min(lp, W64_TABLE - 1);
    return __builtin_amdgcn_readfirstlane(s_table[lp]);
  };
  auto stage_tile = [&](bool is_v, int kb) {             // prologue / sequential form (no interleaving)
    if (kb >= n_kb) return;
    const unsigned slot = is_v ? v_slot(kb) : k_slot(kb);
    if (kb < n_full) {
      const int64_t sb = stage_base_of(kb, page_of_tile_lds(kb));
#pragma unroll
      for (int i = 0; i < 4; ++i) stage_piece(is_v, slot, sb, i);
    } else {
      stage_general(is_v, kb, slot);
    }
  };

  // ---- per-lane LDS read addresses ---------------------------------------------------------------------------------------
  const unsigned smem_u32 = static_cast<unsigned>(reinterpret_cast<size_t>(smem));
  unsigned kaddr[8];                                     // K fragment of k-step ks: key row l31 (+ 32 per key block: immediate)
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) kaddr[ks] = smem_u32 + l31 * 256 + (((2 * ks + h) ^ (l31 & 15)) * 16);
  const unsigned q_off = W64_Q_OFF + wave * (64 * 256);  // Q fragment of k-step ks, rows 0-31 of this wave: kaddr[ks] + q_off (+ 8192: rows 32-63)
  unsigned vaddr[4];                                     // V^T reads of d block db: row 4 h + tq, 8 bytes at d = 32 db + 16 gi + 4 tp
  {
    const int l15 = lane & 15, tq = l15 >> 2, tp = l15 & 3, gi = (lane >> 4) & 1;
#pragma unroll
    for (int db = 0; db < 4; ++db)
      vaddr[db] = smem_u32 + (4 * h + tq) * 256 + ((4 * (db ^ tq) + 2 * gi + (tp >> 1)) * 16) + (tp & 1) * 8;
  }

  // ---- state ----------------------------------------------------------------------------------------------------------
  f32x16 o[4][2];                                        // O^T[d block][row block] (AGPRs)
  float ms[2], lsum[2];                                  // reference maximum of a row in log2 units (scores arrive scaled), row sum
  u32x4 seed_b[2];                                       // B fragment of the seeding MFMA: -ms in k-row 0 of this lane's query row
  const u32x4 seed_a = {h == 0 ? w64_mfma<T>::kOne : 0u, 0u, 0u, 0u};   // A fragment: 1 in k-column 0 of every key row
  bool first = true;                                     // no reference yet (the golden's -inf): the first tile sets it
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    ms[rb] = 0.f;
    lsum[rb] = 0.f;
    seed_b[rb] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[db][rb][e] = 0.f;
  }
  constexpr float LAG = PF_LAZY_LOG2;                    // S^T holds s * scale - ms: a lane maximum above LAG moves the reference

  // ---- building blocks (sequential forms) -------------------------------------------------------------------------------
  struct STile { f32x16 s[2][2]; };                      // S^T[key block][row block] of one 64-key tile (VGPRs)
  struct PTile { u32x4 p[2][2][2]; };                    // P^T fragments [row block][key block][16-key step], 8 storage-type values each

  // the four MFMAs of half-group j (key block j / 4, k-steps 2 (j % 4), + 1) on its six fragments
  auto qk_mfma4 = [&](STile& S, u32x4 (&f)[6], int j) {
    const int kbk = j >> 2;
    if ((j & 3) == 0) {                                  // a key block starts: accumulators = -ms
      w64_mfma<T>::s_first(S.s[kbk][0], seed_a, seed_b[0]);
      w64_mfma<T>::s_first(S.s[kbk][1], seed_a, seed_b[1]);
    }
    w64_mfma<T>::s_acc(S.s[kbk][0], f[0], f[2]);
    w64_mfma<T>::s_acc(S.s[kbk][1], f[0], f[3]);
    w64_mfma<T>::s_acc(S.s[kbk][0], f[1], f[4]);
    w64_mfma<T>::s_acc(S.s[kbk][1], f[1], f[5]);
  };
  auto qk_seq = [&](STile& S, int kb) {                  // S = K(kb) Q^T - ms
    const unsigned so = k_slot(kb);
    u32x4 f0[6], f1[6];                                  // two buffers in turn: a request never targets registers the previous MFMAs may still be reading
    static_for<8>([&](auto JC) {
      constexpr int j = decltype(JC)::value;
      constexpr int ks = 2 * (j & 3);
      u32x4 (&f)[6] = (j & 1) ? f1 : f0;
      if constexpr (j < 4) w64_kq_issue<0>(f, kaddr[ks] + so, kaddr[ks + 1] + so, kaddr[ks] + q_off, kaddr[ks + 1] + q_off);
      else w64_kq_issue<8192>(f, kaddr[ks] + so, kaddr[ks + 1] + so, kaddr[ks] + q_off, kaddr[ks + 1] + q_off);
      w64_kq_retire<0>(f);
      qk_mfma4(S, f, j);
    });
    w64_settle_s(S.s[0][0], S.s[0][1], S.s[1][0], S.s[1][1]);   // vector code reads S next
  };
  auto mask_tile = [&](STile& S, int kb) {               // diagonal / tail / hole tiles
#pragma unroll
    for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = kb * PF_KEYS + 32 * kbk + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (key >= first_neg_key) S.s[kbk][rb][e] = -ms[rb];               // zero K rows: score 0
          if (key > offset + row_pos[rb] || key >= kv_len) S.s[kbk][rb][e] = -INFINITY;
        }
  };
  auto lane_max = [&](const STile& S, int rb) -> float {
    float mx = S.s[0][rb][0];
#pragma unroll
    for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
      for (int e = 0; e < 16; e += 2) mx = w64_max3(mx, S.s[kbk][rb][e], S.s[kbk][rb][e + 1]);
    return mx;
  };
  // New reference for the rows whose maximum moved by more than the lag (every row on the first tile): `d` = how far, in
  // log2 units.  The tile at hand (its scores were formed against the old reference) is shifted, the accumulator seed of the
  // following tiles updated; the caller scales O and l by `alpha` once everything at the old scale is inside them.
  auto new_reference = [&](STile& S, const float (&mxl)[2], float (&alpha)[2]) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      float d = w64_xor32_max(mxl[rb]);
      if (first) d = (d == -INFINITY) ? 0.f : d; else d = fmaxf(d, 0.f);
      const float ms_new = w64_mfma<T>::rounded(ms[rb] + d);   // exactly representable in the storage type (it rides in a B fragment)
      d = ms_new - ms[rb];
      alpha[rb] = first ? 0.f : fast_exp2(-d);           // (first tile: O and l are zero; 2^-d may be infinite)
      ms[rb] = ms_new;
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
        for (int e = 0; e < 16; ++e) S.s[kbk][rb][e] -= d;
      seed_b[rb] = u32x4{h == 0 ? w64_mfma<T>::bits_of(-ms[rb]) : 0u, 0u, 0u, 0u};
    }
    first = false;
  };
  auto scale_o = [&](const float (&alpha)[2]) {
    w64_settle_o(o);                                     // O may just have been written by MFMAs
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      lsum[rb] *= alpha[rb];
#pragma unroll
      for (int db = 0; db < 4; ++db) o[db][rb] *= alpha[rb];
    }
  };
  auto rescale = [&](STile& S, const float (&mxl)[2]) {  // sequential form: nothing is pending
    float alpha[2];
    new_reference(S, mxl, alpha);
    scale_o(alpha);
  };
  auto triggered = [&](const float (&mxl)[2]) -> bool { return first || __any(mxl[0] > LAG || mxl[1] > LAG); };
  // One probability in two steps.  start: p = 2^x (x = s * scale - m arrives from the matrix pipe).  finish: the row sum takes the fp32 value, P the
  // value rounded to the storage type; even elements wait in `hold` so that the two conversions of a register pair sit
  // together (one v_cvt_pk); the four words of a fragment are collected in `w` and the fragment is DEFINED as a whole when
  // its last word arrives (an element insert would read the fragment's previous value and keep the whole previous P tile
  // alive across the loop).  The empty asm statements keep an element's instructions where they are written (hipcc sinks
  // them into the block of their first use otherwise).
  typedef typename vec_of<T, 2>::type T2;
  auto sm_start = [&](const STile& S, int rb, int idx) -> float {
    float p = fast_exp2(S.s[idx >> 4][rb][idx & 15]);
    asm volatile("" : "+v"(p));
    return p;
  };
  auto sm_finish = [&](PTile& P, int rb, int idx, float p, float& ps, float& hold, unsigned (&w)[4], bool hole, int kb) {
    const int kbk = idx >> 4, e = idx & 15;
    ps += p;
    asm volatile("" : "+v"(ps));
    if (hole && kb * PF_KEYS + 32 * kbk + (e & 3) + 8 * (e >> 2) + 4 * h >= first_neg_key) p = 0.f;   // zero V rows
    if ((e & 1) == 0) {
      hold = p;
    } else {
      const T2 pr = {static_cast<T>(hold), static_cast<T>(p)};
      w[(e & 7) >> 1] = __builtin_bit_cast(unsigned, pr);
      asm volatile("" : "+v"(w[(e & 7) >> 1]));
      if ((e & 7) == 7) P.p[rb][kbk][e >> 3] = u32x4{w[0], w[1], w[2], w[3]};
    }
  };
  auto sm_seq = [&](const STile& S, PTile& P, bool hole, int kb) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      float ps = 0.f, hold = 0.f;
      unsigned w[4] = {0, 0, 0, 0};
#pragma unroll
      for (int idx = 0; idx < 32; ++idx) sm_finish(P, rb, idx, sm_start(S, rb, idx), ps, hold, w, hole, kb);
      lsum[rb] += ps;
    }
  };
  auto v_frag = [](const s16x4 (&v)[8], int db) -> u32x4 {
    const s16x4 lo = v[2 * db], hi = v[2 * db + 1];
    const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(u32x4, both);
  };
  // the eight MFMAs of 16-key step f: eight different accumulators in a row
  auto pv_mfma8 = [&](const PTile& P, const s16x4 (&v)[8], int f) {
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) w64_mfma<T>::o_acc(o[db][rb], v_frag(v, db), P.p[rb][f >> 1][f & 1]);
  };
  auto pv_seq = [&](const PTile& P, int kb) {            // O += V(kb)^T P^T
    const unsigned so = v_slot(kb);
    s16x4 v0[8], v1[8];                                  // (two buffers in turn, as in qk_seq)
    static_for<4>([&](auto FC) {
      constexpr int f = decltype(FC)::value;
      s16x4 (&v)[8] = (f & 1) ? v1 : v0;
      w64_v_issue<f>(v, vaddr[0] + so, vaddr[1] + so, vaddr[2] + so, vaddr[3] + so);
      w64_v_retire<0>(v);
      pv_mfma8(P, v, f);
    });
  };
  auto end_of_iteration = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my pieces of the tiles in flight have landed ...
    __builtin_amdgcn_s_barrier();                         // ... everyone's have, and everyone is done reading this iteration's tiles
  };

  // ---- prologue: K(0), K(1), V(0); S(0) -----------------------------------------------------------------------------------
  stage_tile(false, 0);
  stage_tile(false, 1);
  stage_tile(true, 0);
  end_of_iteration();                                    // (also: the Q rows of every wave are in LDS)
  STile SA, SB;
  PTile PA, PB;
  float mxl[2];
  qk_seq(SA, 0);
  if (n_full < 1) mask_tile(SA, 0);
  mxl[0] = lane_max(SA, 0);
  mxl[1] = lane_max(SA, 1);
  int t = 0;

  // ---- pipelined region: tiles that every row of the workgroup sees completely ---------------------------------------------
  // iteration tt: (Scur = S(tt), Pprev = P(tt-1), PV(tt-1) pending) -> (Snext = S(tt+1), Pcur = P(tt)); stages K(tt+2), V(tt+1).
  // What an interleaved iteration expects from its predecessor (`prime`): the V^T fragments of the first 16-key step of
  // tile tt - 1 in va and the page ids of this wave's keys of tiles tt + 2 (K) and tt + 1 (V), requested before the barrier.
  s16x4 va[8];
  int phys_k = 0, phys_v = 0;
  auto prime = [&](int tn) {                             // for iteration tn
    const unsigned so = v_slot(tn - 1);
    w64_v_issue<0>(va, vaddr[0] + so, vaddr[1] + so, vaddr[2] + so, vaddr[3] + so);
    phys_k = page_of_tile_lds(tn + 2);
    phys_v = page_of_tile_lds(tn + 1);
    w64_v_retire<0>(va);
  };
  auto iter_seq0 = [&](STile& Scur, STile& Snext, PTile& Pcur) {     // iteration 0: nothing pending
    stage_tile(false, 2);
    stage_tile(true, 1);
    if (triggered(mxl)) rescale(Scur, mxl);
    sm_seq(Scur, Pcur, false, 0);
    qk_seq(Snext, 1);
    mxl[0] = lane_max(Snext, 0);
    mxl[1] = lane_max(Snext, 1);
    prime(1);
    end_of_iteration();
  };
#ifdef PF_STAMPS
  unsigned tacc = 0, n_fast_iters = 0;                   // (-DPF_STAMPS: cycles between the stamps of iter_fast; results are NOT valid in
  unsigned long long t_prev = __builtin_readcyclecounter();   //  such a build — s_memtime shares lgkmcnt with the counted LDS waits)
#endif
#ifndef W64_ABLATE
#define W64_ABLATE 0                                     // timing experiments only (results are wrong): 1 no softmax, 2 no DMA, 4 no maxima
#endif
  // the interleaved form: MFMA gap by gap (see the header)
  auto iter_fast = [&](STile& Scur, STile& Snext, PTile& Pprev, PTile& Pcur, int tt) {
    PF_STAMP(0);
    const unsigned so_v = v_slot(tt - 1), so_k = k_slot(tt + 1), so_vn = v_slot(tt);
    const unsigned slot_k = k_slot(tt + 2), slot_v = v_slot(tt + 1);
    const bool fast_k = tt + 2 < n_full;                 // K(tt + 2) is a complete tile: its pieces go out one by one between the MFMAs
    if (!fast_k) stage_tile(false, tt + 2);              // (V(tt + 1) is complete in this region: tt + 1 < n_full)
    const int64_t sb_k = fast_k ? stage_base_of(tt + 2, phys_k) : 0;
    const int64_t sb_v = stage_base_of(tt + 1, phys_v);
    // lazy reference: new reference now, O / l follow at the end of the iteration (see the header)
    const bool trig = triggered(mxl);
    float alpha[2] = {1.f, 1.f};
    if (trig) new_reference(Scur, mxl, alpha);           // (also re-seeds the QK^T of this iteration)
    float ps0 = 0.f, ps1 = 0.f, hold = 0.f, p_pend = 0.f;
    unsigned w[4] = {0, 0, 0, 0};
    // Fragment buffers are requested again right behind the last MFMA that read them: the matrix pipe has taken its A / B
    // operands by then (scripts/probes/mfma_war_probe.hip: no corruption at any distance).  What DID corrupt results on the
    // way here were registers the COMPILER wrote with v_accvgpr_write / _mov right in front of an asm MFMA (it inserts no
    // wait states for an instruction it cannot see): keep the AGPR demand low enough that O never leaves its registers.
    u32x4 fa[6], fb[6];                                  // QK^T half-group j uses buffer j % 2, requested one half-group (4 MFMAs) ahead
    s16x4 vb[8];                                         // PV 16-key step f uses va / vb in turn, requested 8 MFMAs ahead
    unsigned page_k = 0, page_v = 0;
    float mxa = -INFINITY, mxb = -INFINITY;
    PF_STAMP(1);
    // 68 gaps: PV(tt - 1) [0, 32); QK^T(tt + 1) [32, 68) = per 32-key block two seeding MFMAs (accumulators = -ms) and four
    // half-groups of four MFMAs.  Half-group j starts at MJ(j).
    static_for<68>([&](auto MC) {
      constexpr int M = decltype(MC)::value;
      constexpr int q = M - 32, qr = q >= 0 ? q % 18 : 0, kbk = q >= 18 ? 1 : 0;
      constexpr bool seed = q >= 0 && qr < 2;
      constexpr int j = q >= 0 && !seed ? kbk * 4 + (qr - 2) / 4 : -1;      // half-group of this MFMA
      constexpr int within = q >= 0 && !seed ? (qr - 2) % 4 : 0;
      // ---- LDS requests and retirements ------------------------------------------------------------------------------------
      if constexpr (M == 0)  w64_v_issue<1>(vb, vaddr[0] + so_v, vaddr[1] + so_v, vaddr[2] + so_v, vaddr[3] + so_v);   // (step 0 came retired)
      if constexpr (M == 8)  { w64_v_issue<2>(va, vaddr[0] + so_v, vaddr[1] + so_v, vaddr[2] + so_v, vaddr[3] + so_v); w64_v_retire<8>(vb); }
      if constexpr (M == 16) { w64_v_issue<3>(vb, vaddr[0] + so_v, vaddr[1] + so_v, vaddr[2] + so_v, vaddr[3] + so_v); w64_v_retire<8>(va); }
      if constexpr (M == 24) w64_v_retire<0>(vb);
      if constexpr (M == 28) w64_kq_issue<0>(fa, kaddr[0] + so_k, kaddr[1] + so_k, kaddr[0] + q_off, kaddr[1] + q_off);
      if constexpr (j >= 0 && within == 0) {             // half-group j starts: request j + 1, retire j
        if constexpr (j + 1 < 8) {
          constexpr int jn = j + 1, ks = 2 * (jn & 3);
          u32x4 (&fn)[6] = (jn & 1) ? fb : fa;
          if constexpr (jn < 4) w64_kq_issue<0>(fn, kaddr[ks] + so_k, kaddr[ks + 1] + so_k, kaddr[ks] + q_off, kaddr[ks + 1] + q_off);
          else w64_kq_issue<8192>(fn, kaddr[ks] + so_k, kaddr[ks + 1] + so_k, kaddr[ks] + q_off, kaddr[ks + 1] + q_off);
        }
        u32x4 (&fj)[6] = (j & 1) ? fb : fa;
        if constexpr (j < 7) w64_kq_retire<6>(fj); else w64_kq_retire<0>(fj);
      }
      if constexpr (j == 7 && within == 2) {
        // the next iteration's first V^T group (tile tt: landed an iteration ago; va was last read by MFMA 23) and its page
        // ids (tiles tt + 3 / tt + 2)
        w64_v_issue<0>(va, vaddr[0] + so_vn, vaddr[1] + so_vn, vaddr[2] + so_vn, vaddr[3] + so_vn);
        int lpk = ((tt + 3) * PF_KEYS + wave * 16) >> a.page_shift, lpv = ((tt + 2) * PF_KEYS + wave * 16) >> a.page_shift;
        lpk = min(lpk, W64_TABLE - 1);
        lpv = min(lpv, W64_TABLE - 1);
        const unsigned tb = smem_u32 + W64_TABLE_OFF;
        asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3" : "=&v"(page_k), "=&v"(page_v) : "v"(tb + lpk * 4), "v"(tb + lpv * 4) : "memory");
      }
      // ---- the MFMA -----------------------------------------------------------------------------------------------------
      if constexpr (M < 32) {                            // PV(tt - 1): 16-key step f = M / 8; accumulators O[0][0], O[0][1], O[1][0], ... O[3][1] in turn
        constexpr int f = M / 8, db = (M % 8) / 2, rb = M % 2;
        const s16x4 (&vv)[8] = (f & 1) ? vb : va;
        w64_mfma<T>::o_acc(o[db][rb], v_frag(vv, db), Pprev.p[rb][f >> 1][f & 1]);
      } else if constexpr (seed) {                       // QK^T(tt + 1), key block kbk: accumulators = -ms
        w64_mfma<T>::s_first(Snext.s[kbk][qr], seed_a, seed_b[qr]);
      } else {                                           // half-group j, MFMA `within`: k-step 2 (j % 4) + within / 2, row block within % 2
        constexpr int w2 = within / 2, rb = within % 2;
        u32x4 (&ff)[6] = (j & 1) ? fb : fa;
        w64_mfma<T>::s_acc(Snext.s[kbk][rb], ff[w2], ff[2 + 2 * w2 + rb]);
      }
      // ---- fillers: start softmax element M, finish element M - 1 ---------------------------------------------------------------
      float p_new = 0.f;
      if constexpr (!(W64_ABLATE & 1) && M < 64) p_new = (M < 32) ? sm_start(Scur, 0, M) : sm_start(Scur, 1, M - 32);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(W64_ABLATE & 1) && M > 0 && M <= 64) {
        if constexpr (M - 1 < 32) sm_finish(Pcur, 0, M - 1, p_pend, ps0, hold, w, false, tt);
        else sm_finish(Pcur, 1, M - 33, p_pend, ps1, hold, w, false, tt);
      }
      p_pend = p_new;
      if constexpr (M < 32 && M % 4 == 1 && !(W64_ABLATE & 2)) {   // DMA pieces (all in the first half: landed long before the barrier)
        constexpr int i = M / 4;
        if constexpr (i < 4) { if (fast_k) stage_piece(false, slot_k, sb_k, i); }
        else stage_piece(true, slot_v, sb_v, i - 4);
      }
      if constexpr (M >= 52 && !(W64_ABLATE & 4)) {      // lane maxima of S(tt + 1), key block 0 (complete since MFMA 49): one v_max3 per gap
        constexpr int jj = M - 52, rb = jj / 8, e0 = 2 * (jj % 8);
        float& mx = rb ? mxb : mxa;
        mx = w64_max3(mx, Snext.s[0][rb][e0], Snext.s[0][rb][e0 + 1]);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    PF_STAMP(2);
    if (trig) scale_o(alpha);                            // the PV(tt - 1) MFMAs ran in the first half: O and l take the new scale
    lsum[0] += ps0;
    lsum[1] += ps1;
    w64_settle_s(Snext.s[0][0], Snext.s[0][1], Snext.s[1][0], Snext.s[1][1]);   // S(tt + 1), key block 1, was written by the last MFMAs
#pragma unroll
    for (int e = 0; e < 16; e += 2) {
      mxa = w64_max3(mxa, Snext.s[1][0][e], Snext.s[1][0][e + 1]);
      mxb = w64_max3(mxb, Snext.s[1][1][e], Snext.s[1][1][e + 1]);
    }
    mxl[0] = mxa;
    mxl[1] = mxb;
    PF_STAMP(3);
    w64_v_retire<0>(va);                                 // before the back edge (the compiler may copy these registers there)
    asm volatile("" : "+v"(page_k), "+v"(page_v));
    phys_k = __builtin_amdgcn_readfirstlane(static_cast<int>(page_k));
    phys_v = __builtin_amdgcn_readfirstlane(static_cast<int>(page_v));
    PF_STAMP(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PF_STAMP(5);
    __builtin_amdgcn_s_barrier();
    PF_STAMP(6);
#ifdef PF_STAMPS
    ++n_fast_iters;
#endif
  };
#ifdef W64_DEBUG_SEQ
  // debugging aid: the pipelined region with SEQUENTIAL iterations (same rings, same priming, same reference logic)
  auto iter_dbg = [&](STile& Scur, STile& Snext, PTile& Pprev, PTile& Pcur, int tt) {
    stage_tile(false, tt + 2);
    stage_tile(true, tt + 1);
    pv_seq(Pprev, tt - 1);
    if (triggered(mxl)) rescale(Scur, mxl);
    sm_seq(Scur, Pcur, false, tt);
    qk_seq(Snext, tt + 1);
    mxl[0] = lane_max(Snext, 0);
    mxl[1] = lane_max(Snext, 1);
    prime(tt + 1);
    end_of_iteration();
  };
#define iter_fast iter_dbg
#endif
  if (n_full >= 2) {
    iter_seq0(SA, SB, PA);                               // now S(1) in SB, P(0) in PA
    t = 1;
    while (t + 2 < n_full) {                             // two iterations per trip: the register sets swap roles by name
      iter_fast(SB, SA, PA, PB, t);
      iter_fast(SA, SB, PB, PA, t + 1);
      t += 2;
    }
    if (t + 1 < n_full) {
      iter_fast(SB, SA, PA, PB, t);
      ++t;
      SB = SA;
      PA = PB;
    }
#ifdef PF_STAMPS
    if (blockIdx.x < 8192) {
      if (lane < 15) g_pf_stamps[(blockIdx.x * 4 + wave) * 16 + lane] = tacc;
      if (lane == 15) g_pf_stamps[(blockIdx.x * 4 + wave) * 16 + 15] = n_fast_iters;
    }
#endif
    pv_seq(PA, t - 1);                                   // drain: S(t) in SB is the last complete tile, nothing pending
  } else {
    SB = SA;
  }

  // ---- remaining tiles (the last complete one, diagonal / tail / hole tiles): plain order, masked -----------------------
  for (; t < n_kb; ++t) {
    stage_tile(false, t + 2);
    stage_tile(true, t + 1);
    if (triggered(mxl)) rescale(SB, mxl);
    sm_seq(SB, PA, t >= n_full && (t + 1) * PF_KEYS > first_neg_key, t);
    pv_seq(PA, t);
    if (t + 1 < n_kb) {
      qk_seq(SB, t + 1);
      if (t + 1 >= n_full) mask_tile(SB, t + 1);
      mxl[0] = lane_max(SB, 0);
      mxl[1] = lane_max(SB, 1);
    }
    end_of_iteration();
  }

  // ---- finish: row sums over the two lane halves, normalise, transpose through LDS, store whole rows --------------------
  // (the last end_of_iteration() barrier means every wave is done with the tile rings, which the staging area overlays)
  w64_settle_o(o);
  lds_c* stage_o = smem + wave * (64 * W64_OROW);
  typedef typename vec_of<T, 4>::type V4;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const float inv = 1.0f / w64_xor32_sum(lsum[rb]);
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        V4 ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) ov[r] = static_cast<T>(o[db][rb][4 * j + r] * inv);
        *reinterpret_cast<__attribute__((address_space(3))) V4*>(stage_o + (rb * 32 + l31) * W64_OROW + (32 * db + 8 * j + 4 * h) * 2) = ov;
      }
  }
  {
    typedef typename vec_of<T, 8>::type V8;
    const int sub = lane >> 4, ch = lane & 15;           // 4 rows per store instruction, 16 bytes per lane
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = i * 4 + sub;
      const int r = wave * 64 + row;
      const int pos = qb * QPB + (r % QPB);
      if (pos >= q_len) continue;
      const int g = r / QPB;
      const int head = a.abab ? g * a.hkv + kvh : kvh * G + g;
      const V8 v = *reinterpret_cast<const __attribute__((address_space(3))) V8*>(stage_o + row * W64_OROW + ch * 16);
      *reinterpret_cast<V8*>(static_cast<T*>(a.out) + (static_cast<int64_t>(q_start + pos) * a.hq + head) * 128 + ch * 8) = v;
    }
  }
}

}  // namespace mojo
