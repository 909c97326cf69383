// MojoPagedPrefillGQA — flash attention (online softmax) over a paged KV cache on MFMA, gfx950.
//
// Work decomposition
//   1-D grid of q blocks x Hkv x B workgroups, query block the slow coordinate and descending (longest first over the whole
//   launch), plus trailing workgroups that zero the padding rows; one 256-thread workgroup owns 128 "rows" = the G = Hq/Hkv
//   query heads of one kv-head times 128/G consecutive query positions, so every K/V tile is read once per kv-head.
//   Each of the 4 waves owns 32 rows = two 16-row MFMA tiles.
//   K/V advance in tiles of 64 keys, double-buffered in LDS (2 x (16 + 16) KiB), filled by direct-to-LDS
//   loads straight from the pages (4 keys x 256 B per wave instruction, page ids looked up per 4 keys).
//
// MFMA formulation (v_mfma_f32_16x16x32): everything is computed transposed so that softmax statistics
// are lane-local per query row and the probabilities feed the second product without leaving registers:
//   S^T[key][q]  = sum_d K[key][d] * Q[q][d]          A = K fragment (ds_read_b128), B = Q fragment (VGPRs)
//   O^T[d][q]   += sum_key V[key][d] * P^T[key][q]    A = V^T fragment (ds_read_b64_tr_b16), B = P^T built from
//                                                     the S^T accumulators (key order inside a k-step is the
//                                                     accumulator's own order; V^T is read in the same order)
// LDS images: rows of 256 B (head_dim <= 128); K: 16-byte chunk c of key r stored at c ^ (r & 15);
// V: chunk c stored at c ^ ((r & 7) << 1); both conflict-free for their read pattern, applied on the
// SOURCE address of the lane-linear LDS-DMA.
//
// Numerics: fp32 scores and statistics (lazy reference maximum: it may lag the true one by 2^8, the final division uses sums
// taken against the same reference), probabilities rounded to the storage type for the PV product (as the golden does),
// fp32 output accumulation; the output leaves through a wave-private LDS transpose as whole rows.  Parity by tolerance:
// atol = rtol = 2e-2.
//
// Algorithmic FLOPs (causal): sum_b 4 * Hq * D * (q_b * kv_b - q_b^2 / 2).   Bound: MFMA.
#include <math.h>

#include <type_traits>

#include "common.h"

namespace mojo {

typedef __attribute__((address_space(3))) char lds_c;

struct PrefillArgs {
  const void* q;
  const void* kc;
  const void* vc;
  void* out;
  const int32_t* cu_q;
  const int32_t* cu_kv;      // may be null: kv_len = q_len
  const int32_t* tables;
  int64_t table_stride, c_blk, c_head, c_tok;
  int hq, hkv, dim, page, page_shift, max_pages;
  int batch, n_qb;           // grid = n_qb * hkv * batch workgroups (see the kernel for the order) + the zero-fill tail
  int64_t total_tokens;
  float scale_log2;
  int abab;
  int fast_stage;            // pages are a power of two >= 16 keys and the per-lane offsets fit 32 bits
  int skew;                  // rotate the sequence coordinate by this much per query-block level (0 or 1)
  // key split (few, long blocks: a chunked prefill against a long cache): every (query block, kv head, sequence) workgroup
  // becomes `ksplit` workgroups that each walk a slice of the block's key tiles and leave un-normalised fp32 partials;
  // prefill_merge_kernel combines them.  ksplit == 1: no workspace, the workgroup writes `out` itself.
  int ksplit;
  int pp;                    // 1 (experiments build only): prefill_pp_kernel (experiments/paged_prefill_pp.h): 8-wave workgroups of 256 rows, n_qb counts blocks of 256 / G positions
  int m32;                   // 1 (experiments build only): prefill_m32_kernel (experiments/paged_prefill_m32.h): the same decomposition on 32x32x16 MFMAs (head_dim 128)
  int w64;                   // 1 (experiments build only): prefill_w64_kernel (experiments/paged_prefill_w64.h): one wave per SIMD, 256-row workgroups, n_qb counts blocks of 256 / G positions
  float* ws_o;               // [blocks * ksplit][128 rows][dim]
  float* ws_ml;              // [blocks * ksplit][128 rows][2]   reference maximum (log2 units), row sum
};

template <typename T> struct pf_mfma;
template <> struct pf_mfma<bf16_t> {
  typedef bf16x8 frag;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct pf_mfma<f16_t> {
  typedef f16x8 frag;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

#if defined(PF_STAMPS) || defined(PF_WG_STAMPS)
// In-kernel stamps (build with MOJO_HIP_EXTRA_CXXFLAGS=-DPF_STAMPS): lane i of `tacc` accumulates the cycles between
// stamp i-1 and stamp i of the hot loop; read back with mojo_hip_debug_prefill_stamps.  Timing tool only.
__device__ unsigned g_pf_stamps[8192 * 4 * 16];
#ifdef PF_STAMPS
#define PF_STAMP(i)                                                              \
  do {                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                           \
    const unsigned long long t_ = __builtin_readcyclecounter();                  \
    const unsigned d_ = static_cast<unsigned>(t_ - t_prev);                      \
    t_prev = t_;                                                                 \
    tacc += (lane == (i)) ? d_ : 0u;                                             \
    __builtin_amdgcn_sched_barrier(0);                                           \
  } while (0)
#else
#define PF_STAMP(i)
#endif
#else
#define PF_STAMP(i)
#endif

#ifdef PF_WG_STAMPS
// Workgroup-level phases (-DPF_WG_STAMPS): prologue, hot loop, remaining loops, epilogue — one record per workgroup.
#define PF_WG_MARK(slot)                                                                          \
  do {                                                                                            \
    const unsigned long long t_ = __builtin_readcyclecounter();                                   \
    if (threadIdx.x == 0 && blockIdx.x < 8192 * 4) g_pf_stamps[blockIdx.x * 16 + (slot)] = static_cast<unsigned>(t_ - wg_t0); \
  } while (0)
#else
#define PF_WG_MARK(slot)
#endif

constexpr float PF_LAZY_LOG2 = 8.f;          // the reference maximum may lag the true one by this many powers of two

constexpr int PF_KEYS = 64;                  // keys per tile
constexpr int PF_ZERO_TOKENS = 32;           // padding tokens one trailing workgroup zeroes
constexpr int PF_TILE_BYTES = PF_KEYS * 256; // 16 KiB per K or V tile
constexpr int PF_TABLE = 1024;                // block-table entries cached in LDS
constexpr int PF_LDS = 4 * PF_TILE_BYTES + PF_TABLE * 4 + 16;    // K0 V0 K1 V1 | table slice | first negative page

template <typename T, int G /* q heads per kv head */, int DK /* head_dim / 32 */, bool SPLIT = false /* key split, see PrefillArgs */>
__global__ __launch_bounds__(256, 2) void prefill_kernel(PrefillArgs a) {
  typedef typename pf_mfma<T>::frag frag;
  constexpr int QPB = 128 / G;               // query positions per workgroup
  constexpr int DT = DK * 2;                 // 16-wide d tiles
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
  const int kvh = rem % a.hkv, b = __builtin_amdgcn_readfirstlane((rem / a.hkv + a.skew * (wg / inner)) % a.batch);
  // The prologue is a chain of dependent memory round trips (1.5-2 us each on a busy chip) in front of a workgroup that may
  // own only a handful of tiles.  Round 4: it is TWO of them — {sequence bounds, page-id window} then {Q fragments, first
  // tile} — where it was three or four (bounds, cached-length bounds, {Q, page ids}, first tile): the bounds are scalar loads
  // issued together, and the page-id window depends on nothing but the sequence index, so its loads are requested in front of
  // the wait for the bounds.
  const int32_t* table = a.tables + static_cast<int64_t>(b) * a.table_stride;
  int ids[PF_TABLE / 256];
#pragma unroll
  for (int j = 0; j < PF_TABLE / 256; ++j) {
    const int i = threadIdx.x + j * 256;
    ids[j] = i < a.max_pages ? __builtin_nontemporal_load(table + i) : -1;
  }
  // (asm scalar loads: left to hipcc the four bounds are vector loads, the cached-length pair behind a branch and a wait each)
  const int32_t* pq = a.cu_q + b;
  const int32_t* pk = (a.cu_kv ? a.cu_kv : a.cu_q) + b;
  int q_start, q_end, kv_lo, kv_end;
  asm volatile("s_load_dword %0, %1, 0x0" : "=s"(q_start) : "s"(pq) : "memory");
  asm volatile("s_load_dword %0, %1, 0x4" : "=s"(q_end) : "s"(pq) : "memory");
  asm volatile("s_load_dword %0, %1, 0x0" : "=s"(kv_lo) : "s"(pk) : "memory");
  asm volatile("s_load_dword %0, %1, 0x4" : "=s"(kv_end) : "s"(pk) : "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(q_start), "+s"(q_end), "+s"(kv_lo), "+s"(kv_end) : : "memory");
#pragma unroll
  for (int j = 0; j < PF_TABLE / 256; ++j) asm volatile("" : "+v"(ids[j]));      // (the window's loads are requested up there, not where hipcc finds their first use)
  const int q_len = q_end - q_start;
  const int kv_len = a.cu_kv ? kv_end - kv_lo : q_len;
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
  const int grp = lane >> 4, l15 = lane & 15;

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

  // ---- this wave's rows: two 16-row tiles; row -> (head g, query position) -------------------------------
  int row_pos[2], row_head[2];
  const T* qptr[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int r = wave * 32 + qt * 16 + l15;
    const int g = r / QPB;
    int pos = qb * QPB + (r % QPB);
    row_head[qt] = a.abab ? g * a.hkv + kvh : kvh * G + g;
    row_pos[qt] = pos;
    if (pos >= q_len) pos = q_len - 1;                   // clamp: computed, never stored
    qptr[qt] = static_cast<const T*>(a.q) + (static_cast<int64_t>(q_start + pos) * a.hq + row_head[qt]) * a.dim;
  }
  frag qf[2][DK];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int ks = 0; ks < DK; ++ks) qf[qt][ks] = *reinterpret_cast<const frag*>(qptr[qt] + ks * 32 + grp * 8);   // (scaled below, behind the first tile's requests)

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
    // (`ids` were loaded at the top of the kernel.  The two barriers are raw s_barrier + LDS waits: __syncthreads() carries a
    // vmcnt(0) that would wait for the Q fragments here, i.e. put the first tile's requests a round trip behind them.)
    int* s_fn = s_table + PF_TABLE;
    if (threadIdx.x == 0) *s_fn = 0x7fffffff;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < PF_TABLE / 256; ++j) {
      const int i = threadIdx.x + j * 256;
      s_table[i] = ids[j];
      if (ids[j] < 0 && i < p1) atomicMin(s_fn, i);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
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
      int cv = cp ^ ((kl & 7) << 1);
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
    int cv = cp ^ ((kl & 7) << 1);
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
  // (sk / sv: the tile's K / V row bases as SCALAR pointers pinned by an empty asm — left to itself hipcc hoists
  // `kbase + voff` out of the loop as a 64-bit vector and adds the tile's base with two vector instructions per piece; a
  // scalar base + a 32-bit lane offset is the instruction's own addressing mode)
  auto stage_fast_piece = [&](int buf, const char* sk, const char* sv, int i) {
    unsigned vo = (i & 1) ? voff_v[i >> 1] : voff_k[i >> 1];
    asm volatile("" : "+v"(vo));                         // keeps the zero-extension next to the load (instruction selection is per block)
    const char* src = ((i & 1) ? sv : sk) + vo;
    lds_c* dst = smem + buf * 2 * PF_TILE_BYTES + (i & 1) * PF_TILE_BYTES + (wave * 16 + (i >> 1) * 4) * 256;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };

  // ---- state ----------------------------------------------------------------------------------------------
  f32x4 o[2][DT];
  // m: reference of the running softmax (log2 units; -inf until a key was seen).  seed = -m (0 until then) starts every S
  // accumulator; thr = what a lane maximum of relative scores must exceed to move the reference (-inf until then: any finite
  // score does).  All three change only on the rare reference-update path.
  // The row sums come out of the matrix pipe: osum[qt] accumulates P against a V^T tile of ones (one MFMA per 32 keys and q
  // tile, +6 % MFMAs) — every lane then holds the sum over ALL keys of its query column, and the 32 vector adds per tile and
  // the closing cross-lane reduction are gone (the kernel is issue-bound on vector instructions, section 4.4).
  float m[2], seed[2], thr[2];
  f32x4 seedv[2];                // seed in the four registers an MFMA reads its C operand from
  f32x4 osum[2];
  frag ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = static_cast<T>(1.0f);
  float lhole[2] = {0.f, 0.f};   // keys behind a negative page id: score 0 counts in the denominator, their V rows are zero -> P is
                                 // zeroed in front of the matrix pipe and their share of the row sum is kept here (masked tiles only)
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    m[qt] = -INFINITY;
    seed[qt] = 0.f;
    seedv[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
    thr[qt] = -INFINITY;
    osum[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const unsigned smem_u32 = static_cast<unsigned>(reinterpret_cast<size_t>(smem));
  unsigned klane[DK];            // K fragment read offsets inside a tile: key row l15 of a 16-key block, swizzled chunk of k-step ks
#pragma unroll
  for (int ks = 0; ks < DK; ++ks) klane[ks] = l15 * 256 + (((ks * 4 + grp) ^ l15) * 16);
  // V^T transposed-read lane offset: lane 4q+p of a 16-group -> key row (4*grp + q), 8 bytes at column 4p
  const int tq = l15 >> 2, tp = l15 & 3;

  // leading key blocks that every row of this workgroup sees completely need no masking at all; the diagonal /
  // tail / hole blocks run the masked variant.  Two loops, so neither carries the other's state.
  const int n_full = min(min(min(kv_len, offset + qb * QPB + 1), first_neg_key) / PF_KEYS, n_kb);
  const int n_fast = a.fast_stage ? n_full : 0;          // tiles [0, n_fast) may be staged the fast way

  if (kb_lo < n_kb) stage(kb_lo, kb_lo & 1);
  int phys_next = a.fast_stage ? page_of_tile(kb_lo + 1) : 0;    // page id for the NEXT stage, loaded a tile ahead
  // q * scale * log2(e), rounded once to the storage type: the scores leave the MFMA chain in log2 units
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int ks = 0; ks < DK; ++ks) {
      frag f = qf[qt][ks];
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = static_cast<T>(static_cast<float>(f[e]) * a.scale_log2);
      qf[qt][ks] = f;
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  PF_WG_MARK(0);

#ifdef PF_STAMPS
  unsigned tacc = 0;
  unsigned long long t_prev = __builtin_readcyclecounter();
#endif
  auto key_block = [&](auto masked_tag, auto fast_tag, int kb) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    constexpr bool FAST = decltype(fast_tag)::value;     // the NEXT tile is complete and staged the fast way
    if constexpr (FAST) { PF_STAMP(0); }
    const int buf = kb & 1;
    int64_t stage_base = 0;
    const char* stage_k = nullptr;
    const char* stage_v = nullptr;
    if constexpr (FAST) {
      page_ready(phys_next);
      stage_base = stage_fast_base(kb + 1, phys_next);
      asm volatile("" : "+s"(stage_base));               // computed here, not sunk behind the reads
      stage_k = reinterpret_cast<const char*>(kbase) + stage_base;
      stage_v = reinterpret_cast<const char*>(vbase) + stage_base;
      asm volatile("" : "+s"(stage_k), "+s"(stage_v));
    }
    const lds_c* kt = smem + buf * 2 * PF_TILE_BYTES;
    const unsigned vt = smem_u32 + buf * 2 * PF_TILE_BYTES + PF_TILE_BYTES;

    // ---- S^T = K Q^T : 4 key tiles x 2 q tiles -------------------------------------------------------------
    f32x4 s[2][4];
    // all K fragments of the tile are requested before the first MFMA: each feeds only two MFMAs (32 cycles), so a
    // read issued two steps ahead (what hipcc schedules for a read-then-use loop) leaves the LDS latency exposed on
    // every step
    frag kf[4][DK];
    if constexpr (FAST) {
      // asm reads, retired key block by key block with COUNTED waits (LDS reads return in order): hipcc put one lgkmcnt(0) in
      // front of the first MFMA, i.e. waited for all sixteen reads before using the first four
      unsigned ka[DK];
#pragma unroll
      for (int ks = 0; ks < DK; ++ks) ka[ks] = smem_u32 + buf * 2 * PF_TILE_BYTES + klane[ks];
#define PF_KREAD(T_)                                                                                                        \
      if constexpr (DK == 4)                                                                                                \
        asm volatile("ds_read_b128 %0, %4 offset:" #T_ "*4096\n\tds_read_b128 %1, %5 offset:" #T_ "*4096\n\t"               \
                     "ds_read_b128 %2, %6 offset:" #T_ "*4096\n\tds_read_b128 %3, %7 offset:" #T_ "*4096"                   \
                     : "=&v"(kf[T_][0]), "=&v"(kf[T_][1]), "=&v"(kf[T_][DK > 2 ? 2 : 0]), "=&v"(kf[T_][DK > 3 ? 3 : 0])     \
                     : "v"(ka[0]), "v"(ka[1]), "v"(ka[DK > 2 ? 2 : 0]), "v"(ka[DK > 3 ? 3 : 0]) : "memory");               \
      else if constexpr (DK == 3)                                                                                           \
        asm volatile("ds_read_b128 %0, %3 offset:" #T_ "*4096\n\tds_read_b128 %1, %4 offset:" #T_ "*4096\n\t"               \
                     "ds_read_b128 %2, %5 offset:" #T_ "*4096"                                                              \
                     : "=&v"(kf[T_][0]), "=&v"(kf[T_][1]), "=&v"(kf[T_][DK > 2 ? 2 : 0])                                    \
                     : "v"(ka[0]), "v"(ka[1]), "v"(ka[DK > 2 ? 2 : 0]) : "memory");                                        \
      else                                                                                                                  \
        asm volatile("ds_read_b128 %0, %2 offset:" #T_ "*4096\n\tds_read_b128 %1, %3 offset:" #T_ "*4096"                   \
                     : "=&v"(kf[T_][0]), "=&v"(kf[T_][1]) : "v"(ka[0]), "v"(ka[1]) : "memory")
      PF_KREAD(0); PF_KREAD(1); PF_KREAD(2); PF_KREAD(3);
#undef PF_KREAD
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int key_row = t * 16 + l15;
#pragma unroll
        for (int ks = 0; ks < DK; ++ks) {
          const int chunk = (ks * 4 + grp) ^ (key_row & 15);
          kf[t][ks] = *reinterpret_cast<const __attribute__((address_space(3))) frag*>(kt + key_row * 256 + chunk * 16);
        }
      }
    }
    // General staging goes out behind the reads (its issue time covers their latency).  Fast staging is issued one DMA
    // instruction at a time between the softmax's vector instructions, where a piece costs a third of what it costs
    // next to LDS reads.
    if constexpr (!FAST) {
      if (kb + 1 < n_kb) stage(kb + 1, buf ^ 1);
    }
    auto dma_piece = [&](int i) {
      if constexpr (FAST) {
        stage_fast_piece(buf ^ 1, stage_k, stage_v, i);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (FAST) { PF_STAMP(1); }
    // key block T_ of the asm reads has arrived when (3 - T_) * DK reads are still in flight; then its 2 * DK MFMAs.
    // S starts at -reference: the MFMA chain does the subtraction, p = 2^s needs no multiply-add of its own
#define PF_QK(T_)                                                                                                           \
    if constexpr (FAST) {                                                                                                   \
      if (T_ > 0) __builtin_amdgcn_sched_barrier(0);          /* the MFMAs of block T_ - 1 stay in front of this wait */     \
      if constexpr (DK == 4)                                                                                                \
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(kf[T_][0]), "+v"(kf[T_][1]), "+v"(kf[T_][DK > 2 ? 2 : 0]), "+v"(kf[T_][DK > 3 ? 3 : 0]) \
                     : "n"((3 - T_) * DK) : "memory");                                                                      \
      else if constexpr (DK == 3)                                                                                           \
        asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(kf[T_][0]), "+v"(kf[T_][1]), "+v"(kf[T_][DK > 2 ? 2 : 0]) : "n"((3 - T_) * DK) : "memory"); \
      else                                                                                                                  \
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(kf[T_][0]), "+v"(kf[T_][1]) : "n"((3 - T_) * DK) : "memory");           \
    }                                                                                                                       \
    s[0][T_] = seedv[0];                                                                                                    \
    s[1][T_] = seedv[1];                                                                                                    \
    _Pragma("unroll") for (int ks = 0; ks < DK; ++ks) {                                                                     \
      s[0][T_] = pf_mfma<T>::run(kf[T_][ks], qf[0][ks], s[0][T_]);                                                          \
      s[1][T_] = pf_mfma<T>::run(kf[T_][ks], qf[1][ks], s[1][T_]);                                                          \
    }
    PF_QK(0) PF_QK(1) PF_QK(2) PF_QK(3)
#undef PF_QK
    // ---- V^T fragments: transposed reads, 4 per d tile, issued in two batches of DT/2 d tiles.
    // Each batch is one asm statement (issue) + one wait statement naming every destination (hipcc must not touch
    // them in between); the waits are lgkmcnt(0) because scalar loads share the counter and return out of order.
    constexpr int HB = DT / 2 * 4;                       // reads per batch (<= 16)
    auto issue_v = [&](s16x4 (&dst)[16], int dt0) {
      // rows 16 apart share row & 7, so the 4 reads of a d tile differ by compile-time offsets; the swizzled chunk
      // depends on dt: chunk = (2dt + (tp>>1)) ^ ((row & 7) << 1)
      unsigned ad[4];
#pragma unroll
      for (int i = 0; i < DT / 2; ++i) {
        const int dt = dt0 + i;
        const int row = 4 * grp + tq;
        ad[i] = vt + row * 256 + (((2 * dt + (tp >> 1)) ^ ((row & 7) << 1)) * 16) + (tp & 1) * 8;
      }
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
    };
    auto retire_v = [&](s16x4 (&dst)[16]) {
      if constexpr (HB == 16) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(dst[0]), "+v"(dst[1]), "+v"(dst[2]), "+v"(dst[3]), "+v"(dst[4]), "+v"(dst[5]), "+v"(dst[6]), "+v"(dst[7]),
                       "+v"(dst[8]), "+v"(dst[9]), "+v"(dst[10]), "+v"(dst[11]), "+v"(dst[12]), "+v"(dst[13]), "+v"(dst[14]), "+v"(dst[15])
                     : : "memory");
      } else if constexpr (HB == 12) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(dst[0]), "+v"(dst[1]), "+v"(dst[2]), "+v"(dst[3]), "+v"(dst[4]), "+v"(dst[5]), "+v"(dst[6]), "+v"(dst[7]),
                       "+v"(dst[8]), "+v"(dst[9]), "+v"(dst[10]), "+v"(dst[11])
                     : : "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(dst[0]), "+v"(dst[1]), "+v"(dst[2]), "+v"(dst[3]), "+v"(dst[4]), "+v"(dst[5]), "+v"(dst[6]), "+v"(dst[7])
                     : : "memory");
      }
    };
    // lane holds, for query column l15 of each q tile, keys  kb*64 + 16t + 4*grp + r
    const int key0 = kb * PF_KEYS + 4 * grp;
    const bool has_hole = MASKED && (kb + 1) * PF_KEYS > first_neg_key;
    frag pf[2][2];                                                                   // [q tile][32-key step]
    // p = 2^s, ONE vector instruction per element ahead of the conversion: Q was multiplied by scale * log2(e) when it was
    // loaded and the accumulators of S were seeded with -m, so s arrives as (q.k * scale - m) * log2(e) (a lone wave hides
    // three vector instructions per MFMA, profiles/r4_mfma_gap_probe.txt — the multiply-add was the one too many).
    // The masked variant (diagonal / tail / hole tiles) is a separate wave-uniform path.
    // Lazy reference maximum: every lane takes the maximum of its own 16 scores (vector unit only); the cross-lane
    // reduction, the new reference and the rescale of O run only when some lane's maximum exceeds the reference by more
    // than 2^PF_LAZY_LOG2 (one wave-uniform branch on a ballot).  Any reference within that distance of the true
    // maximum gives the same result up to rounding: the final division uses sums taken against the same reference.
    // The reduction itself swaps lane rows in the vector unit (v_permlane32_swap / v_permlane16_swap): a ds_bpermute
    // queues behind the V^T reads just requested and cost ~300 cycles apiece, four times per tile.
    auto softmax_tile = [&](int qt) {
      f32x4 (&sc)[4] = s[qt];
      if constexpr (MASKED) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = key0 + 16 * t + r;
            if (has_hole && key >= first_neg_key) sc[t][r] = seed[qt];   // zero K rows: score 0 (relative to the reference)
            if (key > offset + row_pos[qt] || key >= kv_len) sc[t][r] = -INFINITY;
          }
      }
      // lane maximum of its 16 scores as a CHAIN max(max(m, x), y): hipcc folds each link into one v_max3 (8 instructions; a
      // balanced tree of fmaxf became 22 v_max + 5 v_max3).  Not inline asm: the hazard recogniser does not see into asm and
      // a v_max3 that reads S too early behind its MFMA returns stale scores.
      float mx = fmaxf(fmaxf(sc[0][0], sc[0][1]), sc[0][2]);
      mx = fmaxf(fmaxf(mx, sc[0][3]), sc[1][0]);
      mx = fmaxf(fmaxf(mx, sc[1][1]), sc[1][2]);
      mx = fmaxf(fmaxf(mx, sc[1][3]), sc[2][0]);
      mx = fmaxf(fmaxf(mx, sc[2][1]), sc[2][2]);
      mx = fmaxf(fmaxf(mx, sc[2][3]), sc[3][0]);
      mx = fmaxf(fmaxf(mx, sc[3][1]), sc[3][2]);
      mx = fmaxf(mx, sc[3][3]);
      // the scores are already relative to the reference m (log2 units; 0 while no key has been seen)
      if (__any(mx > thr[qt])) {
        mx = xor_max_16_32(mx);
        const bool first = m[qt] == -INFINITY;
        const float d = first ? mx : fmaxf(mx, 0.f);                              // -inf: still no visible key, nothing moves
        if (d != -INFINITY) {
          const float alpha = first ? 0.f : fast_exp2(-d);
          m[qt] = (first ? 0.f : m[qt]) + d;
          seed[qt] = -m[qt];
          seedv[qt] = f32x4{seed[qt], seed[qt], seed[qt], seed[qt]};
          thr[qt] = PF_LAZY_LOG2;
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) sc[t][r] -= d;
          osum[qt] *= alpha;
          if constexpr (MASKED) lhole[qt] *= alpha;
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) o[qt][dt] *= alpha;
        }
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        frag f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float p0 = fast_exp2(sc[2 * kk][r]);
          float p1 = fast_exp2(sc[2 * kk + 1][r]);
          if constexpr (MASKED) {                                                    // zero V rows: no contribution
            if (has_hole && key0 + 32 * kk + r >= first_neg_key) { lhole[qt] += p0; p0 = 0.f; }
            if (has_hole && key0 + 32 * kk + 16 + r >= first_neg_key) { lhole[qt] += p1; p1 = 0.f; }
          }
          f[r] = static_cast<T>(p0);
          f[4 + r] = static_cast<T>(p1);
          if (r & 1) dma_piece(qt * 4 + kk * 2 + (r >> 1));
        }
        pf[qt][kk] = f;
      }
    };
    // V^T batch 0 is requested before the softmax (its LDS latency hides behind the vector work), batch 1 before batch
    // 0's MFMAs; every wait is lgkmcnt(0) on data that has long landed.
    s16x4 vb0[16], vb1[16];
    if constexpr (FAST) { PF_STAMP(2); }
    // the page id of tile kb + 2 (a whole tile ahead of its use).  Requested HERE, behind the QK^T MFMAs: a scalar load in flight
    // makes every LDS wait an lgkmcnt(0) (scalar loads return out of order), and in front of the K fragment reads that meant
    // the first MFMA waited for all sixteen of them instead of the first four.
    if constexpr (FAST) phys_next = page_of_tile(kb + 2);
    issue_v(vb0, 0);
    softmax_tile(0);
    softmax_tile(1);
    if constexpr (FAST) { PF_STAMP(3); }

    // ---- O^T += V^T P^T -------------------------------------------------------------------------------------------
    auto pv_batch = [&](const s16x4 (&src)[16], int dt0) {
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
      if (dt0 == 0) {                                    // row sums: P against a V^T tile of ones
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          osum[0] = pf_mfma<T>::run(ones, pf[0][kk], osum[0]);
          osum[1] = pf_mfma<T>::run(ones, pf[1][kk], osum[1]);
        }
      }
    };
    retire_v(vb0);
    issue_v(vb1, DT / 2);                                // into the registers the scores just vacated
    if constexpr (FAST) { PF_STAMP(4); }
    pv_batch(vb0, 0);
    if constexpr (FAST) { PF_STAMP(5); }
    retire_v(vb1);
    if constexpr (FAST) { PF_STAMP(6); }
    pv_batch(vb1, DT / 2);
    if constexpr (FAST) { PF_STAMP(7); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // next tile landed
    if constexpr (FAST) { PF_STAMP(8); }
    __builtin_amdgcn_s_barrier();                         // ... and everyone is done reading this one
    if constexpr (FAST) { PF_STAMP(9); }
  };
  int kb_i = kb_lo;
  for (; kb_i + 1 < n_fast; ++kb_i) key_block(std::false_type{}, std::true_type{}, kb_i);     // the hot loop
  if (a.fast_stage) page_ready(phys_next);               // retire the last request before its register is reused
#ifdef PF_STAMPS
  if (blockIdx.x < 8192) {
    if (lane < 15) g_pf_stamps[(blockIdx.x * 4 + wave) * 16 + lane] = tacc;
    if (lane == 15) g_pf_stamps[(blockIdx.x * 4 + wave) * 16 + 15] = static_cast<unsigned>(kb_i);
  }
#endif
  PF_WG_MARK(1);
#ifdef PF_WG_STAMPS
  if (threadIdx.x == 0 && blockIdx.x < 8192 * 4) { g_pf_stamps[blockIdx.x * 16 + 4] = kb_i; g_pf_stamps[blockIdx.x * 16 + 5] = n_kb; }
#endif
  for (; kb_i < n_full; ++kb_i) key_block(std::false_type{}, std::false_type{}, kb_i);
  for (; kb_i < n_kb; ++kb_i) key_block(std::true_type{}, std::false_type{}, kb_i);
  PF_WG_MARK(2);

  // ---- finish: reduce the row sums over the 4 lane groups, normalise, store ----------------------------------
  // O^T leaves the accumulators as 8-byte pieces at a row stride (a store instruction would touch 64 separate lines), so
  // each wave transposes its 32 rows through a private LDS region (the tile buffers are free now; rows padded to 272 B:
  // the 8-byte writes of 16 rows then spread over all banks) and stores whole 2 * dim-byte rows, 16 bytes per lane.
  if constexpr (SPLIT) {
    // un-normalised partials of this key slice: row r of the block (= wave * 32 + qt * 16 + l15), dims 16 dt + 4 grp .. + 3
    const int64_t item = static_cast<int64_t>(blockIdx.x);
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const int r = wave * 32 + qt * 16 + l15;
      const float ls = osum[qt][0] + xor_sum_16_32(lhole[qt]);
      float* po = a.ws_o + (item * 128 + r) * a.dim + grp * 4;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) *reinterpret_cast<f32x4*>(po + dt * 16) = o[qt][dt];
      if (grp == 0) {
        a.ws_ml[(item * 128 + r) * 2 + 0] = m[qt];      // (-inf for a slice that saw no visible key)
        a.ws_ml[(item * 128 + r) * 2 + 1] = ls;
      }
    }
    return;
  }
  constexpr int OROW = 272;                       // (68 dwords: the 16 rows of a write land 4 banks apart; 288 left rows l and l + 8 on one bank pair)
  lds_c* stage_o = smem + wave * (32 * OROW);
  typedef typename vec_of<T, 4>::type V4;
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const float inv = 1.0f / (osum[qt][0] + xor_sum_16_32(lhole[qt]));
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
    constexpr int CPR = DT * 2;                          // 16-byte chunks per row (dim / 8)
    constexpr int RPI = 64 / CPR;                        // rows per store instruction
    const int sub = lane / CPR, ch = lane % CPR;
#pragma unroll
    for (int i = 0; i < (32 + RPI - 1) / RPI; ++i) {
      const int row = i * RPI + sub;                     // row of this wave (lanes past RPI * CPR idle: head_dim 96)
      if (sub >= RPI || row >= 32) continue;
      const int r = wave * 32 + row;
      const int pos = qb * QPB + (r % QPB);
      if (pos >= q_len) continue;
      const int g = r / QPB;
      const int head = a.abab ? g * a.hkv + kvh : kvh * G + g;
      const V8 v = *reinterpret_cast<const __attribute__((address_space(3))) V8*>(stage_o + row * OROW + ch * 16);
      *reinterpret_cast<V8*>(static_cast<T*>(a.out) + (static_cast<int64_t>(q_start + pos) * a.hq + head) * a.dim + ch * 8) = v;
    }
  }
  PF_WG_MARK(3);
#ifdef PF_WG_STAMPS
  if (threadIdx.x == 0 && blockIdx.x < 8192 * 4) g_pf_stamps[blockIdx.x * 16 + 7] = static_cast<unsigned>(__builtin_amdgcn_s_memrealtime());
#endif
}

// Combine the key slices of a block (PrefillArgs::ksplit): PF_MERGE_SPLIT workgroups per (query block, kv head, sequence), each
// 128 / PF_MERGE_SPLIT rows, thread = (row, 4 dims); slices in index order (fixed association: the same bits from run to run).
// (Round 3: it was one workgroup per block — 128 workgroups on the chunked-prefill case, each thread walking 16 items of
// dependent loads: 43 us beside a 145 us attention launch.  Sixteen times the workgroups and the slices of an item loaded four
// at a time before they are used.)
constexpr int PF_MERGE_SPLIT = 16;
template <typename T, int G>
__global__ __launch_bounds__(256) void prefill_merge_kernel(PrefillArgs a) {
  constexpr int QPB = 128 / G;
  constexpr int ROWS = 128 / PF_MERGE_SPLIT;
  const int inner = a.hkv * a.batch;
  const int wg = blockIdx.x / PF_MERGE_SPLIT, part = blockIdx.x % PF_MERGE_SPLIT;
  const int qb = a.n_qb - 1 - wg / inner;
  const int rem = wg % inner;
  const int kvh = rem % a.hkv, b = (rem / a.hkv + a.skew * (wg / inner)) % a.batch;
  const int q_start = a.cu_q[b];
  const int q_len = a.cu_q[b + 1] - q_start;
  const int kv_len = a.cu_kv ? a.cu_kv[b + 1] - a.cu_kv[b] : q_len;
  if (qb * QPB >= q_len || kv_len <= 0) return;          // (rows the attention launch zeroed or never owned)
  const int per_row = a.dim / 4;
  typedef typename vec_of<T, 4>::type V4;
  const int64_t slice0 = static_cast<int64_t>(wg) * a.ksplit;
  for (int i = threadIdx.x; i < ROWS * per_row; i += 256) {
    const int r = part * ROWS + i / per_row, d0 = (i % per_row) * 4;
    const int pos = qb * QPB + (r % QPB);
    if (pos >= q_len) continue;
    const int g = r / QPB;
    const int head = a.abab ? g * a.hkv + kvh : kvh * G + g;
    float mx = -INFINITY;
    for (int sl = 0; sl < a.ksplit; ++sl) mx = fmaxf(mx, a.ws_ml[((slice0 + sl) * 128 + r) * 2]);
    f32x4 num = {0.f, 0.f, 0.f, 0.f};
    float den = 0.f;
    for (int sl0 = 0; sl0 < a.ksplit; sl0 += 4) {
      f32x4 v[4];
      float ms[4], ls[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {                      // all loads of the four slices first (a slice past the end repeats the last one, weight 0)
        const int64_t row = (slice0 + min(sl0 + u, a.ksplit - 1)) * 128 + r;
        ms[u] = sl0 + u < a.ksplit ? a.ws_ml[row * 2] : -INFINITY;
        ls[u] = a.ws_ml[row * 2 + 1];
        v[u] = *reinterpret_cast<const f32x4*>(a.ws_o + row * a.dim + d0);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (ms[u] == -INFINITY) continue;
        const float w = exp2f(ms[u] - mx);
        den = fmaf(w, ls[u], den);
        num += v[u] * w;
      }
    }
    const float inv = den > 0.f ? 1.0f / den : 0.f;
    V4 ov;
#pragma unroll
    for (int e = 0; e < 4; ++e) ov[e] = static_cast<T>(num[e] * inv);
    *reinterpret_cast<V4*>(static_cast<T*>(a.out) + (static_cast<int64_t>(q_start + pos) * a.hq + head) * a.dim + d0) = ov;
  }
}

}  // namespace mojo
#ifdef MOJO_HIP_BUILD_EXPERIMENTS        // the default decomposition on 32x32x16 MFMAs: parity-green, measured 4-5 % slower (DESIGN Appendix A #10c)
#include "experiments/paged_prefill_m32.h"
#endif
#ifdef MOJO_HIP_BUILD_EXPERIMENTS        // phase-alternating kernel, measured 5-20 % slower (DESIGN Appendix A #10a): opt-in build only
#include "experiments/paged_prefill_pp.h"
#else
namespace mojo { constexpr int PP_TABLE = 0; }
#endif
#ifdef MOJO_HIP_BUILD_EXPERIMENTS        // one-wave-per-SIMD kernel of round 4: parity-green, measured 2-30 % slower (DESIGN §4.4, Appendix A #10b)
#include "experiments/paged_prefill_w64.h"
#else
namespace mojo { constexpr int W64_TABLE = 0; }
#endif
namespace mojo {

template <typename T, int G, int DK>
static void launch_pf(const PrefillArgs& a, dim3 grid, hipStream_t s) {
  static std::atomic<uint64_t> attr_set{0};
  if (first_call_on_device(attr_set)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&prefill_kernel<T, G, DK, false>), hipFuncAttributeMaxDynamicSharedMemorySize, PF_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&prefill_kernel<T, G, DK, true>), hipFuncAttributeMaxDynamicSharedMemorySize, PF_LDS);
  }
#ifdef MOJO_HIP_BUILD_EXPERIMENTS
  if (a.pp) {
    static std::atomic<uint64_t> pp_attr_set{0};
    if (first_call_on_device(pp_attr_set))
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&prefill_pp_kernel<T, G, DK>), hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS);
    hipLaunchKernelGGL((prefill_pp_kernel<T, G, DK>), grid, dim3(512), PP_LDS, s, a);
    return;
  }
#endif
#ifdef MOJO_HIP_BUILD_EXPERIMENTS
  if constexpr (DK == 4) {
    if (a.w64) {
      static std::atomic<uint64_t> w64_attr_set{0};
      if (first_call_on_device(w64_attr_set))
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&prefill_w64_kernel<T, G>), hipFuncAttributeMaxDynamicSharedMemorySize, W64_LDS);
      hipLaunchKernelGGL((prefill_w64_kernel<T, G>), grid, dim3(256), W64_LDS, s, a);
      return;
    }
  }
#endif
#ifdef MOJO_HIP_BUILD_EXPERIMENTS
  if constexpr (DK == 4) {
    if (a.m32) {
      static std::atomic<uint64_t> m32_attr_set{0};
      if (first_call_on_device(m32_attr_set)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&prefill_m32_kernel<T, G, false>), hipFuncAttributeMaxDynamicSharedMemorySize, PF_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&prefill_m32_kernel<T, G, true>), hipFuncAttributeMaxDynamicSharedMemorySize, PF_LDS);
      }
      if (a.ksplit > 1) {
        hipLaunchKernelGGL((prefill_m32_kernel<T, G, true>), grid, dim3(256), PF_LDS, s, a);
        hipLaunchKernelGGL((prefill_merge_kernel<T, G>), dim3(static_cast<unsigned>(a.n_qb * a.hkv * a.batch * PF_MERGE_SPLIT)), dim3(256), 0, s, a);
      } else {
        hipLaunchKernelGGL((prefill_m32_kernel<T, G, false>), grid, dim3(256), PF_LDS, s, a);
      }
      return;
    }
  }
#endif
  if (a.ksplit > 1) {
    hipLaunchKernelGGL((prefill_kernel<T, G, DK, true>), grid, dim3(256), PF_LDS, s, a);
    hipLaunchKernelGGL((prefill_merge_kernel<T, G>), dim3(static_cast<unsigned>(a.n_qb * a.hkv * a.batch * PF_MERGE_SPLIT)), dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL((prefill_kernel<T, G, DK, false>), grid, dim3(256), PF_LDS, s, a);
  }
}

template <typename T, int G>
static int dispatch_dk(const PrefillArgs& a, dim3 grid, hipStream_t s) {
  switch (a.dim) {
    case 64: launch_pf<T, G, 2>(a, grid, s); break;
    case 96: launch_pf<T, G, 3>(a, grid, s); break;
    case 128: launch_pf<T, G, 4>(a, grid, s); break;
    default: MOJO_REQUIRE(false, MOJO_EUNSUPPORTED, "paged_prefill_gqa: head_dim %d (64, 96, 128)", a.dim);
  }
  MOJO_CHECK_LAUNCH("paged_prefill_gqa");
  note_launch("prefill:%s:%s:ksplit%d", a.pp ? "pp" : a.w64 ? "w64" : a.m32 ? "m32" : "default", a.fast_stage ? "fast_stage" : "general_stage", a.ksplit);
  return MOJO_OK;
}

// Experiments build only: the one-wave-per-SIMD kernel (experiments/paged_prefill_w64.h) takes the launches it applies to
// (head_dim 128, pages of >= 16 keys, unsplit, block tables of <= W64_TABLE - 16 pages) when MOJO_HIP_PREFILL_W64=1.
static bool prefill_use_w64(int64_t max_q, int64_t batch, int hkv, int G) {
#ifdef MOJO_HIP_BUILD_EXPERIMENTS
  if (MOJO_SWITCH("MOJO_HIP_PREFILL_W64", 0) == 1) return true;
#endif
  return false;
}

// Experiments build only: the 32x32x16 form of the default decomposition (experiments/paged_prefill_m32.h), head_dim 128,
// when MOJO_HIP_PREFILL_M32=1.
static bool prefill_use_m32() {
#ifdef MOJO_HIP_BUILD_EXPERIMENTS
  if (MOJO_SWITCH("MOJO_HIP_PREFILL_M32", 0) == 1) return true;
#endif
  return false;
}

static bool prefill_use_pp(int64_t max_q) {
#ifdef MOJO_HIP_BUILD_EXPERIMENTS
  if (MOJO_SWITCH("MOJO_HIP_PREFILL_PP", 0) == 1) return true;
#endif
  return false;
}

template <typename T>
static int dispatch_g(PrefillArgs a, int G, int64_t batch, int64_t max_q, hipStream_t s) {
  // Experiments build only: the phase-alternating kernel takes unsplit launches over pages it can stage with scalar page
  // ids when MOJO_HIP_PREFILL_PP=1.
  a.pp = (a.ksplit == 1 && a.fast_stage && a.max_pages <= PP_TABLE && prefill_use_pp(max_q)) ? 1 : 0;
  a.w64 = (!a.pp && a.ksplit == 1 && a.fast_stage && a.dim == 128 && a.max_pages <= W64_TABLE - 16 &&
           prefill_use_w64(max_q, batch, a.hkv, G)) ? 1 : 0;
  a.m32 = (!a.pp && !a.w64 && a.dim == 128 && prefill_use_m32()) ? 1 : 0;
  const int qpb = ((a.pp || a.w64) ? 256 : 128) / G;
  const int64_t n_qb = ceil_div(max_q, qpb);
  const int64_t n_zero = ceil_div(a.total_tokens, static_cast<int64_t>(PF_ZERO_TOKENS));
  MOJO_REQUIRE((n_qb * a.hkv * batch + n_zero) * a.ksplit < (int64_t{1} << 31), MOJO_EUNSUPPORTED, "paged_prefill_gqa: grid limit");
  a.batch = static_cast<int>(batch);
  a.n_qb = static_cast<int>(n_qb);
  a.skew = 1;
#ifdef MOJO_HIP_BUILD_EXPERIMENTS           // the sequence rotation per query-block level (placement A/B)
  a.skew = MOJO_SWITCH("MOJO_HIP_PREFILL_SKEW", 1) == 0 ? 0 : 1;
#endif
  dim3 grid(static_cast<unsigned>((n_qb * a.hkv * batch + n_zero) * a.ksplit));
  switch (G) {
    case 1: return dispatch_dk<T, 1>(a, grid, s);
    case 2: return dispatch_dk<T, 2>(a, grid, s);
    case 4: return dispatch_dk<T, 4>(a, grid, s);
    case 8: return dispatch_dk<T, 8>(a, grid, s);
    default: MOJO_REQUIRE(false, MOJO_EUNSUPPORTED, "paged_prefill_gqa: group size %d (1, 2, 4, 8)", G);
  }
}

// Key split: a launch of few, long blocks (a chunked prefill of one or two sequences against a long cache: 128 workgroups of
// ~260 key tiles each) leaves half of the chip's 512 workgroup slots empty and ends on its longest block.  Cut every block's
// key range into as many slices as fill the slots ONCE (512 new tokens + 16 384 cached, 32 q / 8 kv heads: 128 blocks; slices
// 1 / 2 / 3 / 4 / 5 / 6 / 8 -> 339 / 197 / 199 / 185 / 223 / 210 / 229 us: a second round of workgroups only adds partials to
// write and merge) and still leave >= 8 tiles (512 keys) per slice.  MOJO_HIP_PREFILL_KSPLIT=<n> forces (1 = off).
static int prefill_ksplit(int64_t blocks, int64_t kv_cap) {
  if (const int v = static_cast<int>(MOJO_SWITCH("MOJO_HIP_PREFILL_KSPLIT", 0)); v >= 1) return v > 16 ? 16 : v;
  if (blocks <= 0 || blocks > 256) return 1;
  int64_t ks = 512 / blocks;
  const int64_t by_keys = kv_cap / 512;
  if (ks > by_keys) ks = by_keys;
  if (ks > 16) ks = 16;
  return ks < 2 ? 1 : static_cast<int>(ks);
}

static void prefill_plan(int64_t total_tokens, int64_t batch, int64_t q_heads, int64_t kv_heads, int64_t block_size,
                         int64_t max_blocks_per_seq, int64_t max_q_len_hint, int64_t max_kv_len_hint, int64_t& n_qb, int& ksplit) {
  const int G = static_cast<int>(q_heads / kv_heads);
  const int64_t max_q = (max_q_len_hint > 0 && max_q_len_hint < total_tokens) ? max_q_len_hint : total_tokens;
  n_qb = ceil_div(max_q, 128 / (G > 0 ? G : 1));
  int64_t kv_cap = block_size * max_blocks_per_seq;
  if (max_kv_len_hint > 0 && max_kv_len_hint < kv_cap) kv_cap = max_kv_len_hint;
  ksplit = prefill_ksplit(n_qb * kv_heads * batch, kv_cap);
}

}  // namespace mojo

using namespace mojo;

extern "C" int64_t mojo_hip_paged_prefill_gqa_workspace_bytes(int64_t total_tokens, int64_t batch, int64_t q_heads,
                                                              int64_t kv_heads, int64_t head_dim, int64_t block_size,
                                                              int64_t max_blocks_per_seq, int64_t max_q_len_hint,
                                                              int64_t max_kv_len_hint) {
  if (total_tokens <= 0 || batch <= 0 || q_heads <= 0 || kv_heads <= 0 || q_heads % kv_heads) return 0;
  int64_t n_qb;
  int ks;
  prefill_plan(total_tokens, batch, q_heads, kv_heads, block_size, max_blocks_per_seq, max_q_len_hint, max_kv_len_hint, n_qb, ks);
  if (ks <= 1) return 0;
  return n_qb * kv_heads * batch * ks * 128 * (head_dim + 2) * static_cast<int64_t>(sizeof(float)) + 64;
}

extern "C" int mojo_hip_paged_prefill_gqa(const void* query, const void* key_cache, const void* value_cache,
                                          const int32_t* cu_q_lens, const int32_t* cu_total_seq_lens,
                                          const int32_t* block_tables, void* out, int64_t total_tokens, int64_t batch,
                                          int64_t q_heads, int64_t kv_heads, int64_t head_dim, int64_t block_size,
                                          int64_t max_blocks_per_seq, int64_t block_table_stride,
                                          int64_t cache_block_stride, int64_t cache_head_stride,
                                          int64_t cache_token_stride, int64_t max_q_len_hint, int64_t max_kv_len_hint,
                                          float softmax_scale, int layout_abab, int dtype, void* workspace,
                                          int64_t workspace_bytes, mojo_stream_t stream) {
  if (total_tokens == 0) return MOJO_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  MOJO_REQUIRE(query && key_cache && value_cache && cu_q_lens && block_tables && out, MOJO_EINVAL,
               "paged_prefill_gqa: null pointer");
  MOJO_REQUIRE(q_heads > 0 && kv_heads > 0 && q_heads % kv_heads == 0 && batch >= 0, MOJO_EINVAL,
               "paged_prefill_gqa: bad head counts Hq=%lld Hkv=%lld", (long long)q_heads, (long long)kv_heads);
  MOJO_REQUIRE(dtype == MOJO_BF16 || dtype == MOJO_F16, MOJO_EUNSUPPORTED, "paged_prefill_gqa: dtype %d (bf16/fp16 only)", dtype);
  MOJO_REQUIRE(block_size % 4 == 0, MOJO_EUNSUPPORTED, "paged_prefill_gqa: block_size %lld must be a multiple of 4",
               (long long)block_size);
  MOJO_REQUIRE(cache_token_stride % 8 == 0 && cache_head_stride % 8 == 0 && cache_block_stride % 8 == 0 &&
                   aligned_to(key_cache, 16) && aligned_to(value_cache, 16) && aligned_to(query, 16) && aligned_to(out, 16),
               MOJO_EUNSUPPORTED, "paged_prefill_gqa: tensors must be 16-byte aligned with 16-byte row strides");
  if (batch == 0) {                                      // no sequences: every row is padding
    if (hipMemsetAsync(out, 0, static_cast<size_t>(total_tokens * q_heads * head_dim * 2), s) != hipSuccess) {
      set_error("paged_prefill_gqa: memset failed");
      return MOJO_ELAUNCH;
    }
    return MOJO_OK;
  }
  PrefillArgs a;
  a.q = query; a.kc = key_cache; a.vc = value_cache; a.out = out; a.cu_q = cu_q_lens; a.cu_kv = cu_total_seq_lens;
  a.total_tokens = total_tokens;
  a.tables = block_tables; a.table_stride = block_table_stride; a.c_blk = cache_block_stride;
  a.c_head = cache_head_stride; a.c_tok = cache_token_stride;
  a.hq = static_cast<int>(q_heads); a.hkv = static_cast<int>(kv_heads); a.dim = static_cast<int>(head_dim);
  a.page = static_cast<int>(block_size);
  a.page_shift = (block_size & (block_size - 1)) == 0 ? __builtin_ctzll(block_size) : -1;
  a.max_pages = static_cast<int>(max_blocks_per_seq);
  a.scale_log2 = softmax_scale * 1.4426950408889634f;
  a.abab = layout_abab ? 1 : 0;
  const bool no_fs = MOJO_SWITCH("MOJO_HIP_PREFILL_FAST_STAGE", 1) == 0;        // 0: general staging everywhere (tests)
  a.fast_stage = (a.page_shift >= 4 && cache_token_stride * 16 * 2 + 256 < (int64_t{1} << 31) && !no_fs) ? 1 : 0;
  int64_t max_q = (max_q_len_hint > 0 && max_q_len_hint < total_tokens) ? max_q_len_hint : total_tokens;
  const int G = static_cast<int>(q_heads / kv_heads);
  {
    int64_t n_qb;
    int ks;
    prefill_plan(total_tokens, batch, q_heads, kv_heads, block_size, max_blocks_per_seq, max_q_len_hint, max_kv_len_hint, n_qb, ks);
    const int64_t rows = n_qb * kv_heads * batch * ks * 128;
    const int64_t need = rows * (head_dim + 2) * static_cast<int64_t>(sizeof(float));
    if (ks > 1 && !(workspace && workspace_bytes >= need && aligned_to(workspace, 16))) ks = 1;   // no workspace: run unsplit
    a.ksplit = ks;
    a.ws_o = ks > 1 ? static_cast<float*>(workspace) : nullptr;
    a.ws_ml = ks > 1 ? a.ws_o + rows * head_dim : nullptr;
  }
  return dtype == MOJO_BF16 ? dispatch_g<bf16_t>(a, G, batch, max_q, s) : dispatch_g<f16_t>(a, G, batch, max_q, s);
}

#if defined(PF_STAMPS) || defined(PF_WG_STAMPS)
extern "C" int mojo_hip_debug_prefill_stamps(unsigned* host_out, int64_t count) {
  if (hipDeviceSynchronize() != hipSuccess) return MOJO_ELAUNCH;
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mojo::g_pf_stamps), static_cast<size_t>(count) * 4) == hipSuccess ? MOJO_OK : MOJO_ELAUNCH;
}
#endif
