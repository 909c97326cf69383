// In-launch combine of split-K partial slabs: "the last K slice to arrive sums ALL slices in index order".
//
// Round 2 summed the slices of a decode-sized product in a second launch (a kernel boundary of ~1.5-1.9 us plus the launch
// itself on products that take 7-30 us in all).  Here every slice workgroup stores its fp32 / int32 accumulators to its slab
// WRITE-THROUGH (buffer stores with sc1: the bytes leave the XCD's L2, no release fence is needed), drains them, and draws a
// ticket from an agent-scope counter of its output tile; the workgroup that draws the last ticket reads every slab of the
// tile with sc1 loads (they bypass its CU's L1; its own slab too, so the sum is formed in slice order 0 .. splitk - 1 whoever
// happens to be last: the same bits as the finalize kernel, run to run) and writes the tile.  Nobody waits for anybody: no
// residency requirement, no deadlock next to a collective's kernels.  (CDNA4 guide, "In-launch split-K reduction", sc1 form.)
//
// The counters must be zero on entry without a launch of their own: they live in a static device array of SLOTS x TILES words
// (zero at module load), the last arriver of a tile zeroes the tile's word again, and every call takes the next slot round
// robin on the host — a call can only meet a stale word if SLOTS calls are in flight at once.  (Under graph replay the slot
// is baked into the node; replays of one graph are ordered on their stream.)
#pragma once
#include <atomic>

#include "../common.h"

namespace mojo {

constexpr int SK_SLOTS = 64, SK_TILES = 4096;
static __device__ unsigned g_splitk_tickets[SK_SLOTS * SK_TILES];

// MEASURED (round 3, HIP-graph replay, one MI355X; benchmarks/one.py bench_dense_decode / bench_quant_gemm, two launches ->
// one): bf16 64 x 8192 x 8192 35.5 -> 35.8 us, 64 x 14336 x 4096 29.7 -> 31.4, 1 x 8192 x 8192 30.5 -> 29.3; int8
// 128 x 7168 x 4096 15.9 -> 17.7, 32 x 7168 x 4096 10.1 -> 12.0.  The slabs of one 64-column tile are 16-32 KiB per slice
// and 4-16 slices: the last arriver reads them serially behind everybody else's work (the guide's "a few tens of KB per
// tile" limit), and under graph replay the finalize launch it replaces costs ~1.5 us + a kernel that finds the slabs in L2.
// So the combine is OFF by default (MOJO_HIP_SPLITK_INLAUNCH=1 turns it on); it is kept, tested bit for bit against the
// two-launch form, for products whose tiles carry less slab.
//
// next ticket slot of this translation unit's counter array, or -1 when the combine is off (default; read per call) or the
// launch has more tiles than a slot holds: the caller then runs its finalize kernel
static inline int splitk_take_slot(int64_t tiles) {
  if (tiles > SK_TILES) return -1;
  if (MOJO_SWITCH("MOJO_HIP_SPLITK_INLAUNCH", 0) != 1) return -1;
  static std::atomic<unsigned> next{0};
  return static_cast<int>(next.fetch_add(1, std::memory_order_relaxed) % SK_SLOTS);
}

typedef __amdgpu_buffer_rsrc_t sk_rsrc_t;
__device__ __forceinline__ sk_rsrc_t splitk_rsrc(void* slab, long long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(slab, 0, static_cast<int>(bytes), 0x00020000);
}
__device__ __forceinline__ void splitk_store16(sk_rsrc_t r, long long byte_off, u32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, static_cast<int>(byte_off), 0, 16);          // aux 16 = sc1: write-through
}
__device__ __forceinline__ u32x4 splitk_load16(sk_rsrc_t r, long long byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, static_cast<int>(byte_off), 0, 16);       // sc1: served by L2, never by this CU's L1
}

// Call once per workgroup after ITS slab stores, from uniform control flow.  True (in every thread) in the workgroup whose
// ticket is the tile's last: every other slice's stores are then visible to sc1 loads.  `s_flag`: one int of LDS.
__device__ __forceinline__ bool splitk_arrive(int slot, int tile, int splitk, int* s_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // EVERY storing wave drains its write-through stores ...
  __syncthreads();                                       // ... before the one lane that signals for all of them
  if (threadIdx.x == 0) {
    unsigned* cnt = g_splitk_tickets + slot * SK_TILES + tile;
    const unsigned t = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = t == static_cast<unsigned>(splitk - 1);
    if (last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // zero again for the slot's next user
    *s_flag = last;
  }
  __syncthreads();                                       // the other waves load only behind the barrier the adding wave joins
  return *s_flag != 0;
}

}  // namespace mojo
