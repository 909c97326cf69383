// Direct reduce-scatter / all-gather between the GPUs of one node over HIP-IPC peer buffers — the exchange step of
// MojoGemmAllReduce and MojoGemmReduceScatter without a ring.
//
// Role in the reference: `runtime/comm_context.py:107-153` (`allocate_peer_mem`: symmetric buffers cached per
// (group, size)) + the pull-and-add loops of `backends/ttx/kernels/npu/a2/gemm_allreduce.py:85-145` and
// `gemm_reduce_scatter.py:108-156`.  Built here for xGMI: every GPU has a direct link to every other GPU of the node, so
// rank r PULLS its 1/ws share of every peer's partial product over ws-1 links at once (reduce-scatter), and for the
// all-reduce pulls the other ranks' reduced shares the same way (all-gather).  A ring moves 2(ws-1)/ws of the payload
// over ONE link per GPU; the direct form moves the same bytes over ws-1 links in parallel (110 us against
// 767 us for 64 MiB at ws = 8).
//
// Memory: each rank owns one symmetric allocation (data area + flag words), exported with hipIpcGetMemHandle and opened
// by every peer.  All remote accesses are READS of data plus WRITES of 4-byte flags.
//
// Ordering (no host involvement once the kernels are enqueued):
//   producer  : GEMM chunk kernel -> [kernel boundary: its stores are written back] -> signal kernel:
//               system-scope fence, then system-scope store of `epoch` to flag[kind][me][chunk] in EVERY peer's flag area
//   consumer  : spins on ITS OWN flag words (local memory, written remotely) until they reach `epoch`, system-scope
//               acquire fence, then reads the peers' data.
// Epochs grow by one per operator call and never reset, so flags need no clearing; the data area is double-buffered by
// epoch parity, which is enough because a rank can only be one call ahead of its slowest peer (every call waits for
// every peer's flags of that call).
//
// Captured mode (HIP-graph replay; round 4): a captured launch has its arguments baked in, so neither an epoch that grows on
// the host nor a data offset that alternates with its parity can be replayed.  A captured call therefore passes epoch 0 =
// "take the epoch from the control area": `mojo_hip_peer_begin` opens the call by incrementing a device-resident epoch word
// (every later kernel of the call reads it), and instead of the parity halves there is ONE data area guarded by a third flag
// kind: at the end of a call a rank tells every peer "I have finished reading your data of this call" (kind 2), and the
// begin step of the next call waits for those flags before the GEMM overwrites the area.  Eager and captured calls use
// separate exchange objects (comm/peer.py), so the two epoch sequences never meet.
//
// Liveness: every wait is bounded (wall-clock ticks of the 100 MHz constant counter).  On expiry — or when the sticky
// error word is already set — the waiter stops waiting, poisons what it would have produced with NaN, still raises its
// own flags (so peers do not time out in cascade) and the grid drains.  The host reads the error word when it next looks.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace mojo {

constexpr int PEER_MAX = 16;              // ranks of one node
constexpr int PEER_MAX_CHUNKS = 64;       // row chunks of one operator call
constexpr int PEER_FLAG_KINDS = 3;        // 0: partial product of (rank, chunk) is in memory; 1: reduced share is; 2: rank has finished READING the call's data (captured mode, chunk 0)
constexpr int PEER_FLAG_WORDS = PEER_FLAG_KINDS * PEER_MAX * PEER_MAX_CHUNKS;
// words behind the flags: [0] sticky error, [1..] one arrival counter per chunk for the "last workgroup signals" step
constexpr int PEER_ERR_WORD = PEER_FLAG_WORDS;
constexpr int PEER_EPOCH_WORD = PEER_FLAG_WORDS + 1;   // captured mode: the epoch of the call in flight (0 before the first)
constexpr int PEER_CNT_WORD = PEER_FLAG_WORDS + 16;
constexpr int PEER_CTRL_WORDS = PEER_CNT_WORD + PEER_MAX_CHUNKS;

struct PeerPtrs {
  char* data[PEER_MAX];                   // base of every rank's data area (index = rank; own entry = local pointer)
  uint32_t* flags[PEER_MAX];              // base of every rank's control words
};

__device__ __forceinline__ uint32_t* flag_word(uint32_t* base, int kind, int src, int chunk) {
  return base + (kind * PEER_MAX + src) * PEER_MAX_CHUNKS + chunk;
}

// true when the flag reached `epoch`; false on timeout / sticky error (then *err is set)
__device__ bool wait_flag(uint32_t* f, uint32_t epoch, uint32_t* err, long long timeout_ticks) {
  const long long t0 = wall_clock64();
  for (;;) {
    const uint32_t v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (static_cast<int32_t>(v - epoch) >= 0) return true;                 // wrap-safe "v >= epoch"
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
    if (wall_clock64() - t0 > timeout_ticks) {
      __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
    __builtin_amdgcn_s_sleep(20);
  }
}

// epoch argument 0 = captured mode: the call's epoch lives in this rank's control area (written by peer_begin_kernel)
__device__ __forceinline__ uint32_t resolve_epoch(uint32_t* my_flags, uint32_t epoch) {
  return epoch ? epoch : __hip_atomic_load(my_flags + PEER_EPOCH_WORD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- begin (captured mode): wait until every peer has finished reading the previous call's data, then open the next epoch --
__global__ __launch_bounds__(64) void peer_begin_kernel(PeerPtrs pp, int ws, int rank, long long timeout_ticks) {
  uint32_t* my = pp.flags[rank];
  const uint32_t cur = __hip_atomic_load(my + PEER_EPOCH_WORD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int p = threadIdx.x;
  if (p < ws && p != rank && cur != 0) (void)wait_flag(flag_word(my, 2, p, 0), cur, my + PEER_ERR_WORD, timeout_ticks);
  __syncthreads();
  if (p == 0) {
    uint32_t nxt = cur + 1;
    if (nxt == 0) nxt = 1;                                                  // 0 is "take it from here" in the other kernels
    __hip_atomic_store(my + PEER_EPOCH_WORD, nxt, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- signal: "my partial product of this chunk is in memory" (kind 0) ------------------------------------------
__global__ __launch_bounds__(64) void peer_signal_kernel(PeerPtrs pp, int ws, int rank, int kind, int chunk, uint32_t epoch) {
  const int p = threadIdx.x;
  if (p >= ws) return;
  epoch = resolve_epoch(pp.flags[rank], epoch);
  __atomic_thread_fence(__ATOMIC_SEQ_CST);                                  // (system scope: the default of the builtin)
  __hip_atomic_store(flag_word(pp.flags[p], kind, rank, chunk), epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- pull-and-add: out[rows, n] = sum over ranks of partial_p[src_off + ...] -----------------------------------
// The source block is contiguous [rows][n] at byte offset src_off of every rank's data area.  dst is local.
// write_back: also store the reduced values over this rank's own partial (peers pull them from there in the gather
// step) and, once every workgroup of the launch has done so, raise flag kind 1 at every peer.
template <typename T>
__global__ __launch_bounds__(256) void peer_reduce_kernel(PeerPtrs pp, int ws, int rank, int chunk, uint32_t epoch,
                                                          long long src_off, long long rows, int n, T* dst,
                                                          long long ld_dst, int write_back, long long timeout_ticks) {
  typedef typename vec_of<T, 16 / sizeof(T)>::type V;
  constexpr int VE = 16 / sizeof(T);
  __shared__ int s_ok;
  uint32_t* my_flags = pp.flags[rank];
  uint32_t* err = my_flags + PEER_ERR_WORD;
  epoch = resolve_epoch(my_flags, epoch);
  if (threadIdx.x == 0) s_ok = 1;
  __syncthreads();
  if (threadIdx.x < ws && threadIdx.x != rank) {
    if (!wait_flag(flag_word(my_flags, 0, threadIdx.x, chunk), epoch, err, timeout_ticks)) s_ok = 0;
  }
  __syncthreads();
  __atomic_thread_fence(__ATOMIC_ACQUIRE);                                  // system scope: peers' data written before their flags
  const bool ok = s_ok != 0;
  const int vec_per_row = n / VE;
  const long long total = rows * vec_per_row;
  for (long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<long long>(gridDim.x) * 256) {
    const long long row = i / vec_per_row;
    const int col = static_cast<int>(i - row * vec_per_row) * VE;
    const long long off = src_off + (row * n + col) * static_cast<long long>(sizeof(T));
    V part[PEER_MAX];
#pragma unroll
    for (int p = 0; p < PEER_MAX; ++p)
      if (p < ws) part[p] = ok ? *reinterpret_cast<const V*>(pp.data[p] + off) : V{};
    float acc[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[e] = 0.f;
#pragma unroll
    for (int p = 0; p < PEER_MAX; ++p) {                                    // rank order: one fixed association on every rank
      if (p < ws) {
#pragma unroll
        for (int e = 0; e < VE; ++e) acc[e] += elt<T>::to_f(vget<T, VE>(part[p], e));
      }
    }
    V o;
#pragma unroll
    for (int e = 0; e < VE; ++e) vset<T, VE>(o, e, elt<T>::from_f(ok ? acc[e] : NAN));
    *reinterpret_cast<V*>(dst + row * ld_dst + col) = o;
    if (write_back) *reinterpret_cast<V*>(pp.data[rank] + off) = o;
  }
  if (!write_back) return;
  // last workgroup to arrive raises "reduced share ready" at every peer
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t* cnt = my_flags + PEER_CNT_WORD + chunk;
    const uint32_t prev = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    s_ok = prev == gridDim.x - 1 ? 2 : 0;
    if (s_ok == 2) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (s_ok == 2 && threadIdx.x < ws) {
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    __hip_atomic_store(flag_word(pp.flags[threadIdx.x], 1, rank, chunk), epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ---- gather: copy every peer's reduced share of this chunk into the local output ---------------------------------
// Share p of the chunk = rows [row_lo(p), row_lo(p+1)) with row_lo(p) = rows * p / ws, contiguous at byte offset
// chunk_off + row_lo(p) * n * sizeof(T) of rank p's data area; it lands at dst + row_lo(p) * ld_dst.
// blockIdx.y = index among the OTHER ranks (the own share was written by the reduce kernel).
template <typename T>
__global__ __launch_bounds__(256) void peer_gather_kernel(PeerPtrs pp, int ws, int rank, int chunk, uint32_t epoch,
                                                          long long chunk_off, long long rows, int n, T* dst,
                                                          long long ld_dst, long long timeout_ticks) {
  typedef typename vec_of<T, 16 / sizeof(T)>::type V;
  constexpr int VE = 16 / sizeof(T);
  __shared__ int s_ok;
  const int p = (rank + 1 + blockIdx.y) % ws;                               // start with the next rank: links are used evenly
  uint32_t* my_flags = pp.flags[rank];
  epoch = resolve_epoch(my_flags, epoch);
  const long long r0 = rows * p / ws, r1 = rows * (p + 1) / ws;
  if (r1 == r0) return;                                                     // rank p owns no row of this chunk: nothing to wait for
  if (threadIdx.x == 0) s_ok = wait_flag(flag_word(my_flags, 1, p, chunk), epoch, my_flags + PEER_ERR_WORD, timeout_ticks) ? 1 : 0;
  __syncthreads();
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  const bool ok = s_ok != 0;
  const int vec_per_row = n / VE;
  const long long total = (r1 - r0) * vec_per_row;
  const char* src = pp.data[p] + chunk_off;
  for (long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<long long>(gridDim.x) * 256) {
    const long long row = r0 + i / vec_per_row;
    const int col = static_cast<int>(i % vec_per_row) * VE;
    V v;
    if (ok) {
      v = *reinterpret_cast<const V*>(src + (row * n + col) * static_cast<long long>(sizeof(T)));
    } else {
#pragma unroll
      for (int e = 0; e < VE; ++e) vset<T, VE>(v, e, elt<T>::from_f(NAN));
    }
    *reinterpret_cast<V*>(dst + row * ld_dst + col) = v;
  }
}

// ---- pull: copy one contiguous block from every rank's data area into consecutive slots of a local buffer -------------
// (the all-gather of MojoAllGatherGemm: slot p of dst <- `bytes` at src_off of rank p's data area, after rank p's flag
// (kind, flag_chunk) reached `epoch`; blockIdx.y walks the ranks starting at the own one, which needs no wait)
__global__ __launch_bounds__(256) void peer_pull_kernel(PeerPtrs pp, int ws, int rank, int kind, int flag_chunk, uint32_t epoch,
                                                        long long src_off, long long bytes, char* dst, long long dst_stride,
                                                        int first, long long timeout_ticks) {
  __shared__ int s_ok;
  const int p = (rank + first + blockIdx.y) % ws;
  uint32_t* my_flags = pp.flags[rank];
  epoch = resolve_epoch(my_flags, epoch);
  if (threadIdx.x == 0)
    s_ok = (p == rank || wait_flag(flag_word(my_flags, kind, p, flag_chunk), epoch, my_flags + PEER_ERR_WORD, timeout_ticks)) ? 1 : 0;
  __syncthreads();
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  const bool ok = s_ok != 0;
  const char* src = pp.data[p] + src_off;
  char* out = dst + p * dst_stride;
  const long long vecs = bytes / 16;
  for (long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; i < vecs; i += static_cast<long long>(gridDim.x) * 256) {
    u32x4 v = {0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u};       // bf16 / fp16 / fp32 NaN patterns on a failed wait
    if (ok) v = *reinterpret_cast<const u32x4*>(src + i * 16);
    *reinterpret_cast<u32x4*>(out + i * 16) = v;
  }
}

static int fill_ptrs(PeerPtrs& pp, void* const* data, void* const* flags, int ws) {
  for (int i = 0; i < PEER_MAX; ++i) {
    pp.data[i] = i < ws ? static_cast<char*>(data[i]) : nullptr;
    pp.flags[i] = i < ws ? static_cast<uint32_t*>(flags[i]) : nullptr;
    if (i < ws && (!pp.data[i] || !pp.flags[i])) return MOJO_EINVAL;
  }
  return MOJO_OK;
}

static int peer_max_blocks() {
  const int n = static_cast<int>(MOJO_SWITCH("MOJO_HIP_PEER_BLOCKS", 64));
  return n >= 1 ? n : 64;
}

// Bound of every flag wait, in ticks of the 100 MHz constant counter.  `mojo_hip_peer_set_timeout_ms()` (the self-test of
// comm/select.py shortens the bound for its own calls and restores it) overrides MOJO_HIP_PEER_TIMEOUT_MS (default 20 s).
// The value is an ARGUMENT of every exchange kernel: a launch — and a captured graph — keeps the bound it was enqueued with.
static std::atomic<long long> g_peer_timeout_ms{0};
static long long timeout_ticks() {
  long long ms = g_peer_timeout_ms.load(std::memory_order_relaxed);
  if (ms <= 0) ms = MOJO_SWITCH("MOJO_HIP_PEER_TIMEOUT_MS", 20000);
  return (ms > 0 ? ms : 20000) * 100000LL;
}

}  // namespace mojo

using namespace mojo;

// ---- setup-time API: the only entry points of the library that allocate ------------------------------------------
extern "C" int64_t mojo_hip_peer_ctrl_bytes(void) { return static_cast<int64_t>(PEER_CTRL_WORDS) * 4; }
extern "C" int64_t mojo_hip_peer_max_ranks(void) { return PEER_MAX; }
extern "C" int64_t mojo_hip_peer_max_chunks(void) { return PEER_MAX_CHUNKS; }
extern "C" int64_t mojo_hip_peer_handle_bytes(void) { return static_cast<int64_t>(sizeof(hipIpcMemHandle_t)); }

// Sets the bound of the flag waits of every exchange step enqueued from now on (milliseconds; <= 0 = back to
// MOJO_HIP_PEER_TIMEOUT_MS / 20 s) and returns the previous setting (0 = the environment's).
extern "C" int64_t mojo_hip_peer_set_timeout_ms(int64_t ms) {
  return g_peer_timeout_ms.exchange(ms > 0 ? ms : 0, std::memory_order_relaxed);
}

extern "C" int mojo_hip_peer_alloc(void** ptr_out, int64_t bytes, int uncached) {
  MOJO_REQUIRE(ptr_out && bytes > 0, MOJO_EINVAL, "peer_alloc: bad arguments");
  void* p = nullptr;
  hipError_t e = uncached ? hipExtMallocWithFlags(&p, static_cast<size_t>(bytes), hipDeviceMallocUncached)
                          : hipMalloc(&p, static_cast<size_t>(bytes));
  if (e != hipSuccess) {
    (void)hipGetLastError();
    set_error("peer_alloc(%lld B, uncached=%d): %s", (long long)bytes, uncached, hipGetErrorString(e));
    return MOJO_ELAUNCH;
  }
  e = hipMemset(p, 0, static_cast<size_t>(bytes));
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    (void)hipFree(p);
    set_error("peer_alloc: clearing the buffer failed: %s", hipGetErrorString(e));
    return MOJO_ELAUNCH;
  }
  *ptr_out = p;
  return MOJO_OK;
}

extern "C" int mojo_hip_peer_free(void* ptr) {
  if (!ptr) return MOJO_OK;
  hipError_t e = hipFree(ptr);
  MOJO_REQUIRE(e == hipSuccess, MOJO_ELAUNCH, "peer_free: %s", hipGetErrorString(e));
  return MOJO_OK;
}

extern "C" int mojo_hip_peer_export(void* ptr, void* handle_out) {
  MOJO_REQUIRE(ptr && handle_out, MOJO_EINVAL, "peer_export: null pointer");
  hipIpcMemHandle_t h;
  hipError_t e = hipIpcGetMemHandle(&h, ptr);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    set_error("peer_export: hipIpcGetMemHandle: %s (is HSA_ENABLE_IPC_MODE_LEGACY=0 exported?)", hipGetErrorString(e));
    return MOJO_ELAUNCH;
  }
  memcpy(handle_out, &h, sizeof(h));
  return MOJO_OK;
}

extern "C" int mojo_hip_peer_open(const void* handle, void** ptr_out) {
  MOJO_REQUIRE(handle && ptr_out, MOJO_EINVAL, "peer_open: null pointer");
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof(h));
  void* p = nullptr;
  hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    set_error("peer_open: hipIpcOpenMemHandle: %s", hipGetErrorString(e));
    return MOJO_ELAUNCH;
  }
  *ptr_out = p;
  return MOJO_OK;
}

extern "C" int mojo_hip_peer_close(void* ptr) {
  if (!ptr) return MOJO_OK;
  hipError_t e = hipIpcCloseMemHandle(ptr);
  MOJO_REQUIRE(e == hipSuccess, MOJO_ELAUNCH, "peer_close: %s", hipGetErrorString(e));
  return MOJO_OK;
}

// Reads `bytes` of a peer's buffer (an opened mapping, or the own allocation) into host memory with a plain runtime copy:
// the set-up check that an opened mapping really shows the peer's memory.  Synchronises.
extern "C" int mojo_hip_peer_peek(const void* peer_ptr, void* host_out, int64_t bytes) {
  MOJO_REQUIRE(peer_ptr && host_out && bytes >= 0, MOJO_EINVAL, "peer_peek: bad argument");
  hipError_t e = hipMemcpy(host_out, peer_ptr, static_cast<size_t>(bytes), hipMemcpyDeviceToHost);
  MOJO_REQUIRE(e == hipSuccess, MOJO_ELAUNCH, "peer_peek: %s", hipGetErrorString(e));
  return MOJO_OK;
}

// Reads (and optionally clears) the sticky error word of this rank's control area.  Synchronises the device.
extern "C" int mojo_hip_peer_error(void* local_flags, int clear, int32_t* error_out) {
  MOJO_REQUIRE(local_flags && error_out, MOJO_EINVAL, "peer_error: null pointer");
  uint32_t v = 0;
  hipError_t e = hipMemcpy(&v, static_cast<uint32_t*>(local_flags) + PEER_ERR_WORD, 4, hipMemcpyDeviceToHost);
  MOJO_REQUIRE(e == hipSuccess, MOJO_ELAUNCH, "peer_error: %s", hipGetErrorString(e));
  if (clear && v) {
    const uint32_t z = 0;
    e = hipMemcpy(static_cast<uint32_t*>(local_flags) + PEER_ERR_WORD, &z, 4, hipMemcpyHostToDevice);
    MOJO_REQUIRE(e == hipSuccess, MOJO_ELAUNCH, "peer_error: %s", hipGetErrorString(e));
  }
  *error_out = static_cast<int32_t>(v);
  return MOJO_OK;
}

// ---- exchange steps (enqueue only: no allocation, no host sync) -----------------------------------------------------
// Every step takes the call's epoch; 0 = captured mode (see the header of this file).
extern "C" int mojo_hip_peer_begin(void* const* peer_data, void* const* peer_flags, int64_t world, int64_t rank, mojo_stream_t stream) {
  MOJO_REQUIRE(world >= 1 && world <= PEER_MAX && rank >= 0 && rank < world, MOJO_EINVAL, "peer_begin: bad arguments");
  PeerPtrs pp;
  MOJO_REQUIRE(fill_ptrs(pp, peer_data, peer_flags, static_cast<int>(world)) == MOJO_OK, MOJO_EINVAL, "peer_begin: null peer pointer");
  hipLaunchKernelGGL(peer_begin_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), pp, static_cast<int>(world),
                     static_cast<int>(rank), timeout_ticks());
  MOJO_CHECK_LAUNCH("peer_begin");
  return MOJO_OK;
}

extern "C" int mojo_hip_peer_signal(void* const* peer_data, void* const* peer_flags, int64_t world, int64_t rank,
                                    int kind, int64_t chunk, uint32_t epoch, mojo_stream_t stream) {
  MOJO_REQUIRE(world >= 1 && world <= PEER_MAX && rank >= 0 && rank < world && chunk >= 0 && chunk < PEER_MAX_CHUNKS &&
                   kind >= 0 && kind < PEER_FLAG_KINDS,
               MOJO_EINVAL, "peer_signal: bad arguments");
  PeerPtrs pp;
  MOJO_REQUIRE(fill_ptrs(pp, peer_data, peer_flags, static_cast<int>(world)) == MOJO_OK, MOJO_EINVAL, "peer_signal: null peer pointer");
  hipLaunchKernelGGL(peer_signal_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), pp, static_cast<int>(world),
                     static_cast<int>(rank), kind, static_cast<int>(chunk), epoch);
  MOJO_CHECK_LAUNCH("peer_signal");
  return MOJO_OK;
}

template <typename T>
static int launch_reduce(const PeerPtrs& pp, int ws, int rank, int chunk, uint32_t epoch, int64_t src_off, int64_t rows,
                         int64_t n, void* dst, int64_t ld_dst, int write_back, hipStream_t s) {
  const int64_t vecs = rows * (n / (16 / static_cast<int64_t>(sizeof(T))));
  // Few workgroups: they share the chip with the next chunk's GEMM, whose workgroups need a whole CU each (all of its
  // registers and 128 KiB of LDS) — a waiting pull workgroup on every CU would keep that GEMM off the chip until the peers
  // have signalled (and, with two ranks time-sharing ONE GPU as in the tests, for ever: the peer's GEMM is what it waits for).
  // 64 workgroups keep > 1 MiB of 16-byte reads in flight, far more than seven xGMI links need.
  int64_t blocks = ceil_div(vecs, 256 * 2);
  if (blocks > peer_max_blocks()) blocks = peer_max_blocks();
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(peer_reduce_kernel<T>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, pp, ws, rank, chunk, epoch,
                     static_cast<long long>(src_off), static_cast<long long>(rows), static_cast<int>(n), static_cast<T*>(dst),
                     static_cast<long long>(ld_dst), write_back, timeout_ticks());
  return MOJO_OK;
}

extern "C" int mojo_hip_peer_reduce(void* const* peer_data, void* const* peer_flags, int64_t world, int64_t rank,
                                    int64_t chunk, uint32_t epoch, int64_t src_offset_bytes, int64_t rows, int64_t n,
                                    void* dst, int64_t ld_dst, int write_back, int dtype, mojo_stream_t stream) {
  MOJO_REQUIRE(world >= 1 && world <= PEER_MAX && rank >= 0 && rank < world && chunk >= 0 && chunk < PEER_MAX_CHUNKS,
               MOJO_EINVAL, "peer_reduce: bad arguments");
  // An empty share (a chunk with fewer rows than ranks: decode-sized M under TP 8) moves no data, but with write_back the
  // peers' gather step still waits for this rank's "share ready" flag: the launch must run its flag-raising tail.
  if (rows == 0 && !write_back) return MOJO_OK;
  MOJO_REQUIRE((dst || rows == 0) && rows >= 0 && n > 0 && n < (1LL << 31) && src_offset_bytes >= 0, MOJO_EINVAL,
               "peer_reduce: bad shape");
  MOJO_REQUIRE(dtype == MOJO_BF16 || dtype == MOJO_F16 || dtype == MOJO_F32, MOJO_EUNSUPPORTED, "peer_reduce: dtype %d", dtype);
  const int64_t ve = dtype == MOJO_F32 ? 4 : 8;
  MOJO_REQUIRE(n % ve == 0 && ld_dst % ve == 0 && (rows == 0 || aligned_to(dst, 16)) && src_offset_bytes % 16 == 0,
               MOJO_EUNSUPPORTED, "peer_reduce: rows must be whole 16-byte vectors");
  PeerPtrs pp;
  MOJO_REQUIRE(fill_ptrs(pp, peer_data, peer_flags, static_cast<int>(world)) == MOJO_OK, MOJO_EINVAL, "peer_reduce: null peer pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int ws = static_cast<int>(world), rk = static_cast<int>(rank), ck = static_cast<int>(chunk);
  switch (dtype) {
    case MOJO_BF16: launch_reduce<bf16_t>(pp, ws, rk, ck, epoch, src_offset_bytes, rows, n, dst, ld_dst, write_back, s); break;
    case MOJO_F16: launch_reduce<f16_t>(pp, ws, rk, ck, epoch, src_offset_bytes, rows, n, dst, ld_dst, write_back, s); break;
    default: launch_reduce<float>(pp, ws, rk, ck, epoch, src_offset_bytes, rows, n, dst, ld_dst, write_back, s); break;
  }
  MOJO_CHECK_LAUNCH("peer_reduce");
  return MOJO_OK;
}

extern "C" int mojo_hip_peer_gather(void* const* peer_data, void* const* peer_flags, int64_t world, int64_t rank,
                                    int64_t chunk, uint32_t epoch, int64_t chunk_offset_bytes, int64_t rows, int64_t n,
                                    void* dst, int64_t ld_dst, int dtype, mojo_stream_t stream) {
  MOJO_REQUIRE(world >= 1 && world <= PEER_MAX && rank >= 0 && rank < world && chunk >= 0 && chunk < PEER_MAX_CHUNKS,
               MOJO_EINVAL, "peer_gather: bad arguments");
  if (rows == 0 || world == 1) return MOJO_OK;
  MOJO_REQUIRE(dst && rows > 0 && n > 0 && n < (1LL << 31) && chunk_offset_bytes >= 0, MOJO_EINVAL, "peer_gather: bad shape");
  MOJO_REQUIRE(dtype == MOJO_BF16 || dtype == MOJO_F16 || dtype == MOJO_F32, MOJO_EUNSUPPORTED, "peer_gather: dtype %d", dtype);
  const int64_t ve = dtype == MOJO_F32 ? 4 : 8;
  MOJO_REQUIRE(n % ve == 0 && ld_dst % ve == 0 && aligned_to(dst, 16) && chunk_offset_bytes % 16 == 0, MOJO_EUNSUPPORTED,
               "peer_gather: rows must be whole 16-byte vectors");
  PeerPtrs pp;
  MOJO_REQUIRE(fill_ptrs(pp, peer_data, peer_flags, static_cast<int>(world)) == MOJO_OK, MOJO_EINVAL, "peer_gather: null peer pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int ws = static_cast<int>(world), rk = static_cast<int>(rank), ck = static_cast<int>(chunk);
  const int64_t share_vecs = ceil_div(rows, world) * (n / ve);
  int64_t bx = ceil_div(share_vecs, 256 * 2);
  if (bx > peer_max_blocks() / (ws - 1)) bx = peer_max_blocks() / (ws - 1);     // same budget as the pull-and-add launch
  if (bx < 1) bx = 1;
  const dim3 grid(static_cast<unsigned>(bx), static_cast<unsigned>(ws - 1));
  const long long tt = timeout_ticks();
#define LAUNCH(T) hipLaunchKernelGGL(peer_gather_kernel<T>, grid, dim3(256), 0, s, pp, ws, rk, ck, epoch,                 \
                                     static_cast<long long>(chunk_offset_bytes), static_cast<long long>(rows), static_cast<int>(n), \
                                     static_cast<T*>(dst), static_cast<long long>(ld_dst), tt)
  switch (dtype) {
    case MOJO_BF16: LAUNCH(bf16_t); break;
    case MOJO_F16: LAUNCH(f16_t); break;
    default: LAUNCH(float); break;
  }
#undef LAUNCH
  MOJO_CHECK_LAUNCH("peer_gather");
  return MOJO_OK;
}

extern "C" int mojo_hip_peer_pull(void* const* peer_data, void* const* peer_flags, int64_t world, int64_t rank, int kind,
                                  int64_t flag_chunk, uint32_t epoch, int64_t src_offset_bytes, int64_t bytes, void* dst,
                                  int64_t dst_stride_bytes, int include_self, mojo_stream_t stream) {
  MOJO_REQUIRE(world >= 1 && world <= PEER_MAX && rank >= 0 && rank < world && flag_chunk >= 0 && flag_chunk < PEER_MAX_CHUNKS &&
                   kind >= 0 && kind < PEER_FLAG_KINDS,
               MOJO_EINVAL, "peer_pull: bad arguments");
  const int first = include_self ? 0 : 1;
  const int n_src = static_cast<int>(world) - first;
  if (bytes == 0 || n_src <= 0) return MOJO_OK;
  MOJO_REQUIRE(dst && bytes > 0 && src_offset_bytes >= 0 && dst_stride_bytes >= bytes, MOJO_EINVAL, "peer_pull: bad shape");
  MOJO_REQUIRE(bytes % 16 == 0 && src_offset_bytes % 16 == 0 && dst_stride_bytes % 16 == 0 && aligned_to(dst, 16), MOJO_EUNSUPPORTED,
               "peer_pull: blocks must be whole 16-byte vectors");
  PeerPtrs pp;
  MOJO_REQUIRE(fill_ptrs(pp, peer_data, peer_flags, static_cast<int>(world)) == MOJO_OK, MOJO_EINVAL, "peer_pull: null peer pointer");
  int64_t bx = ceil_div(bytes / 16, 256 * 2);
  const int64_t cap = peer_max_blocks() / n_src > 0 ? peer_max_blocks() / n_src : 1;
  if (bx > cap) bx = cap;
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(peer_pull_kernel, dim3(static_cast<unsigned>(bx), static_cast<unsigned>(n_src)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), pp, static_cast<int>(world), static_cast<int>(rank), kind,
                     static_cast<int>(flag_chunk), epoch, static_cast<long long>(src_offset_bytes), static_cast<long long>(bytes),
                     static_cast<char*>(dst), static_cast<long long>(dst_stride_bytes), first, timeout_ticks());
  MOJO_CHECK_LAUNCH("peer_pull");
  return MOJO_OK;
}
