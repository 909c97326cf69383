"""Symmetric peer buffers over HIP IPC and the direct (ring-free) exchange built on them.

Role in the reference: `MojoSymmetricMemoryManager.allocate_peer_mem` (`runtime/comm_context.py:107-153`): one buffer per
rank, the same size everywhere, every rank holding a pointer to every other rank's copy, cached per (group, size).  Here a
buffer is a `hipExtMallocWithFlags(hipDeviceMallocUncached)` allocation (plain `hipMalloc` if that cannot be exported)
shared with `hipIpcGetMemHandle` / `hipIpcOpenMemHandle`; the 64-byte handles travel through the process group itself
(`all_gather_object`), so any backend — RCCL on a real node, gloo in the single-GPU tests — can set it up.

`PeerExchange` then drives csrc/peer_comm.hip: per row chunk, GEMM -> signal on the caller's stream; pull-and-add
(+ pull-gather for the all-reduce) on a side stream, so the xGMI traffic of chunk c overlaps the GEMM of chunk c + 1.
Nothing here synchronises with the host after set-up.
"""
import ctypes
import os
import threading
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from .. import switches
from ..backends.hip import lib as L

_LOCK = threading.RLock()
_CACHE: Dict[Tuple[object, int], "PeerExchange"] = {}
# Exchanges that were replaced by a larger one: kept mapped — and checked — until release_all(), never freed in between.
# Two reasons, both found the hard way: (1) a captured launch has its twin's raw peer pointers baked in — freeing
# the twin would make the next replay read and write freed or re-mapped IPC memory on every rank; (2) a buffer freed and
# re-allocated at the SAME virtual address exports the same IPC handle bytes, and a peer's `hipIpcOpenMemHandle` of that
# handle then showed the OLD memory (caught by `_verify_mapping` in the 4-rank test of round 5: "opened mappings do not
# show the peers' memory").  Capacity at least doubles on every rebuild, so the retired buffers add up to less than the live one.
_RETIRED: List["PeerExchange"] = []


class _DeviceBytes:
    """`__cuda_array_interface__` view of raw device memory, so torch can alias it without owning it."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def _ptr_array(ptrs: List[int]):
    arr = (ctypes.c_void_p * len(ptrs))(*ptrs)
    return arr


class PeerExchange:
    """Symmetric data + control areas of one process group, and the per-call bookkeeping (epoch, parity).

    ``captured=True`` builds the twin that HIP-graph capture uses: ONE data area (no parity halves), the epoch in the control
    area on the device (csrc/peer_comm.hip, "captured mode"): its calls bake no host-side counter into a launch."""

    def __init__(self, group, capacity_bytes: int, captured: bool = False):
        self.captured = bool(captured)
        self.halves = 1 if captured else 2
        self.handed_out_under_capture = False               # (twins) a graph may hold this exchange's pointers
        self.twin: Optional["PeerExchange"] = None
        self.group = group
        self.ws = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        lib = L.load()
        if self.ws > lib.mojo_hip_peer_max_ranks():
            raise NotImplementedError(f"direct exchange supports up to {lib.mojo_hip_peer_max_ranks()} ranks per node")
        self.capacity = (int(capacity_bytes) + 4095) // 4096 * 4096            # bytes of ONE parity half
        self.ctrl_bytes = int(lib.mojo_hip_peer_ctrl_bytes())
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.max_chunks = int(lib.mojo_hip_peer_max_chunks())
        total = self.halves * self.capacity + 4096 + self.ctrl_bytes
        self._local = ctypes.c_void_p()
        self._opened: List[ctypes.c_void_p] = []
        uncached = True                                     # (falls back to plain hipMalloc when the uncached form cannot be exported)
        handle = (ctypes.c_char * int(lib.mojo_hip_peer_handle_bytes()))()
        setup_error = None
        try:
            for attempt in ((1, 0) if uncached else (0,)):
                L.check(lib.mojo_hip_peer_alloc(ctypes.byref(self._local), total, attempt), "peer buffer allocation")
                rc = lib.mojo_hip_peer_export(self._local, handle)
                if rc == 0:
                    self.uncached = bool(attempt)
                    break
                lib.mojo_hip_peer_free(self._local)
                self._local = ctypes.c_void_p()
                if attempt == 0:
                    L.check(rc, "peer buffer export")
        except Exception as e:              # reported to every rank below: nobody may be left waiting in a collective
            setup_error = repr(e)
        # exchange (handle, pid): a rank must not IPC-open its own allocation
        mine = (bytes(handle), os.getpid(), total, setup_error)
        everyone: List[Optional[tuple]] = [None] * self.ws
        dist.all_gather_object(everyone, mine, group=group)
        if any(e[3] for e in everyone):
            self.close()
            raise RuntimeError(f"peer buffers: allocation / export failed (per rank: {[e[3] for e in everyone]})")
        if any(e[2] != total for e in everyone):
            raise RuntimeError("peer buffers: ranks disagree on the buffer size (every rank must make the same calls)")
        self._bases: List[int] = []
        failure = None
        for r, (h, _pid, _sz, _err) in enumerate(everyone):
            if r == self.rank:
                self._bases.append(self._local.value)
                continue
            p = ctypes.c_void_p()
            try:
                L.check(lib.mojo_hip_peer_open(ctypes.create_string_buffer(h, len(h)), ctypes.byref(p)), f"opening rank {r}'s peer buffer")
            except Exception as e:          # keep going: the ranks must reach the next collective together
                failure = failure or repr(e)
                self._bases.append(0)
                continue
            self._opened.append(p)
            self._bases.append(p.value)
        failures: List[Optional[str]] = [None] * self.ws
        dist.all_gather_object(failures, failure, group=group)
        if any(failures):
            self.close()
            raise RuntimeError(f"peer buffers: a rank could not open a peer's buffer (per rank: {failures})")
        self._flag_off = self.halves * self.capacity + 4096
        self._data = _ptr_array(self._bases)
        self._flags = _ptr_array([b + self._flag_off for b in self._bases])
        self._alias = torch.as_tensor(_DeviceBytes(self._local.value, self.halves * self.capacity), device=self.device)
        self.epoch = 0
        self.side = torch.cuda.Stream(device=self.device)
        dist.barrier(group=group)               # every rank has opened every buffer (and cleared its own) before first use
        self._verify_mapping()

    def _verify_mapping(self) -> None:
        """Every rank stamps the head of its data area and reads every peer's stamp back through the opened pointer with a
        plain device copy: a mapping that does not reach the peer's memory fails here, as an exception on every rank,
        and not as a poisoned result (or a fault) inside the first exchange kernel."""
        words = 64
        stamp = torch.arange(words, dtype=torch.int32, device=self.device) * 16 + self.rank
        self._alias[: words * 4].view(torch.int32).copy_(stamp)
        torch.cuda.synchronize(self.device)
        dist.barrier(group=self.group)
        bad = []
        lib = L.load()
        host = (ctypes.c_int32 * words)()
        for r, base in enumerate(self._bases):              # a runtime copy through the opened pointer, not a kernel: a
            rc = lib.mojo_hip_peer_peek(ctypes.c_void_p(base), host, words * 4)   # bad mapping is an error code here
            if rc != 0 or any(host[i] != i * 16 + r for i in range(words)):
                bad.append(r)
        torch.cuda.synchronize(self.device)
        verdicts: List[Optional[list]] = [None] * self.ws
        dist.all_gather_object(verdicts, bad, group=self.group)   # also: nobody clears a stamp a peer is still reading
        self._alias[: words * 4].zero_()
        torch.cuda.synchronize(self.device)
        dist.barrier(group=self.group)
        if any(verdicts):
            raise RuntimeError(f"peer buffers: opened mappings do not show the peers' memory (per rank: {verdicts})")

    # ---- per-call state -----------------------------------------------------------------------------------------
    def begin_call(self) -> Tuple[int, int]:
        """(epoch, byte offset of this call's half of the data area).  Every rank makes the same sequence of calls.
        Captured twin: enqueues the begin step (wait for the peers' "done reading" flags of the previous call, advance the
        device-resident epoch) on the current stream and returns (0, 0): epoch 0 = "read it from the control area"."""
        if self.captured:
            L.check(L.load().mojo_hip_peer_begin(self._data, self._flags, self.ws, self.rank,
                                                 ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)), "peer begin")
            return 0, 0
        self.epoch += 1
        return self.epoch & 0xFFFFFFFF, (self.epoch & 1) * self.capacity

    def end_call(self) -> None:
        """Captured twin: tell every peer that this rank has finished reading their data of this call (flag kind 2), on the
        current stream, which by now waits for the side stream's pulls.  Eager exchange: nothing (parity halves)."""
        if self.captured:
            self.signal(2, 0, 0, ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))

    def local_view(self, byte_offset: int, rows: int, cols: int, dtype: torch.dtype) -> torch.Tensor:
        n = rows * cols * torch.empty((), dtype=dtype).element_size()
        return self._alias[byte_offset: byte_offset + n].view(dtype).view(rows, cols)

    def signal(self, kind: int, chunk: int, epoch: int, stream) -> None:
        L.check(L.load().mojo_hip_peer_signal(self._data, self._flags, self.ws, self.rank, kind, chunk, epoch, stream),
                "peer signal")

    def reduce(self, chunk, epoch, src_off, rows, n, dst: torch.Tensor, write_back: bool, stream) -> None:
        L.check(L.load().mojo_hip_peer_reduce(self._data, self._flags, self.ws, self.rank, chunk, epoch, src_off, rows, n,
                                              L.ptr(dst), dst.stride(0), 1 if write_back else 0, L.dtype_code(dst.dtype),
                                              stream), "peer reduce")

    def gather(self, chunk, epoch, chunk_off, rows, n, dst: torch.Tensor, stream) -> None:
        L.check(L.load().mojo_hip_peer_gather(self._data, self._flags, self.ws, self.rank, chunk, epoch, chunk_off, rows, n,
                                              L.ptr(dst), dst.stride(0), L.dtype_code(dst.dtype), stream), "peer gather")

    def pull(self, kind, flag_chunk, epoch, src_off, nbytes, dst: torch.Tensor, dst_stride_bytes, include_self, stream) -> None:
        L.check(L.load().mojo_hip_peer_pull(self._data, self._flags, self.ws, self.rank, kind, flag_chunk, epoch, src_off,
                                            nbytes, L.ptr(dst), dst_stride_bytes, 1 if include_self else 0, stream), "peer pull")

    def check(self, clear: bool = True) -> None:
        """Raise if a wait of an earlier call — eager, or a replay of a graph captured over the twin — timed out (its outputs
        were poisoned with NaN).  Synchronises."""
        if self.twin is not None:
            self.twin.check(clear)
        err = ctypes.c_int32(0)
        L.check(L.load().mojo_hip_peer_error(ctypes.c_void_p(self._local.value + self._flag_off), 1 if clear else 0,
                                             ctypes.byref(err)), "peer error word")
        if err.value:
            raise RuntimeError(f"direct peer exchange ({'captured twin' if self.captured else 'eager'}): a wait for a peer's flag "
                               "timed out (MOJO_HIP_PEER_TIMEOUT_MS); the affected outputs were filled with NaN")

    def close(self) -> None:
        if self.twin is not None:
            self.twin.close()
            self.twin = None
        lib = L.load()
        for p in self._opened:
            lib.mojo_hip_peer_close(p)
        self._opened = []
        if self._local:
            lib.mojo_hip_peer_free(self._local)
            self._local = ctypes.c_void_p()


def captured_ready(group, need_bytes: int) -> bool:
    """True when a captured twin of at least ``need_bytes`` exists for the group (it is built together with the eager
    exchange, i.e. by any eager call of a direct operator at this size or larger: warm the step up before capturing it)."""
    key = getattr(group, "group_name", None) or id(group)
    with _LOCK:
        return any(k == key and cap >= need_bytes and ex.twin is not None for (k, cap), ex in _CACHE.items())


def check_all(group=None, clear: bool = True) -> None:
    """Raise if any exchange of ``group`` (every group when None) recorded a timed-out wait: the eager exchanges, their
    captured twins and the retired exchanges old graphs may still replay over.  Call it after a graph replay (or any step) whose
    outputs matter: a timed-out wait poisons its outputs with NaN and makes every later wait of that exchange give up at
    once, so one unnoticed timeout would otherwise turn every later replay into silent NaN.  Synchronises."""
    with _LOCK:
        exs = [ex for ex in list(_CACHE.values()) + list(_RETIRED) if group is None or ex.group is group]
    first = None
    for ex in exs:
        try:
            ex.check(clear)
        except RuntimeError as e:                            # read (and clear) them all before reporting the first
            first = first or e
    if first is not None:
        raise first


def get_exchange(group, need_bytes: int) -> PeerExchange:
    """The group's exchange with at least ``need_bytes`` per parity half; (re)built collectively when it must grow, so every
    rank has to ask with the same sizes in the same order (they do: the ops are SPMD).  Under HIP-graph capture the
    captured twin is returned; it cannot be built there (allocation, collectives): NotImplementedError when it is missing."""
    # (the group's name identifies it for its lifetime; `id()` of a collected group object could be reused)
    key = getattr(group, "group_name", None) or id(group)
    capturing = torch.cuda.is_current_stream_capturing()
    with _LOCK:
        for (k, cap), ex in list(_CACHE.items()):
            if k == key and cap >= need_bytes:
                if not capturing:
                    return ex
                if ex.twin is not None:
                    ex.twin.handed_out_under_capture = True
                    return ex.twin
        if capturing:
            raise NotImplementedError("direct peer exchange under graph capture: no captured buffer of this size yet — run the "
                                      "operator once eagerly at this size (or larger) before capturing")
        largest = 0
        for (k, cap) in [kc for kc in _CACHE if kc[0] == key]:
            torch.cuda.synchronize()
            dist.barrier(group=group)           # (every rank retires the same exchanges in the same order)
            _RETIRED.append(_CACHE.pop((k, cap)))   # retired, NOT freed: see _RETIRED
            largest = max(largest, cap)
        cap = max(int(need_bytes), switches.get_int("MOJO_HIP_PEER_MIN_BYTES", 64 << 20), 2 * largest)
        ex = PeerExchange(group, cap)
        ex.twin = PeerExchange(group, cap, captured=True)           # the graph-capturable twin, built while building is allowed
        _CACHE[(key, ex.capacity)] = ex
        return ex


def release_all() -> None:
    """Free every exchange, INCLUDING retired ones and twins that captured graphs may hold: destroy those graphs first."""
    with _LOCK:
        for ex in list(_CACHE.values()) + list(_RETIRED):
            ex.close()
        _CACHE.clear()
        _RETIRED.clear()


# ---------------------------------------------------------------------------------------------------------------
# the two operators whose exchange is a reduction
# ---------------------------------------------------------------------------------------------------------------
def _side_ptr(ex: PeerExchange):
    return ctypes.c_void_p(ex.side.cuda_stream)


def direct_supported(x: torch.Tensor, n: int, rows: int) -> bool:
    """Whole 16-byte vectors per output row, 16-bit or fp32 data, at least one row."""
    es = x.element_size()
    return x.dtype in (torch.bfloat16, torch.float16, torch.float32) and (n * es) % 16 == 0 and rows > 0


def gemm_all_reduce_direct(engine, x2: torch.Tensor, weight, bias, trans_weight: bool, group) -> torch.Tensor:
    """allreduce_sum(x2 @ W (+ bias)) with the direct exchange.  Per row chunk c (rows [lo, hi)):
         main stream : GEMM -> partial in this rank's peer buffer; signal "partial (me, c) ready" to every peer
         side stream : pull-and-add this rank's 1/ws share of the chunk from every rank (fp32 sum in rank order, one
                       rounding), store it to `out` and back over the own partial; last workgroup signals "share ready";
                       pull the other ranks' shares into `out`.
    The golden sums storage-type values (`fc.all_reduce` of the rounded products, compute_with_comm.py:106-110); summing
    the same rounded partials in fp32 and rounding once is at least as accurate and equal for ws = 2."""
    from .pipelines import plan_row_chunks

    n = engine.out_features(weight, trans_weight)
    m = x2.shape[0]
    es = x2.element_size()
    ex = get_exchange(group, m * n * es)
    epoch, base = ex.begin_call()
    out = torch.empty(m, n, dtype=x2.dtype, device=x2.device)
    main = torch.cuda.current_stream(x2.device)
    ws, rank = ex.ws, ex.rank
    chunks = plan_row_chunks(m, n, ws, 1, es)
    assert len(chunks) <= ex.max_chunks
    for c, (lo, hi) in enumerate(chunks):
        off = base + lo * n * es
        rows = hi - lo
        engine(x2[lo:hi], weight, bias, trans_weight, out=ex.local_view(off, rows, n, x2.dtype))
        ex.signal(0, c, epoch, L.stream_of(x2))
        ex.side.wait_stream(main)
        r0, r1 = rows * rank // ws, rows * (rank + 1) // ws
        ex.reduce(c, epoch, off + r0 * n * es, r1 - r0, n, out[lo + r0: lo + r1], True, _side_ptr(ex))
        ex.gather(c, epoch, off, rows, n, out[lo:hi], _side_ptr(ex))
    main.wait_stream(ex.side)
    ex.end_call()
    return out


def gemm_reduce_scatter_direct(engine, x2: torch.Tensor, weight, bias, trans_weight: bool, group) -> torch.Tensor:
    """Rows [rank * M/ws, (rank + 1) * M/ws) of sum_ranks(x2 @ W (+ bias)): per chunk the GEMM's A-row map produces
    "sub-chunk c of every rank's row block" contiguously ([dest rank][rc rows]) in the peer buffer; this rank pulls and adds
    block `rank` of every rank's buffer."""
    from .pipelines import plan_row_chunks

    n = engine.out_features(weight, trans_weight)
    m = x2.shape[0]
    es = x2.element_size()
    ex = get_exchange(group, m * n * es)
    ws, rank = ex.ws, ex.rank
    assert m % ws == 0
    ml = m // ws
    epoch, base = ex.begin_call()
    out = torch.empty(ml, n, dtype=x2.dtype, device=x2.device)
    main = torch.cuda.current_stream(x2.device)
    chunks = plan_row_chunks(ml, n, ws, ws, es)
    assert len(chunks) <= ex.max_chunks
    for c, (lo, hi) in enumerate(chunks):
        rc = hi - lo
        off = base + ws * lo * n * es
        engine(x2, weight, bias, trans_weight, out=ex.local_view(off, ws * rc, n, x2.dtype), rows=ws * rc, a_map=(rc, ml, lo))
        ex.signal(0, c, epoch, L.stream_of(x2))
        ex.side.wait_stream(main)
        ex.reduce(c, epoch, off + rank * rc * n * es, rc, n, out[lo:hi], False, _side_ptr(ex))
    main.wait_stream(ex.side)
    ex.end_call()
    return out


def all_gather_gemm_direct(engine, x2: torch.Tensor, weight, bias, trans_weight: bool, group) -> torch.Tensor:
    """allgather(x2, rows) @ W (+ bias) with the gather pulled straight from the peers' buffers: this rank's shard is copied
    into its peer buffer and flagged once; per row chunk c a side-stream launch pulls rows [lo, hi) of EVERY rank's shard
    (ws - 1 links in parallel) into a contiguous [ws * rc, K] block, and the GEMM on that block writes its rows to their final
    place through the C-row map — the pull of chunk c + 1 overlaps the GEMM of chunk c."""
    from .pipelines import plan_row_chunks

    n = engine.out_features(weight, trans_weight)
    ml, k = x2.shape
    es = x2.element_size()
    ex = get_exchange(group, ml * k * es)
    ws = ex.ws
    epoch, base = ex.begin_call()
    main = torch.cuda.current_stream(x2.device)
    ex.local_view(base, ml, k, x2.dtype).copy_(x2)
    ex.signal(0, 0, epoch, L.stream_of(x2))
    out = torch.empty(ws * ml, n, dtype=x2.dtype, device=x2.device)
    chunks = plan_row_chunks(ml, n, ws, ws, es, link_cols=k)
    ex.side.wait_stream(main)
    stages = []
    for lo, hi in chunks:
        rc = hi - lo
        buf = torch.empty(ws * rc, k, dtype=x2.dtype, device=x2.device)
        ex.pull(0, 0, epoch, base + lo * k * es, rc * k * es, buf, rc * k * es, True, _side_ptr(ex))
        ev = torch.cuda.Event()
        ev.record(ex.side)
        stages.append((buf, ev))
    for (lo, hi), (buf, ev) in zip(chunks, stages):
        main.wait_event(ev)
        engine(buf, weight, bias, trans_weight, out=out, rows=buf.shape[0], c_map=(hi - lo, ml, lo))
    ex.end_call()                                           # (main has waited for the last pull's event: every pull is done)
    return out
