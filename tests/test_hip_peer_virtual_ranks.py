"""The direct peer exchange's kernels with 4 and 8 ranks inside ONE process (round 5).

The GPU pool gives the builder one MI355X and allows at most 6 of a user's processes on it, so 8 ranks cannot run as 8
processes here (4 do: tests/test_hip_comm_ranks.py).  What 8 real ranks exercise beyond 4 — the flag indexing
`[kind][source rank][chunk]` with 7 peers, the row shares `rows * p / ws`, epochs and parity halves over several calls, the
captured mode's "done reading" flags from 7 peers — lives in csrc/peer_comm.hip and needs no second process: every "rank" is a
buffer of this process (`mojo_hip_peer_alloc`, the very allocation the operators share over HIP-IPC), a stream, and the same
C-ABI calls `mojo_opset_amd/comm/peer.py` makes, in the same order.  Launches are enqueued in dependency order (all signals of
a step before the first wait on them), so the grid can always drain; every wait is bounded at 3 s anyway.  Data are small
integers: sums over 8 ranks are exact in fp32 and in bf16, so every comparison is `torch.equal`.

Golden semantics: `core/operators/compute_with_comm.py:57-116` (all-reduce), `:264-340` (reduce-scatter), `:119-184` (all-gather).
"""
import ctypes

import pytest
import torch

from hip_utils import DEV
from mojo_opset_amd.backends.hip import lib as L
from mojo_opset_amd.comm.peer import _DeviceBytes, _ptr_array

pytestmark = pytest.mark.gpu


_POOL = {}                      # bytes -> the 8 buffers of this process, allocated ONCE and never freed (see VirtualRanks)


class VirtualRanks:
    """`ws` symmetric peer buffers of ONE process + the per-call bookkeeping of `PeerExchange` (epoch, parity halves).

    The buffers come from a per-process pool that is never freed: on this platform an uncached allocation that is freed and
    re-allocated at the same virtual address can show its OLD contents to some compute units (this test failed intermittently
    — stale operands, never a timeout — while it allocated and freed eight buffers per case, and never on plain `hipMalloc`
    memory, which the runtime pools; a bare write-then-read probe on one uncached buffer, scripts/probes/uc_visibility_probe.hip,
    is clean).  The product follows the same rule: `comm/peer.py` retires an exchange it outgrows instead of freeing it."""

    def __init__(self, ws, capacity, captured=False):
        self.lib = L.load()
        self.ws, self.cap, self.captured = ws, capacity, captured
        self.halves = 1 if captured else 2
        ctrl = int(self.lib.mojo_hip_peer_ctrl_bytes())
        total = 2 * capacity + 4096 + ctrl                      # (sized for two halves; the captured layout uses the front part)
        if total not in _POOL:
            bufs = []
            for _ in range(int(self.lib.mojo_hip_peer_max_ranks()) // 2):      # 8 of the 16 ranks the kernels support
                p = ctypes.c_void_p()
                rc = self.lib.mojo_hip_peer_alloc(ctypes.byref(p), total, 1)
                if rc != 0:
                    L.check(self.lib.mojo_hip_peer_alloc(ctypes.byref(p), total, 0), "peer_alloc")
                bufs.append(p)
            _POOL[total] = bufs
        self.bufs = _POOL[total][:ws]
        whole = [torch.as_tensor(_DeviceBytes(b.value, total), device=torch.device(DEV, 0)) for b in self.bufs]
        for w in whole:
            w.zero_()                                           # data, flags, epoch word, counters, error word: a fresh exchange
        torch.cuda.synchronize()
        self.flag_off = self.halves * capacity + 4096
        self.data = _ptr_array([b.value for b in self.bufs])
        self.flags = _ptr_array([b.value + self.flag_off for b in self.bufs])
        self.alias = [w[: self.halves * capacity] for w in whole]
        self.streams = [torch.cuda.Stream() for _ in range(ws)]
        self.epoch = 0
        self.old_timeout = self.lib.mojo_hip_peer_set_timeout_ms(3000)

    def close(self):
        torch.cuda.synchronize()
        self.lib.mojo_hip_peer_set_timeout_ms(self.old_timeout)

    def sp(self, r):
        return ctypes.c_void_p(self.streams[r].cuda_stream)

    def view(self, r, off, rows, n, dtype):
        nb = rows * n * torch.empty((), dtype=dtype).element_size()
        return self.alias[r][off: off + nb].view(dtype).view(rows, n)

    def begin(self):
        """(epoch argument, byte offset of the call's data half) — as PeerExchange.begin_call on every rank.  The ranks'
        streams first wait for the caller's stream: the operands and the NaN-filled outputs were produced there (torch's
        side streams are non-blocking: nothing orders them behind the default stream by itself)."""
        here = torch.cuda.current_stream()
        for st in self.streams:
            st.wait_stream(here)
        if self.captured:
            for r in range(self.ws):
                L.check(self.lib.mojo_hip_peer_begin(self.data, self.flags, self.ws, r, self.sp(r)), "begin")
            return 0, 0
        self.epoch += 1
        return self.epoch, (self.epoch & 1) * self.cap

    def end(self):
        if self.captured:
            for r in range(self.ws):
                L.check(self.lib.mojo_hip_peer_signal(self.data, self.flags, self.ws, r, 2, 0, 0, self.sp(r)), "signal kind 2")

    def errors(self):
        torch.cuda.synchronize()
        out = []
        for r in range(self.ws):
            e = ctypes.c_int32(0)
            L.check(self.lib.mojo_hip_peer_error(ctypes.c_void_p(self.bufs[r].value + self.flag_off), 1, ctypes.byref(e)), "error word")
            out.append(e.value)
        return out


def _pattern(rank, rows, n, dtype, salt=0):
    i = torch.arange(rows, device=DEV, dtype=torch.int64).unsqueeze(1)
    j = torch.arange(n, device=DEV, dtype=torch.int64).unsqueeze(0)
    return (((i * 131 + j * 7 + rank * 29 + salt * 13) % 17) - 8).to(dtype)


def _chunks(m, n_chunks):
    step = -(-m // n_chunks)
    return [(lo, min(m, lo + step)) for lo in range(0, m, step)]


def all_reduce(v, parts, n_chunks):
    """The step sequence of comm/peer.py gemm_all_reduce_direct for every virtual rank; returns the per-rank outputs."""
    ws = v.ws
    m, n = parts[0].shape
    dtype, es = parts[0].dtype, parts[0].element_size()
    code = L.dtype_code(dtype)
    outs = [torch.full((m, n), float("nan"), dtype=dtype, device=DEV) for _ in range(ws)]
    epoch, base = v.begin()                                    # (behind everything the caller's stream produced: operands and outputs)
    for c, (lo, hi) in enumerate(_chunks(m, n_chunks)):
        rows, off = hi - lo, base + lo * n * es
        for r in range(ws):                                    # "GEMM" into the peer buffer, then signal — every rank first
            with torch.cuda.stream(v.streams[r]):
                v.view(r, off, rows, n, dtype).copy_(parts[r][lo:hi])
            L.check(v.lib.mojo_hip_peer_signal(v.data, v.flags, ws, r, 0, c, epoch, v.sp(r)), "signal")
        for r in range(ws):                                    # pull-and-add the own share, write it back, raise kind 1
            r0, r1 = rows * r // ws, rows * (r + 1) // ws
            dst = outs[r][lo + r0: lo + r1]
            L.check(v.lib.mojo_hip_peer_reduce(v.data, v.flags, ws, r, c, epoch, off + r0 * n * es, r1 - r0, n,
                                               L.ptr(dst) if r1 > r0 else None, outs[r].stride(0), 1, code, v.sp(r)), "reduce")
        for r in range(ws):                                    # pull the other ranks' shares
            L.check(v.lib.mojo_hip_peer_gather(v.data, v.flags, ws, r, c, epoch, off, rows, n, L.ptr(outs[r][lo:hi]), outs[r].stride(0),
                                               code, v.sp(r)), "gather")
    v.end()
    return outs


def reduce_scatter(v, parts, n_chunks):
    """comm/peer.py gemm_reduce_scatter_direct: chunk c of the peer buffer holds [dest rank][rc rows]; rank r adds block r."""
    ws = v.ws
    m, n = parts[0].shape
    ml = m // ws
    dtype, es = parts[0].dtype, parts[0].element_size()
    code = L.dtype_code(dtype)
    outs = [torch.full((ml, n), float("nan"), dtype=dtype, device=DEV) for _ in range(ws)]
    epoch, base = v.begin()                                    # (behind everything the caller's stream produced: operands and outputs)
    for c, (lo, hi) in enumerate(_chunks(ml, n_chunks)):
        rc, off = hi - lo, base + ws * lo * n * es
        for r in range(ws):
            with torch.cuda.stream(v.streams[r]):                # the A-row map's product: sub-chunk c of every destination's rows
                blk = torch.cat([parts[r][d * ml + lo: d * ml + hi] for d in range(ws)])
                v.view(r, off, ws * rc, n, dtype).copy_(blk)
            L.check(v.lib.mojo_hip_peer_signal(v.data, v.flags, ws, r, 0, c, epoch, v.sp(r)), "signal")
        for r in range(ws):
            L.check(v.lib.mojo_hip_peer_reduce(v.data, v.flags, ws, r, c, epoch, off + r * rc * n * es, rc, n, L.ptr(outs[r][lo:hi]),
                                               outs[r].stride(0), 0, code, v.sp(r)), "reduce")
    v.end()
    return outs


def all_gather(v, shards):
    """comm/peer.py all_gather_gemm_direct's exchange: every rank pulls every rank's shard."""
    ws = v.ws
    ml, k = shards[0].shape
    dtype, es = shards[0].dtype, shards[0].element_size()
    outs = [torch.full((ws * ml, k), float("nan"), dtype=dtype, device=DEV) for _ in range(ws)]
    epoch, base = v.begin()                                    # (behind everything the caller's stream produced: operands and outputs)
    for r in range(ws):
        with torch.cuda.stream(v.streams[r]):
            v.view(r, base, ml, k, dtype).copy_(shards[r])
        L.check(v.lib.mojo_hip_peer_signal(v.data, v.flags, ws, r, 0, 0, epoch, v.sp(r)), "signal")
    for r in range(ws):
        L.check(v.lib.mojo_hip_peer_pull(v.data, v.flags, ws, r, 0, 0, epoch, base, ml * k * es, L.ptr(outs[r]), ml * k * es, 1, v.sp(r)), "pull")
    v.end()
    return outs


@pytest.mark.parametrize("ws", [4, 8])
@pytest.mark.parametrize("captured", [False, True], ids=["eager", "captured_mode"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_direct_exchange_kernels_with_virtual_ranks(ws, captured, dtype):
    """Six calls back to back on one set of buffers (epochs 1..6: both parity halves twice in eager mode; in captured mode the
    device-resident epoch and the "done reading" wait of every call on all ws - 1 peers): all-reduce with row counts that leave
    ranks unequal and EMPTY shares (M = ws - 1, 5), with several chunks, reduce-scatter, all-gather — every rank's result equal
    to the closed form, no wait timed out."""
    assert ws <= L.load().mojo_hip_peer_max_ranks()
    v = VirtualRanks(ws, 4 << 20, captured=captured)
    try:
        n = 256
        call = 0
        for m, n_chunks in ((ws - 1, 1), (5, 1), (8 * ws + 3, 2), (1024, 4)):
            parts = [_pattern(r, m, n, dtype, salt=call) for r in range(ws)]
            want = torch.stack([p.float() for p in parts]).sum(0).to(dtype)
            outs = all_reduce(v, parts, n_chunks)
            torch.cuda.synchronize()
            for r in range(ws):
                assert torch.equal(outs[r], want), f"all-reduce M {m}: rank {r} differs (call {call})"
            call += 1
        m = 64 * ws
        parts = [_pattern(r, m, n, dtype, salt=call) for r in range(ws)]
        total = torch.stack([p.float() for p in parts]).sum(0).to(dtype)
        outs = reduce_scatter(v, parts, 2)
        torch.cuda.synchronize()
        for r in range(ws):
            assert torch.equal(outs[r], total[r * 64:(r + 1) * 64]), f"reduce-scatter: rank {r} differs"
        shards = [_pattern(r, 48, n, dtype, salt=77) for r in range(ws)]
        outs = all_gather(v, shards)
        torch.cuda.synchronize()
        for r in range(ws):
            assert torch.equal(outs[r], torch.cat(shards)), f"all-gather: rank {r} differs"
        assert v.errors() == [0] * ws, "a bounded flag wait expired"
    finally:
        v.close()


def test_a_missing_peer_times_out_and_poisons_only_the_waiters():
    """One of eight ranks never signals: the seven waiters' bounded waits expire, their outputs are NaN, their error words
    are set and the grid drains (csrc/peer_comm.hip `wait_flag`)."""
    ws, n, m = 8, 256, 64
    v = VirtualRanks(ws, 1 << 20)
    try:
        v.lib.mojo_hip_peer_set_timeout_ms(300)
        es, code = 2, L.dtype_code(torch.bfloat16)
        outs = [torch.zeros(m, n, dtype=torch.bfloat16, device=DEV) for _ in range(ws)]
        pats = [_pattern(r, m, n, torch.bfloat16) for r in range(ws)]
        epoch, base = v.begin()
        for r in range(ws - 1):                                  # rank 7 stays silent
            with torch.cuda.stream(v.streams[r]):
                v.view(r, base, m, n, torch.bfloat16).copy_(pats[r])
            L.check(v.lib.mojo_hip_peer_signal(v.data, v.flags, ws, r, 0, 0, epoch, v.sp(r)), "signal")
        for r in range(ws - 1):
            r0, r1 = m * r // ws, m * (r + 1) // ws
            L.check(v.lib.mojo_hip_peer_reduce(v.data, v.flags, ws, r, 0, epoch, base + r0 * n * es, r1 - r0, n, L.ptr(outs[r][r0:r1]),
                                               n, 0, code, v.sp(r)), "reduce")
        errs = v.errors()
        assert errs[: ws - 1] == [1] * (ws - 1) and errs[ws - 1] == 0, errs
        for r in range(ws - 1):
            r0, r1 = m * r // ws, m * (r + 1) // ws
            assert torch.isnan(outs[r][r0:r1].float()).all()
    finally:
        v.close()
