"""Which exchange a GEMM + collective operator uses: the collective library's ring per row chunk (``"rccl"``) or the direct
peer exchange over HIP-IPC buffers (``"direct"``, comm/peer.py).

The reference has one path per backend (`core/operators/compute_with_comm.py:57-116`: GEMM, then a blocking collective).  Here
there are two, and which one is faster depends on the fabric and the payload: a ring moves 2(ws-1)/ws of the payload over ONE
xGMI link per GPU, the direct form the same bytes over ws-1 links at once, but it has only ever run with two ranks on one
device.  So nothing is assumed:

* ``MOJO_HIP_COMM_DIRECT=1`` / ``=0`` force a path (as before).
* Unset, the choice is made ONCE per (process group, operator, payload bucket), collectively, the first time an operator
  meets that key:
    1. once per group, a SELF-TEST of the direct exchange on deterministic integer data — all-reduce, reduce-scatter and
       all-gather through the real kernels, bit-compared with the closed-form sums (exact in fp32 and in the storage type);
       bounded flag waits, the sticky error word read afterwards.  Any exception, timeout or mismatch on ANY rank (the
       verdicts are exchanged) disables the direct path for the group for good;
    2. both paths are TIMED on the caller's own operands (one warm-up, ``TIMED`` calls each, HIP events, MAX over ranks) and
       the faster one is cached.
  Every rank takes the same decisions because every number that decides is reduced over the group first.
* Under HIP-graph capture nothing can be tested or timed: the direct exchange is taken only where it was forced or chosen
  BEFORE the capture and its graph-capturable twin exists (device-resident epoch, one data area guarded by "done reading"
  flags: comm/peer.py, csrc/peer_comm.hip); otherwise "rccl".

``report()`` returns what was decided and why (bench.py puts it on the result line).
"""
import os
from typing import Callable, Dict, Optional, Tuple

import torch
import torch.distributed as dist

TIMED = 3
_SELF_TEST: Dict[object, Tuple[bool, str]] = {}              # group key -> (direct usable, why)
_CHOICE: Dict[Tuple[object, str, int], dict] = {}            # (group key, op, bucket) -> {"algorithm", "direct_us", "rccl_us", ...}


def _key(group):
    return getattr(group, "group_name", None) or id(group)


def forced() -> Optional[str]:
    env = os.environ.get("MOJO_HIP_COMM_DIRECT")
    if env == "1":
        return "direct"
    if env == "0":
        return "rccl"
    return None


def bucket(nbytes: int) -> int:
    """Payload bucket: the next power of two, at least 1 MiB."""
    b = 1 << 20
    while b < nbytes:
        b <<= 1
    return b


def _agree_max(group, device, *values: float):
    """MAX over the ranks of each value (on the device for RCCL groups, on the host for gloo)."""
    where = device if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor(list(values), dtype=torch.float64, device=where)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return [float(v) for v in t.cpu()]


class _RowCopyEngine:
    """`GemmEngine` whose "product" is the input row itself (N = K): drives the exchange kernels without a GEMM."""

    def __init__(self, n):
        self.n = n

    def out_features(self, weight, trans_weight):
        return self.n

    def __call__(self, x, weight, bias, trans_weight, *, out=None, rows=None, a_map=None, c_map=None):
        rows = x.shape[0] if rows is None else rows
        m = torch.arange(rows, device=x.device)

        def mapped(mp):
            return m if mp is None else (m // mp[0]) * mp[1] + mp[2] + m % mp[0]

        if out is None:
            out = torch.empty(rows, self.n, dtype=x.dtype, device=x.device)
        out[mapped(c_map)] = x[mapped(a_map)]
        return out


def _pattern(rank: int, rows: int, n: int, device, dtype):
    i = torch.arange(rows, device=device, dtype=torch.int64).unsqueeze(1)
    j = torch.arange(n, device=device, dtype=torch.int64).unsqueeze(0)
    return (((i * 131 + j * 7 + rank * 29) % 17) - 8).to(dtype)          # integers in [-8, 8]: sums over <= 16 ranks are exact


def self_test(group, device) -> Tuple[bool, str]:
    """(usable, why) of the direct exchange on this group; cached; the same verdict on every rank."""
    key = _key(group)
    if key in _SELF_TEST:
        return _SELF_TEST[key]
    from . import peer

    ws, rank = dist.get_world_size(group), dist.get_rank(group)
    err = None
    old_timeout = os.environ.get("MOJO_HIP_PEER_TIMEOUT_MS")
    os.environ.setdefault("MOJO_HIP_PEER_TIMEOUT_MS", "3000")
    try:
        dtype, rows, n = torch.bfloat16, 2048 * ws, 1024                # four chunks per call
        eng = _RowCopyEngine(n)
        parts = [_pattern(r, rows, n, device, dtype) for r in range(ws)]
        total = torch.stack([p.float() for p in parts]).sum(0).to(dtype)
        got = peer.gemm_all_reduce_direct(eng, parts[rank], None, None, True, group)
        if not torch.equal(got, total):
            err = "all-reduce differs from the closed form"
        got = peer.gemm_reduce_scatter_direct(eng, parts[rank], None, None, True, group)
        ml = rows // ws
        if err is None and not torch.equal(got, total[rank * ml:(rank + 1) * ml]):
            err = "reduce-scatter differs from the closed form"
        shard = parts[rank][:ml].contiguous()
        got = peer.all_gather_gemm_direct(eng, shard, None, None, True, group)
        if err is None and not torch.equal(got, torch.cat([p[:ml] for p in parts])):
            err = "all-gather differs from the closed form"
        for ex in list(peer._CACHE.values()):
            if ex.group is group:
                ex.check()                                               # sticky error word of the bounded waits
    except Exception as e:                                               # set-up failures, unsupported sizes, timeouts
        err = repr(e)
    finally:
        if old_timeout is None:
            os.environ.pop("MOJO_HIP_PEER_TIMEOUT_MS", None)
    verdicts = [None] * ws
    dist.all_gather_object(verdicts, err, group=group)
    bad = {r: v for r, v in enumerate(verdicts) if v}
    res = (not bad, "self-test passed (all-reduce, reduce-scatter, all-gather bit-exact on integer data)" if not bad
           else f"self-test failed: {bad}")
    _SELF_TEST[key] = res
    return res


def _time(fn: Callable[[], object], device) -> float:
    fn()                                                                 # warm-up (allocations, first-call set-up)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(device)
    e0.record()
    for _ in range(TIMED):
        fn()
    e1.record()
    torch.cuda.synchronize(device)
    return e0.elapsed_time(e1) * 1e3 / TIMED                             # us per call


def choose(group, op: str, payload_bytes: int, x: torch.Tensor, run_direct: Callable[[], object],
           run_rccl: Callable[[], object]) -> str:
    """"direct" or "rccl" for this (group, operator, payload); see the module docstring.  Collective: every rank calls it
    with the same arguments in the same order (the operators are SPMD)."""
    if group is None or not x.is_cuda or dist.get_world_size(group) <= 1:
        return "rccl"
    f = forced()
    key = (_key(group), op, bucket(payload_bytes))
    rec = _CHOICE.get(key)
    if torch.cuda.is_current_stream_capturing():
        # nothing can be tested or timed under capture: the direct exchange only where it was forced or chosen BEFORE the
        # capture and its graph-capturable twin (device-resident epoch, comm/peer.py) exists at this size
        from . import peer

        want = f if f is not None else (rec["algorithm"] if rec is not None else "rccl")
        return "direct" if want == "direct" and peer.captured_ready(group, payload_bytes) else "rccl"
    if f is not None:
        return f
    if rec is not None:
        return rec["algorithm"]
    ok, why = self_test(group, x.device)
    rec = {"op": op, "payload_bucket_MB": key[2] / 2 ** 20, "self_test": why}
    if not ok:
        rec["algorithm"] = "rccl"
    else:
        from . import peer

        failed = 0.0
        t_direct = float("inf")
        try:
            t_direct = _time(run_direct, x.device)
            for ex in list(peer._CACHE.values()):
                if ex.group is group:
                    ex.check()
        except Exception as e:
            failed, rec["direct_error"] = 1.0, repr(e)
        t_rccl = _time(run_rccl, x.device)
        failed, = _agree_max(group, x.device, failed)
        if failed:
            _SELF_TEST[key[0]] = (False, "the direct exchange failed while being timed")
            rec["algorithm"] = "rccl"
        else:
            t_direct, t_rccl = _agree_max(group, x.device, t_direct, t_rccl)
            rec.update({"direct_us": t_direct, "rccl_us": t_rccl, "algorithm": "direct" if t_direct < t_rccl else "rccl"})
    _CHOICE[key] = rec
    return rec["algorithm"]


def report():
    """What was decided so far: one record per (operator, payload bucket)."""
    return [dict(v) for v in _CHOICE.values()]


def reset():
    _SELF_TEST.clear()
    _CHOICE.clear()
