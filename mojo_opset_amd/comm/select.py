"""Which exchange a GEMM + collective operator uses: the collective library's ring per row chunk (``"rccl"``) or the direct
peer exchange over HIP-IPC buffers (``"direct"``, comm/peer.py).

The reference has one path per backend (`core/operators/compute_with_comm.py:57-116`: GEMM, then a blocking collective).  Here
there are two, and which one is faster depends on the fabric and the payload: a ring moves 2(ws-1)/ws of the payload over ONE
xGMI link per GPU, the direct form the same bytes over ws-1 links at once, but it has only ever run with two ranks on one
device.  So nothing is assumed:

* ``MOJO_HIP_COMM_DIRECT=1`` / ``=0`` force a path.
* Unset, the choice is made ONCE per (process group, operator, payload bucket), collectively, by ``decide``:
    1. once per group, a SELF-TEST of the direct exchange on deterministic integer data — all-reduce, reduce-scatter and
       all-gather through the real kernels, bit-compared with the closed-form sums (exact in fp32 and in the storage type);
       flag waits bounded at 3 s for the test's own launches (``mojo_hip_peer_set_timeout_ms``, restored afterwards), the
       sticky error word read afterwards.  Any exception, timeout or mismatch on ANY rank (the verdicts are exchanged)
       disables the direct path for the group for good;
    2. both paths are TIMED (one warm-up, ``TIMED`` calls each, HIP events, MAX over ranks); the direct exchange is taken
       only where it is at least ``MARGIN`` (10 %) faster — near-equal timings must not flip the algorithm, and with it the
       bits of a reduction (direct: one fp32 sum in rank order, one rounding; ring: storage-type partial sums), between
       two runs of one job.  The decision is cached for the life of the process.
  WHEN that happens (round 5): in ``warm(group, shapes)``, which a serving engine calls once at start-up next to its other
  warm-up passes (graph capture, allocator priming) with the (operator, M, K_local, N, dtype) tuples it will run — NOT inside
  the first user call, which used to stall for a self-test (IPC set-up, 2 x 64 MiB of peer buffers) plus eight timed calls.
  A key that was never warmed takes the collective-library pipeline at once (recorded as such in ``report()``);
  ``MOJO_HIP_COMM_AUTOTUNE=1`` restores the decide-on-first-call behaviour for hosts that cannot enumerate their shapes.
  Every rank takes the same decisions because every number that decides is reduced over the group first.  Auto mode is
  reproducible run to run only up to this timing; pin ``MOJO_HIP_COMM_DIRECT`` where bit-reproducibility matters.
* Under HIP-graph capture nothing can be tested or timed: the direct exchange is taken only where it was forced or chosen
  BEFORE the capture and its graph-capturable twin exists (device-resident epoch, one data area guarded by "done reading"
  flags: comm/peer.py, csrc/peer_comm.hip); otherwise "rccl".

``report()`` returns what was decided and why (bench.py puts it on the result line).
"""
from typing import Callable, Dict, Iterable, Optional, Tuple

import torch
import torch.distributed as dist

from .. import switches

TIMED = 3
MARGIN = 0.10                                                # the direct exchange must be this much faster to be chosen
SELF_TEST_TIMEOUT_MS = 3000
_SELF_TEST: Dict[object, Tuple[bool, str]] = {}              # group key -> (direct usable, why)
_CHOICE: Dict[Tuple[object, str, int], dict] = {}            # (group key, op, bucket) -> {"algorithm", "direct_us", "rccl_us", ...}


def _key(group):
    return getattr(group, "group_name", None) or id(group)


def forced() -> Optional[str]:
    env = switches.get("MOJO_HIP_COMM_DIRECT")
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

    from ..backends.hip import lib as L

    ws, rank = dist.get_world_size(group), dist.get_rank(group)
    err = None
    # the test's own launches wait at most 3 s for a flag; the bound is an ARGUMENT of each launch (csrc/peer_comm.hip), so
    # restoring the previous setting afterwards really restores it for every later exchange (the environment
    # variable this used to set was latched by the library on first use)
    old_timeout = L.load().mojo_hip_peer_set_timeout_ms(SELF_TEST_TIMEOUT_MS)
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
        torch.cuda.synchronize(device)
        peer.check_all(group)                                            # sticky error word of the bounded waits
    except Exception as e:                                               # set-up failures, unsupported sizes, timeouts
        err = repr(e)
    finally:
        try:
            torch.cuda.synchronize(device)                               # the test's launches ran under the short bound
        except Exception:
            pass
        L.load().mojo_hip_peer_set_timeout_ms(old_timeout)
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


_WARMING = 0                                                 # > 0 while warm() runs: unknown keys are decided, not defaulted


def decide(group, op: str, payload_bytes: int, x: torch.Tensor, run_direct: Callable[[], object],
           run_rccl: Callable[[], object]) -> dict:
    """Self-test (once per group) + timing of both paths for this (group, operator, payload bucket); caches and returns the
    record.  Collective: every rank calls it with the same arguments in the same order."""
    key = (_key(group), op, bucket(payload_bytes))
    ok, why = self_test(group, x.device)
    rec = {"op": op, "payload_bucket_MB": key[2] / 2 ** 20, "world": dist.get_world_size(group), "self_test": why}
    if not ok:
        rec["algorithm"] = "rccl"
    else:
        from . import peer

        failed = 0.0
        t_direct = float("inf")
        try:
            t_direct = _time(run_direct, x.device)
            peer.check_all(group)
        except Exception as e:
            failed, rec["direct_error"] = 1.0, repr(e)
        t_rccl = _time(run_rccl, x.device)
        failed, = _agree_max(group, x.device, failed)
        if failed:
            _SELF_TEST[key[0]] = (False, "the direct exchange failed while being timed")
            rec["algorithm"] = "rccl"
        else:
            t_direct, t_rccl = _agree_max(group, x.device, t_direct, t_rccl)
            rec.update({"direct_us": t_direct, "rccl_us": t_rccl, "margin": MARGIN,
                        "algorithm": "direct" if t_direct < (1.0 - MARGIN) * t_rccl else "rccl"})
    _CHOICE[key] = rec
    return rec


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
    if rec is not None and not (rec.get("unwarmed") and _WARMING):
        return rec["algorithm"]
    if _WARMING or switches.get("MOJO_HIP_COMM_AUTOTUNE", "0") == "1":
        return decide(group, op, payload_bytes, x, run_direct, run_rccl)["algorithm"]
    _CHOICE[key] = {"op": op, "payload_bucket_MB": key[2] / 2 ** 20, "world": dist.get_world_size(group), "algorithm": "rccl",
                    "unwarmed": True, "why": "not warmed: comm.select.warm(group, shapes) lets the direct exchange compete"}
    return "rccl"


def warm(group, shapes: Iterable[Tuple[str, int, int, int, torch.dtype]], device=None):
    """Decide the exchange for the given workloads NOW — at start-up, outside any serving step.  ``shapes``: tuples
    ``(operator, M, K_local, N, dtype)`` with operator in {"gemm_all_reduce", "gemm_reduce_scatter", "all_gather_gemm"}; M is
    the row count of the FULL product (all-gather: every rank holds M / ws rows of the [M, K_local] input), K_local the
    reduction length on one rank, N the output columns on one rank.  Runs the real operators on synthetic operands (weights
    ``[K_local, N]``), so the keys are the ones user calls will look up.  Collective; returns ``report()``."""
    global _WARMING
    from ..backends.hip.operators.compute_with_comm import HIPAllGatherGemm, HIPGemmAllReduce, HIPGemmReduceScatter

    ws = dist.get_world_size(group)
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    _WARMING += 1
    try:
        for op, m, k, n, dtype in shapes:
            w = torch.randn(k, n, device=device, dtype=torch.float32).mul_(0.02).to(dtype)
            if op == "gemm_all_reduce":
                mod, rows = HIPGemmAllReduce(w, None, True, process_group=group), m
            elif op == "gemm_reduce_scatter":
                mod, rows = HIPGemmReduceScatter(w, None, True, scatter_dim=0, process_group=group), m
            elif op == "all_gather_gemm":
                mod, rows = HIPAllGatherGemm(w, None, True, gather_dim=0, process_group=group), m // ws
            else:
                raise ValueError(f"select.warm: unknown operator {op!r}")
            if rows <= 0 or (op != "gemm_all_reduce" and m % ws):
                continue
            mod(torch.randn(rows, k, device=device, dtype=torch.float32).to(dtype))
        torch.cuda.synchronize(device)
    finally:
        _WARMING -= 1
    return report()


def disable_direct(group, why: str) -> None:
    """Take the direct exchange out of the running for ``group`` (a rank saw it fail): the self-test verdict becomes ``why``
    and every cached choice of the group goes back to the collective-library pipeline.  Collective in spirit: call it on
    every rank (the callers agree on the failure through a max-reduce first)."""
    key = _key(group)
    _SELF_TEST[key] = (False, why)
    for (k, _op, _b), rec in _CHOICE.items():
        if k == key and rec.get("algorithm") == "direct":
            rec.update({"algorithm": "rccl", "disabled": why})


def report():
    """What was decided so far: one record per (operator, payload bucket)."""
    return [dict(v) for v in _CHOICE.values()]


def reset():
    _SELF_TEST.clear()
    _CHOICE.clear()
