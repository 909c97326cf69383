"""One rank of the multi-process GPU test of the GEMM + collective operators (tests/test_hip_comm_ranks.py starts
``WORLD_SIZE`` copies of this script as child processes).

The test box has ONE MI355X, and RCCL refuses two ranks on one device, so the ranks share ``cuda:0`` and the process
group is gloo (device tensors are staged by gloo) — the same arrangement the reference uses when it runs these ops
on a host without its vendor collective library (`tests/dist_common.py:38-81`).  What runs is the product path:
`HIP<Op>.forward` -> `mojo_opset_amd.comm` pipelines -> `mojo_hip_gemm_rowmap` through the C ABI, chunked, with the
exchange step on the process group; with ``MOJO_HIP_COMM_DIRECT=1`` the all-reduce / reduce-scatter / all-gather exchange
is this repository's own pull kernels over HIP-IPC peer buffers instead.

Checks (reference: tests/accuracy/operators/test_compute_with_comm.py):
  * the reference's per-rank vectors (tests/golden/compute_with_comm.pt) at the reference's bounds (5e-3 / 1e-4);
  * the oracle classes running over the same gloo group on CPU copies of the same per-rank inputs, at the reference's
    seeds and a quarter of each of its dimensions (:97-103, :141-146, :180-185, :215-247), and the golden's definition in
    fp32 torch on the device at the reference's full shapes (a full-size fp16 product takes the box's CPU share ~40 s).
Writes one JSON line per check to stdout; exits non-zero on the first failure.
"""
import json
import os
import sys
import traceback

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

DEV = "cuda"


def _setenv(**values):
    """Set (None: unset) MOJO_HIP_* switches and make both layers re-read them (latched at first use)."""
    from mojo_opset_amd import switches

    for k, v in values.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    switches.reload()


def _ulps(got, want):
    from hip_utils import max_ulp_bf16ish
    return max_ulp_bf16ish(got, want, atol=1e-3)


def _report(rank, **kw):
    print(json.dumps(kw), flush=True)
    try:                                       # progress files: a long multi-process run must not look hung from outside
        with open(os.path.join(os.path.dirname(HERE), "gpurun_out", f"comm_ranks_progress_rank{rank}.log"), "a") as f:
            f.write(json.dumps(kw) + "\n")
    except OSError:
        pass


def _ulp_at(mag, dtype):
    mant = {torch.bfloat16: 7, torch.float16: 10}[dtype]
    return torch.pow(2.0, torch.floor(torch.log2(mag.double().clamp_min(2.0 ** -14))) - mant)


def _partials_magnitude(ws, m, k, n, dtype, rank=None):
    """sum over ranks of |round(x_r @ w_r)| for the per-rank recipe of the reference's tests (seed 42 + r, x [m, k / ws], w
    [k / ws, n]): the magnitude at which the ws storage-type partials of an element are added.  (rank: the reduce-scatter chunk.)"""
    kl = k // ws
    mag = None
    for r in range(ws):
        torch.manual_seed(42 + r)
        x = torch.randn(m, kl, dtype=dtype).to(DEV)
        w = torch.randn(kl, n, dtype=dtype).to(DEV)
        y = (x.float() @ w.float()).to(dtype).float().abs()
        mag = y if mag is None else mag + y
    if rank is not None:
        mag = mag.chunk(ws, dim=0)[rank]
    return mag.cpu()


def _compare(rank, name, got, want, tol, mag=None):
    """The reference's bound is atol = rtol = tol (5e-3 for 16-bit data, 1e-4 for fp32).  Both sides round every rank's
    product to the storage type and then add storage-type values, so they can differ only where the fp32 accumulation
    order of the two GEMMs flips the rounding of a PARTIAL: at most one unit in the last place of each rank's partial.  For
    bf16 one ulp is 0.39-0.78 % — above 5e-3 — and where the ranks' partials cancel, one ulp of a large partial exceeds
    5e-3 of the small sum in fp16 too (measured: 4e-5 of the elements at 4096^3).  The 16-bit bound is therefore stated as
    '>= 99.9 % of the elements inside the reference's bound (bf16: 99.5 %), none farther than one ulp per rank at the
    largest output magnitude'; fp32 meets the reference's bound outright.  With MORE than two ranks the golden's own
    result depends on the association the collective library happens to use for the ws storage-type partials (gloo's ring
    here, RCCL's on a node; the direct exchange adds them in fp32 and rounds once): every one of the ws - 1 additions may
    move an element by an ulp of a partial sum, so the share of elements inside 5e-3 falls (measured at four ranks: 95.5 %
    of the reference's own bf16 reduce-scatter vectors, 79 % at 1024^3 bf16).  Where the caller supplies `mag` = sum_r |y_r|
    (the magnitude the partials are added at), the bound is stated per element in those units: |got - want| <= ws ulps at
    `mag` (ws - 1 additions + one for the GEMMs' own summation order), for EVERY element."""
    got, want = got.detach().cpu(), torch.as_tensor(want).detach().cpu()
    assert got.shape == want.shape and got.dtype == want.dtype, (name, got.shape, want.shape, got.dtype, want.dtype)
    diff = (got.double() - want.double()).abs()
    inside = diff <= tol + tol * want.double().abs()
    frac = float(inside.double().mean())
    rec = {"check": name, "max_abs": float(diff.max()), "frac_inside_ref_tol": frac}
    if got.dtype == torch.float32:
        ok = frac == 1.0
    else:
        mant = {torch.bfloat16: 7, torch.float16: 10}[got.dtype]
        top = float(want.double().abs().max())
        ulp_top = 2.0 ** (torch.floor(torch.log2(torch.tensor(max(top, 2.0 ** -14)))).item() - mant)
        ws = dist.get_world_size()
        rec["max_ulp"] = _ulps(got, want)
        rec["bound_abs"] = 2 * ws * ulp_top
        if ws > 2 and mag is not None:
            excess = diff - ws * _ulp_at(mag, got.dtype)
            rec["max_excess_over_ws_ulps_at_partials"] = float(excess.max())
            ok = float(excess.max()) <= 0 and rec["max_abs"] <= rec["bound_abs"]
        else:
            ok = frac >= (0.995 if got.dtype == torch.bfloat16 else 0.999) and rec["max_abs"] <= rec["bound_abs"]
    _report(rank, **rec)
    assert ok, rec


def check_vectors(rank, ws, group):
    from conftest import load_golden
    from hip_utils import hip_cls

    for case in load_golden("compute_with_comm"):
        if len(case["ranks"]) != ws:
            continue
        me = case["ranks"][rank]
        x, w, want = me["x"].to(DEV), me["w"].to(DEV), me["out"]
        op = hip_cls(case["op"])(w, None, True, process_group=group, **case["ctor_kwargs"])
        got = op(x)
        torch.cuda.synchronize()
        mag = None
        if ws > 2 and x.dtype != torch.float32 and case["op"] in ("MojoGemmAllReduce", "MojoGemmReduceScatter"):
            mag = torch.stack([(c["x"].float() @ c["w"].float()).to(x.dtype).float().abs() for c in case["ranks"]]).sum(0)
            if case["op"] == "MojoGemmReduceScatter":
                mag = mag.chunk(ws, dim=0)[rank]
        _compare(rank, f"vector:{case['op']}:{case['name']}", got, want, 5e-3 if x.dtype != torch.float32 else 1e-4, mag=mag)


_CACHE = {}


def _scales():
    """Divisors of the reference / config-4 shapes to run: 4 = oracle over the same gloo group at a quarter of every dimension,
    1 = the golden's definition in fp32 torch on the device at full size.  MOJO_TEST_COMM_SCALES selects (default both)."""
    return tuple(int(v) for v in os.environ.get("MOJO_TEST_COMM_SCALES", "4,1").split(",") if v)


def _cached(key, make):
    if key not in _CACHE:
        _CACHE[key] = make()
    return _CACHE[key]


def _device_closed_form(ws, rank, m, k, n, dtype, op):
    """What the golden defines, evaluated without communication on this rank's GPU in fp32 torch (the allowed fp32
    reference of a floating-point kernel): every rank's inputs follow from its seed (42 + r), its product is rounded to the
    storage type, the rounded products are added in the storage type (2 ranks: one addition, order-free)."""
    kl = k // ws
    total = None
    for r in range(ws):
        torch.manual_seed(42 + r)
        x = torch.randn(m, kl, dtype=dtype).to(DEV)
        w = torch.randn(kl, n, dtype=dtype).to(DEV)
        y = (x.float() @ w.float()).to(dtype)
        total = y if total is None else (total + y)
    if op == "MojoGemmReduceScatter":
        total = total.chunk(ws, dim=0)[rank]
    return total.cpu()


def check_reference_shapes(rank, ws, group):
    """The reference's own cases and seeds (tests/accuracy/operators/test_compute_with_comm.py:97-103, :141-146, :180-185,
    :215-247).  The pinned oracle (CPU golden classes over the same gloo group) is the other side at a QUARTER of every
    dimension — the box grants 16 CPU cores and has no fast fp16 GEMM: one full-size fp16 oracle product takes ~40 s there —
    and at the reference's full shapes the other side is the golden's definition evaluated in fp32 torch on the device."""
    import torch.nn.functional as F

    from hip_utils import hip_cls, torch_cls

    full = ((4096, 4096, 4096, torch.float16), (2048, 8192, 4096, torch.float16), (8192, 4096, 2048, torch.float16),
            (4096, 4096, 4096, torch.bfloat16))
    for scale in _scales():
        for m0, k0, n0, dtype in full:
            m, k, n = m0 // scale, k0 // scale, n0 // scale
            kl = k // ws
            torch.manual_seed(42 + rank)
            x = torch.randn(m, kl, dtype=dtype)
            w = torch.randn(kl, n, dtype=dtype)
            tag = f"{m}x{k}x{n}:{str(dtype)[6:]}"
            for name, kw in (("MojoGemmAllReduce", {}), ("MojoGemmReduceScatter", {"scatter_dim": 0})):
                if scale == 1:
                    want = _cached((name, tag), lambda: _device_closed_form(ws, rank, m, k, n, dtype, name))
                    side = "fp32ref"
                else:
                    want = _cached((name, tag), lambda: torch.as_tensor(
                        torch_cls(name)(weight=w, bias=None, trans_weight=True, process_group=group, **kw)(x)).clone())
                    side = "oracle"
                got = hip_cls(name)(weight=w.to(DEV), bias=None, trans_weight=True, process_group=group, **kw)(x.to(DEV))
                torch.cuda.synchronize()
                mag = _cached(("mag", name, tag), lambda: _partials_magnitude(ws, m, k, n, dtype, rank if name == "MojoGemmReduceScatter" else None)) if ws > 2 else None
                _compare(rank, f"{side}:{name}:{tag}", got, want, 5e-3, mag=mag)
    # AllGatherGemm: :97-103, one seed for all ranks, [N, K] weights, bias in the fp16 cases
    for scale in _scales():
        for m0, k0, n0, dtype, use_bias in ((4096, 4096, 4096, torch.float16, True), (2048, 4096, 8192, torch.float16, True),
                                            (8192, 2048, 4096, torch.float16, True), (4096, 4096, 4096, torch.bfloat16, False)):
            m, k, n = m0 // scale, k0 // scale, n0 // scale
            torch.manual_seed(42)
            x_full = torch.randn(m, k, dtype=dtype)
            w = torch.randn(n, k, dtype=dtype)
            b = torch.randn(n, dtype=dtype) if use_bias else None
            ml = m // ws
            x = x_full[rank * ml:(rank + 1) * ml].contiguous()
            tag = f"{m}x{k}x{n}:{str(dtype)[6:]}"
            if scale == 1:
                def closed():
                    y = (x_full.to(DEV).float() @ w.to(DEV).float().t()).to(dtype)       # F.linear: product rounded, then the bias
                    return (y if b is None else (y + b.to(DEV))).cpu()
                want, side = _cached(("ag", tag), closed), "fp32ref"
            else:
                want = _cached(("ag", tag), lambda: torch.as_tensor(torch_cls("MojoAllGatherGemm")(
                    weight=w, bias=b, trans_weight=False, gather_dim=0, process_group=group)(x)).clone())
                side = "oracle"
            got = hip_cls("MojoAllGatherGemm")(weight=w.to(DEV), bias=None if b is None else b.to(DEV), trans_weight=False,
                                               gather_dim=0, process_group=group)(x.to(DEV))
            torch.cuda.synchronize()
            _compare(rank, f"{side}:MojoAllGatherGemm:{tag}", got, want, 5e-3)
    # GemmAll2All: :215-247, fp32, the closed form the reference test builds without communication
    torch.manual_seed(42)
    m, k, n = 32, 64, 128
    ml = m // ws
    x_full, w, b = torch.randn(m, k), torch.randn(n, k), torch.randn(n)
    shards = [x_full[i * ml:(i + 1) * ml].contiguous() for i in range(ws)]
    outs = [F.linear(s, w, b) for s in shards]
    want = torch.cat([outs[j].chunk(ws, dim=0)[rank] for j in range(ws)], dim=0)
    got = hip_cls("MojoGemmAll2All")(weight=w.to(DEV), bias=b.to(DEV), trans_weight=False, scatter_dim=0, gather_dim=0,
                                     process_group=group)(shards[rank].to(DEV))
    torch.cuda.synchronize()
    _compare(rank, "closed_form:MojoGemmAll2All:32x64x128:float32", got, want, 1e-4)


def check_fewer_rows_than_ranks(rank, ws, group):
    """Decode-sized row counts (M = 1, 3 with two ranks; M < ws in general): some rank's share of a chunk is EMPTY.  That
    rank must still raise its "share ready" flag (csrc/peer_comm.hip `mojo_hip_peer_reduce`, rows == 0) or its peers'
    gather step waits until the timeout and poisons every later call.  Checked on the direct exchange and, for the same
    shapes, on the collective-library pipeline; the other side is the golden's definition in fp32 torch on the device."""
    import time

    from hip_utils import hip_cls
    from mojo_opset_amd.comm import peer

    k, n = 1024, 2048
    for direct in ("1", "0"):
        _setenv(MOJO_HIP_COMM_DIRECT=direct)
        for m in (1, 3, 2 * ws, 5):
            for dtype in (torch.bfloat16, torch.float32):
                want = _device_closed_form(ws, rank, m, k, n, dtype, "MojoGemmAllReduce")
                torch.manual_seed(42 + rank)
                x = torch.randn(m, k // ws, dtype=dtype).to(DEV)
                w = torch.randn(k // ws, n, dtype=dtype).to(DEV)
                t0 = time.time()
                got = hip_cls("MojoGemmAllReduce")(weight=w, bias=None, trans_weight=True, process_group=group)(x)
                torch.cuda.synchronize()
                assert time.time() - t0 < 10, "an empty share stalled the exchange"
                _compare(rank, f"fp32ref:MojoGemmAllReduce:tiny{'_direct' if direct == '1' else ''}:{m}x{k}x{n}:{str(dtype)[6:]}",
                         got, want, 5e-3 if dtype != torch.float32 else 2e-4,
                         mag=_partials_magnitude(ws, m, k, n, dtype) if ws > 2 and dtype != torch.float32 else None)
        if direct == "1":
            assert peer._CACHE, "MOJO_HIP_COMM_DIRECT=1 but no peer exchange was built"
            for ex in peer._CACHE.values():
                ex.check()                           # no wait timed out
    _setenv(MOJO_HIP_COMM_DIRECT=None)


def check_config4_shapes(rank, ws, group):
    """BASELINE configs[3] at TP = ws: Llama-3-70B row-parallel GemmAllReduce (down-proj K 28672, o-proj K 8192, N 8192) and
    column-parallel AllGatherGemm (QKV N 10240, gate|up N 57344), M in {1024, 4096}, bf16, per-rank seed 42 + rank
    (tests/accuracy/operators/test_compute_with_comm.py:151).  The oracle (golden classes over the same gloo group on CPU)
    is the other side at a quarter of every dimension; at full size it is the golden's definition in fp32 torch on the device."""
    from hip_utils import hip_cls, torch_cls

    dtype = torch.bfloat16
    for scale in _scales():
        for m0 in (1024, 4096):
            for k0, n0 in ((28672, 8192), (8192, 8192)):
                m, k, n = m0 // scale, k0 // scale, n0 // scale
                tag = f"cfg4:{m}x{k}x{n}:tp{ws}"
                torch.manual_seed(42 + rank)
                x = torch.randn(m, k // ws, dtype=dtype)
                w = torch.randn(k // ws, n, dtype=dtype)
                if scale == 1:
                    want, side = _device_closed_form(ws, rank, m, k, n, dtype, "MojoGemmAllReduce"), "fp32ref"
                else:
                    want = torch.as_tensor(torch_cls("MojoGemmAllReduce")(weight=w, bias=None, trans_weight=True,
                                                                          process_group=group)(x)).clone()
                    side = "oracle"
                got = hip_cls("MojoGemmAllReduce")(weight=w.to(DEV), bias=None, trans_weight=True, process_group=group)(x.to(DEV))
                torch.cuda.synchronize()
                _compare(rank, f"{side}:MojoGemmAllReduce:{tag}", got, want, 5e-3, mag=_partials_magnitude(ws, m, k, n, dtype) if ws > 2 else None)
                del got, want
            for n_total in (10240, 57344):
                m, k, n = m0 // scale, 8192 // scale, n_total // scale // ws
                tag = f"cfg4:{m}x{k}x{n}:tp{ws}"
                torch.manual_seed(42)
                x_full = torch.randn(m, k, dtype=dtype)
                torch.manual_seed(142 + rank)                   # this rank's column shard of the weight
                w = torch.randn(n, k, dtype=dtype)
                ml = m // ws
                x = x_full[rank * ml:(rank + 1) * ml].contiguous()
                if scale == 1:
                    want, side = (x_full.to(DEV).float() @ w.to(DEV).float().t()).to(dtype).cpu(), "fp32ref"
                else:
                    want = torch.as_tensor(torch_cls("MojoAllGatherGemm")(weight=w, bias=None, trans_weight=False, gather_dim=0,
                                                                          process_group=group)(x)).clone()
                    side = "oracle"
                got = hip_cls("MojoAllGatherGemm")(weight=w.to(DEV), bias=None, trans_weight=False, gather_dim=0,
                                                   process_group=group)(x.to(DEV))
                torch.cuda.synchronize()
                _compare(rank, f"{side}:MojoAllGatherGemm:{tag}", got, want, 5e-3)
                del got, want
        torch.cuda.empty_cache()


def check_timeout_path(rank, ws, group):
    """Liveness of the peer exchange: a rank whose peer never shows up must not spin for ever.  Both ranks make one
    normal call (that builds the exchange); then only rank 0 calls again.  Its waits expire (MOJO_HIP_PEER_TIMEOUT_MS), the
    kernels drain, the output is NaN-poisoned and `PeerExchange.check()` raises."""
    import time

    from hip_utils import hip_cls
    from mojo_opset_amd.comm import peer

    _setenv(MOJO_HIP_COMM_DIRECT="1")
    torch.manual_seed(rank)
    x = torch.randn(512, 256, dtype=torch.bfloat16).to(DEV)
    w = torch.randn(256, 512, dtype=torch.bfloat16).to(DEV)
    op = hip_cls("MojoGemmAllReduce")(weight=w, bias=None, trans_weight=True, process_group=group)
    first = op(x)
    torch.cuda.synchronize()
    assert torch.isfinite(first.float()).all()
    for ex in peer._CACHE.values():
        ex.check()
    dist.barrier()
    if rank == 0:
        t0 = time.time()
        lonely = op(x)
        torch.cuda.synchronize()
        waited = time.time() - t0
        assert torch.isnan(lonely.float()).any(), "a timed-out exchange must poison its output"
        raised = False
        try:
            for ex in peer._CACHE.values():
                ex.check()
        except RuntimeError:
            raised = True
        assert raised, "PeerExchange.check() must report the expired wait"
        _report(rank, check="timeout:MojoGemmAllReduce", waited_s=round(waited, 2), poisoned=True, reported=True)
    dist.barrier()


def check_auto_selection(rank, ws, group):
    """comm/select.py with MOJO_HIP_COMM_DIRECT unset (round 5 semantics).  (1) A key that was never warmed takes the
    collective-library pipeline AT ONCE — no self-test, no timing inside a user call — and says so in the report.
    (2) `select.warm(group, shapes)` runs the group's self-test of the direct exchange (bit-exact on integer data, flag waits
    bounded through `mojo_hip_peer_set_timeout_ms`, restored afterwards) and times both paths per payload bucket; the direct
    exchange wins only with a 10 % margin.  (3) Afterwards a user call takes the cached choice: its result equals the forced
    path of the chosen algorithm bit for bit (and at two ranks both forced paths agree: one addition, order-free).  Every
    rank reaches the same decisions."""
    from hip_utils import hip_cls
    from mojo_opset_amd.backends.hip import lib as L
    from mojo_opset_amd.comm import peer, select

    select.reset()
    _setenv(MOJO_HIP_COMM_DIRECT=None, MOJO_HIP_COMM_AUTOTUNE=None)
    torch.manual_seed(7 + rank)
    k, n = 512, 1024
    w = (torch.randn(k, n, dtype=torch.bfloat16) * 0.05).to(DEV)
    # (1) unwarmed: the pipeline, immediately
    x = torch.randn(2048, k, dtype=torch.bfloat16).to(DEV)
    op = hip_cls("MojoGemmAllReduce")(weight=w, bias=None, trans_weight=True, process_group=group)
    first = op(x)
    rep = select.report()
    assert len(rep) == 1 and rep[0]["algorithm"] == "rccl" and rep[0].get("unwarmed") and not peer._CACHE, rep
    # (2) warm
    before = L.load().mojo_hip_peer_set_timeout_ms(0)
    L.load().mojo_hip_peer_set_timeout_ms(before)
    shapes = [(o, m, k, n, torch.bfloat16) for m in (2048, 8192) for o in ("gemm_all_reduce", "gemm_reduce_scatter", "all_gather_gemm")]
    rep = select.warm(group, shapes)
    after = L.load().mojo_hip_peer_set_timeout_ms(before)
    assert after == before, "the self-test must restore the flag-wait bound it shortened"
    assert len(rep) >= 5 and all(r["algorithm"] in ("direct", "rccl") and not r.get("unwarmed") for r in rep), rep
    assert all(r["self_test"].startswith("self-test passed") for r in rep), rep
    assert all("direct_us" in r and "rccl_us" in r and r["world"] == ws for r in rep), rep
    assert all(r["algorithm"] == ("direct" if r["direct_us"] < 0.9 * r["rccl_us"] else "rccl") for r in rep), rep      # the margin
    peer.check_all(group)
    # (3) user calls follow the cached choice
    outs = {}
    for m in (2048, 8192):
        x = torch.randn(m, k, dtype=torch.bfloat16).to(DEV)
        for name, key, kw in (("MojoGemmAllReduce", "gemm_all_reduce", {}), ("MojoGemmReduceScatter", "gemm_reduce_scatter", {"scatter_dim": 0})):
            op = hip_cls(name)(weight=w, bias=None, trans_weight=True, process_group=group, **kw)
            auto = op(x)
            torch.cuda.synchronize()
            for flag in ("0", "1"):
                _setenv(MOJO_HIP_COMM_DIRECT=flag)
                outs[flag] = op(x)
                torch.cuda.synchronize()
            _setenv(MOJO_HIP_COMM_DIRECT=None)
            choice = [r for r in select.report() if r["op"] == key and r["payload_bucket_MB"] == select.bucket(m * n * 2) / 2 ** 20]
            assert len(choice) == 1, (key, m, select.report())
            assert torch.equal(auto, outs["1" if choice[0]["algorithm"] == "direct" else "0"]), f"{name}: the selected path's result differs"
            if ws == 2:
                assert torch.equal(outs["0"], outs["1"]), f"{name}: pipeline and direct exchange disagree at two ranks"
            else:                                   # different associations of ws storage-type partials: ulps of the partials apart, not more
                mags = [None] * ws
                y = (x.float() @ w.float()).to(x.dtype).float().abs().cpu()
                dist.all_gather_object(mags, y, group=group)
                mag = torch.stack(mags).sum(0)
                if key == "gemm_reduce_scatter":
                    mag = mag.chunk(ws, dim=0)[rank]
                excess = (outs["1"].double().cpu() - outs["0"].double().cpu()).abs() - ws * _ulp_at(mag, x.dtype)
                assert float(excess.max()) <= 0, f"{name}: direct and pipeline differ by more than ws ulps at the partials' magnitude"
            _report(rank, check=f"auto:{name}:M{m}")
        xs = x[: m // ws].contiguous()
        op = hip_cls("MojoAllGatherGemm")(weight=w, bias=None, trans_weight=True, gather_dim=0, process_group=group)
        auto = op(xs)
        _setenv(MOJO_HIP_COMM_DIRECT="0")
        want = op(xs)
        _setenv(MOJO_HIP_COMM_DIRECT=None)
        torch.cuda.synchronize()
        assert torch.equal(auto, want)
        _report(rank, check=f"auto:MojoAllGatherGemm:M{m}")
    assert len(select.report()) == len(rep), "a warmed key must not be decided again"
    peer.check_all(group)
    # every rank reached the same decisions
    mine = [(r["op"], r["payload_bucket_MB"], r["algorithm"]) for r in select.report()]
    everyone = [None] * ws
    dist.all_gather_object(everyone, mine, group=group)
    assert all(e == mine for e in everyone), everyone
    del first
    _report(rank, selection=select.report())


def check_captured_direct(rank, ws, group):
    """The direct exchange under HIP-graph capture (device-resident epoch, one data area guarded by "done reading" flags:
    csrc/peer_comm.hip captured mode, comm/peer.py twin): GemmAllReduce -> GemmReduceScatter -> AllGatherGemm captured as ONE
    graph after an eager warm-up (which builds the twin), replayed on new inputs several times and compared, bit for bit,
    with the same operators run eagerly over the direct exchange on the same inputs.  Five replays: the single data area is
    reused every call, so a missing "done reading" wait shows up as a torn result."""
    from hip_utils import hip_cls
    from mojo_opset_amd.comm import peer

    _setenv(MOJO_HIP_COMM_DIRECT="1", MOJO_HIP_COMM_CHUNKS="2", MOJO_HIP_PEER_MIN_BYTES=str(1 << 20))
    dtype = torch.bfloat16
    m, k, n = 64, 512, 1024                                   # decode-sized rows: the case graphs exist for
    torch.manual_seed(7 + rank)
    w1 = (torch.randn(k, n) * 0.05).to(dtype).to(DEV)
    w2 = (torch.randn(n, k) * 0.05).to(dtype).to(DEV)
    w3 = (torch.randn(k, n) * 0.05).to(dtype).to(DEV)
    ar = hip_cls("MojoGemmAllReduce")(weight=w1, bias=None, trans_weight=True, process_group=group)
    rs = hip_cls("MojoGemmReduceScatter")(weight=w2, bias=None, trans_weight=True, process_group=group)
    ag = hip_cls("MojoAllGatherGemm")(weight=w3, bias=None, trans_weight=True, process_group=group)
    x = torch.zeros(m, k, dtype=dtype, device=DEV)

    def step():
        a = ar(x)                                             # [m, n] on every rank
        b = rs(a)                                             # [m / ws, k]
        return ag(b)                                          # [m, n]

    def load(i):
        g = torch.Generator().manual_seed(1000 * i + rank)
        x.copy_(torch.randn(m, k, generator=g).to(dtype))

    load(0)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()                                               # eager warm-up: builds the exchange and its captured twin
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    dist.barrier(group=group)
    assert all(ex.twin is not None for ex in peer._CACHE.values()) and peer._CACHE
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = step()
    for i in range(1, 6):
        load(i)
        graph.replay()
        torch.cuda.synchronize()
        got = out.clone()
        want = step()                                         # eager, over the (non-captured) direct exchange
        torch.cuda.synchronize()
        assert torch.isfinite(got.float()).all(), f"replay {i}: poisoned output"
        assert torch.equal(got, want), f"replay {i}: captured and eager direct exchange disagree (max diff {(got.float() - want.float()).abs().max().item()})"
    peer.check_all(group)                                     # eager exchanges AND their captured twins
    # a later eager call with a larger payload rebuilds the exchange.  The twin the graph was captured over must
    # stay mapped (the graph holds its raw peer pointers): it is retired, not freed, and the graph keeps replaying correctly.
    # (nothing is freed to get there — an uncached buffer freed and re-allocated at the same address showed its OLD contents to
    # peers on this platform: the payload is simply made larger than whatever exchange an earlier mode left behind)
    cap_now = max(ex.capacity for ex in peer._CACHE.values())
    retired_before = len(peer._RETIRED)
    big = hip_cls("MojoGemmAllReduce")(weight=(torch.randn(k, 4096) * 0.05).to(dtype).to(DEV), bias=None, trans_weight=True, process_group=group)
    xb = torch.randn(cap_now // (4096 * 2) + 256, k).to(dtype).to(DEV)          # payload just above the current capacity
    yb = big(xb)
    torch.cuda.synchronize()
    assert torch.isfinite(yb.float()).all()
    assert len(peer._RETIRED) == retired_before + 1 and peer._RETIRED[-1].twin is not None and peer._RETIRED[-1].twin.handed_out_under_capture, \
        "the exchange the graph was captured over must be retired, not freed"
    assert max(ex.capacity for ex in peer._CACHE.values()) >= 2 * cap_now, "capacity at least doubles on a rebuild"
    del yb, xb, big
    for i in range(6, 9):
        load(i)
        graph.replay()
        torch.cuda.synchronize()
        got = out.clone()
        want = step()
        torch.cuda.synchronize()
        assert torch.equal(got, want), f"replay {i} after the exchange grew: captured and eager disagree"
    peer.check_all(group)
    _report(rank, check="captured:direct_exchange:5_replays", ok=True, replays_after_growth=3)
    _setenv(MOJO_HIP_COMM_DIRECT=None, MOJO_HIP_COMM_CHUNKS=None, MOJO_HIP_PEER_MIN_BYTES=None)


def main():
    import faulthandler

    rank, ws = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    # The oracle side is CPU work (a 4096 x 2048 x 4096 matmul per case and rank).  torch sizes its thread pool from the
    # cores it SEES, the box grants a fraction of them: two ranks x all-visible-cores OpenMP teams spinning on a 16-core
    # share turned seconds into minutes.  A fixed small team per rank keeps the oracle in seconds.
    torch.set_num_threads(int(os.environ.get("MOJO_TEST_RANK_THREADS", "6")))
    faulthandler.dump_traceback_later(int(os.environ.get("MOJO_TEST_DUMP_AFTER_S", "120")), exit=False)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    group = dist.group.WORLD
    rc = 0
    try:
        modes = [m for m in os.environ.get("MOJO_TEST_COMM_MODES", "chunks1,chunks4").split(",") if m]
        for mode in modes:
            _setenv(MOJO_HIP_COMM_DIRECT="0")     # pipeline unless the mode says otherwise ("auto": unset, comm/select.py decides)
            if mode == "auto":
                _report(rank, mode=mode)
                _setenv(MOJO_HIP_COMM_DIRECT=None)
                _setenv(MOJO_HIP_COMM_CHUNKS="4")
                check_auto_selection(rank, ws, group)
                continue
            if mode == "timeout":
                _report(rank, mode=mode)
                check_timeout_path(rank, ws, group)
                continue
            if mode == "captured":
                _report(rank, mode=mode)
                check_captured_direct(rank, ws, group)
                continue
            if mode == "tiny":
                _report(rank, mode=mode)
                _setenv(MOJO_HIP_COMM_CHUNKS="4")
                check_fewer_rows_than_ranks(rank, ws, group)
                continue
            if mode.startswith("config4"):
                _report(rank, mode=mode)                 # config4[auto][direct]: "auto" = the pipelines' own chunk count
                _setenv(MOJO_HIP_COMM_CHUNKS=None if "auto" in mode else "4")
                if mode.endswith("direct"):
                    _setenv(MOJO_HIP_COMM_DIRECT="1")
                check_config4_shapes(rank, ws, group)
                if mode.endswith("direct"):
                    from mojo_opset_amd.comm import peer
                    peer.check_all(group)
                continue
            if mode.startswith("chunks"):
                _setenv(MOJO_HIP_COMM_CHUNKS=mode[6:])
            elif mode.startswith("direct"):
                _setenv(MOJO_HIP_COMM_DIRECT="1")
                _setenv(MOJO_HIP_COMM_CHUNKS=mode[6:] or "4")
            _report(rank, mode=mode)
            check_vectors(rank, ws, group)
            check_reference_shapes(rank, ws, group)
            if mode.startswith("direct"):
                from mojo_opset_amd.comm import peer
                assert peer._CACHE, "MOJO_HIP_COMM_DIRECT=1 but no peer exchange was built"
                for ex in peer._CACHE.values():
                    ex.check()                       # no wait timed out
                    _report(rank, direct_exchange={"ranks": ex.ws, "calls": ex.epoch, "uncached_buffer": ex.uncached,
                                                   "capacity_MiB": ex.capacity / 2 ** 20})
        dist.barrier()
    except Exception:
        traceback.print_exc()
        rc = 1
    finally:
        try:
            dist.destroy_process_group()
        except Exception:
            pass
    # leave without running interpreter teardown: a peer that failed must not leave this rank waiting in a collective
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(rc)


if __name__ == "__main__":
    main()
