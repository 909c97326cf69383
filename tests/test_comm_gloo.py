"""Multi-process CPU tests (gloo, world sizes 2, 4 and 8) of the GEMM + collective pipelines (`mojo_opset_amd.comm`): the
same orchestration the hip backend runs over RCCL, driven here by a torch GEMM engine, checked against
(1) the reference's per-rank vectors (tests/golden/compute_with_comm.pt, captured from the reference running
over gloo at 2, 4 and 8 ranks — its own harness is built for 8, `tests/dist_common.py:38-81`) and (2) the oracle classes
running in the same processes (shapes / seeds of `tests/accuracy/operators/test_compute_with_comm.py:95-247`)."""
import os
import socket
import traceback

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden


class TorchEngine:
    """Test-only GEMM engine with the `GemmEngine` call signature (row maps included)."""

    @staticmethod
    def out_features(weight, trans_weight):
        return weight.shape[1] if trans_weight else weight.shape[0]

    def __call__(self, x, weight, bias, trans_weight, *, out=None, rows=None, a_map=None, c_map=None):
        rows = x.shape[0] if rows is None else rows
        m = torch.arange(rows)

        def mapped(mp_):
            if mp_ is None:
                return m
            rc, ml, off = mp_
            return (m // rc) * ml + off + m % rc

        y = x[mapped(a_map)] @ (weight if trans_weight else weight.t())
        if bias is not None:
            y = y + bias
        if out is None:
            assert c_map is None
            return y
        out[mapped(c_map)] = y
        return out


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, ws, port, fn, args, errq):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.set_num_threads(max(1, 8 // ws))                # 8 cores here: ws ranks x all-core teams would thrash
        dist.init_process_group("gloo", rank=rank, world_size=ws)
        fn(rank, ws, *args)
        dist.barrier()
    except Exception:
        errq.put((rank, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def run_dist(fn, *args, ws=2):
    ctx = mp.get_context("spawn")
    errq = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, ws, _free_port_once(), fn, args, errq)) for r in range(ws)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    errs = []
    while not errq.empty():
        errs.append(errq.get())
    assert not errs, "\n".join(f"[rank {r}]\n{t}" for r, t in errs)
    assert all(p.exitcode == 0 for p in procs)


_PORT = {}


def _free_port_once():
    if "p" not in _PORT:
        _PORT["p"] = _free_port()
    return _PORT["p"]


@pytest.fixture(autouse=True)
def _fresh_port():
    _PORT.clear()
    yield


# ---- worker bodies (module level: spawn pickles them by name) ---------------------------------------------
def _check_vectors(rank, ws, cases):
    from mojo_opset_amd import comm

    eng = TorchEngine()
    group = dist.group.WORLD
    seen = 0
    for case in cases:
        if len(case["ranks"]) != ws:
            continue
        seen += 1
        me = case["ranks"][rank]
        x, w, want = me["x"], me["w"], me["out"]
        kw = case["ctor_kwargs"]
        if case["op"] == "MojoGemmAllReduce":
            got = comm.gemm_all_reduce(eng, x, w, None, True, group)
        elif case["op"] == "MojoAllGatherGemm":
            got = comm.all_gather_gemm(eng, x, w, None, True, group, kw["gather_dim"])
        elif case["op"] == "MojoGemmReduceScatter":
            got = comm.gemm_reduce_scatter(eng, x, w, None, True, group, kw["scatter_dim"])
        else:
            got = comm.gemm_all2all(eng, x, w, None, True, group, kw["scatter_dim"], kw["gather_dim"])
        tol = 5e-3 if x.dtype != torch.float32 else 1e-4            # reference bounds: test_compute_with_comm.py:124,164,247
        if ws > 2 and x.dtype != torch.float32 and case["op"] in ("MojoGemmAllReduce", "MojoGemmReduceScatter"):
            # More than two 16-bit partials: the golden adds storage-type values in the COLLECTIVE LIBRARY's order (gloo's
            # ring here, RCCL's on a node), rounding after every addition; any other association differs by up to one unit in
            # the last place of a partial sum per addition.  One bf16 ulp is 0.4-0.8 % — above the reference's 5e-3 — so the
            # bound is stated in those units: |got - want| <= (ws - 1) ulps at the magnitude sum_r |y_r| of the element.
            ys = torch.stack([(c["x"].float() @ c["w"].float()).to(x.dtype).float() for c in case["ranks"]])
            mag = ys.abs().sum(0)
            if case["op"] == "MojoGemmReduceScatter":
                mag = mag.chunk(ws, dim=0)[rank]
            ulp = torch.pow(2.0, torch.floor(torch.log2(mag.clamp_min(2.0 ** -14))) - 7)
            excess = (got.float() - want.float()).abs() - (ws - 1) * ulp
            assert float(excess.max()) <= 0, f"{case['name']}: {float(excess.max())} above (ws - 1) bf16 ulps"
            continue
        torch.testing.assert_close(got.float(), want.float(), atol=tol, rtol=tol, msg=lambda m: f"{case['name']}: {m}")
    assert seen == 5, f"expected the five operators' vectors at world size {ws}, found {seen}"


def _check_against_oracle(rank, ws, chunks):
    import mojo_opset_amd as mo
    import oracle  # noqa: F401
    from mojo_opset_amd import comm

    from mojo_opset_amd import switches

    if chunks:
        os.environ["MOJO_HIP_COMM_CHUNKS"] = str(chunks)
    else:
        os.environ.pop("MOJO_HIP_COMM_CHUNKS", None)                   # the payload- and world-size-aware rule
    switches.reload()
    eng = TorchEngine()
    group = dist.group.WORLD
    torch.manual_seed(42 + rank)
    m, k, n = (1024 if ws <= 2 else 512) * ws, 96, 80
    for trans in (True, False):
        for with_bias in (False, True):
            x = torch.randn(m, k)
            w = torch.randn(k, n) * 0.1 if trans else torch.randn(n, k) * 0.1
            b = torch.randn(n) if with_bias else None
            ref = mo.MojoGemmAllReduce.get_backend_impl("torch")(w, b, trans)(x)
            got = comm.gemm_all_reduce(eng, x, w, b, trans, group)
            torch.testing.assert_close(got, torch.as_tensor(ref), atol=1e-4, rtol=1e-4)
            ref = mo.MojoGemmReduceScatter.get_backend_impl("torch")(w, b, trans, scatter_dim=0)(x)
            got = comm.gemm_reduce_scatter(eng, x, w, b, trans, group, 0)
            torch.testing.assert_close(got, ref, atol=1e-4, rtol=1e-4)
            ref = mo.MojoAllGatherGemm.get_backend_impl("torch")(w, b, trans, gather_dim=0)(x)
            got = comm.all_gather_gemm(eng, x, w, b, trans, group, 0)
            torch.testing.assert_close(got, torch.as_tensor(ref), atol=1e-4, rtol=1e-4)
            for sd, gd in ((0, 1), (0, 0), (1, 0)):
                ref = mo.MojoGemmAll2All.get_backend_impl("torch")(w, b, trans, scatter_dim=sd, gather_dim=gd)(x)
                got = comm.gemm_all2all(eng, x, w, b, trans, group, sd, gd)
                torch.testing.assert_close(got, ref, atol=1e-4, rtol=1e-4)
    # row counts that leave ranks unequal or empty shares of a chunk (ws - 1, ws, 8 ws + 3 rows) through the all-reduce; row
    # counts the world size does not divide are a ValueError of the scattering operators (golden: chunk() would be ragged)
    w = torch.randn(k, n) * 0.1
    for rows in (ws - 1, ws, 8 * ws + 3):
        x = torch.randn(rows, k)
        ref = mo.MojoGemmAllReduce.get_backend_impl("torch")(w, None, True)(x)
        torch.testing.assert_close(comm.gemm_all_reduce(eng, x, w, None, True, group), torch.as_tensor(ref), atol=1e-4, rtol=1e-4)
        if rows % ws:
            for bad in (lambda: comm.gemm_reduce_scatter(eng, x, w, None, True, group, 0),
                        lambda: comm.gemm_all2all(eng, x, w, None, True, group, 0, 1)):
                try:
                    bad()
                except ValueError:
                    pass
                else:
                    raise AssertionError(f"{rows} rows over {ws} ranks must raise ValueError")
        else:
            ref = mo.MojoGemmReduceScatter.get_backend_impl("torch")(w, None, True, scatter_dim=0)(x)
            torch.testing.assert_close(comm.gemm_reduce_scatter(eng, x, w, None, True, group, 0), ref, atol=1e-4, rtol=1e-4)
        xs = torch.randn(max(rows // ws, 1), k)                       # all-gather: any shard height
        ref = mo.MojoAllGatherGemm.get_backend_impl("torch")(w, None, True, gather_dim=0)(xs)
        torch.testing.assert_close(comm.all_gather_gemm(eng, xs, w, None, True, group, 0), torch.as_tensor(ref), atol=1e-4, rtol=1e-4)
    # 3-D input, scatter along a non-leading dimension (fallback path)
    x3 = torch.randn(4, 6 * ws, k)
    w = torch.randn(k, n) * 0.1
    ref = mo.MojoGemmReduceScatter.get_backend_impl("torch")(w, None, True, scatter_dim=1)(x3)
    got = comm.gemm_reduce_scatter(eng, x3, w, None, True, group, 1)
    torch.testing.assert_close(got, ref, atol=1e-4, rtol=1e-4)
    ref = mo.MojoAllGatherGemm.get_backend_impl("torch")(w, None, True, gather_dim=1)(x3)
    got = comm.all_gather_gemm(eng, x3, w, None, True, group, 1)
    torch.testing.assert_close(got, torch.as_tensor(ref), atol=1e-4, rtol=1e-4)
    # 3-D input through the all-to-all: scatter along the leading dimension (the row-chunk pipeline; gather on every
    # dimension) and along an inner one (whole-product fallback)
    x3 = torch.randn(8 * ws, 160, k)
    for sd, gd in ((0, 1), (0, 2), (0, 0), (1, 0), (1, 2)):
        ref = mo.MojoGemmAll2All.get_backend_impl("torch")(w, None, True, scatter_dim=sd, gather_dim=gd)(x3)
        got = comm.gemm_all2all(eng, x3, w, None, True, group, sd, gd)
        assert got.shape == ref.shape
        torch.testing.assert_close(got, ref, atol=1e-4, rtol=1e-4)


def _check_chunks_in_flight(rank, ws):
    """The wait ordering of the asynchronous pipelines (comm/pipelines.py gemm_all_reduce / gemm_reduce_scatter /
    all_gather_gemm): every chunk's collective is ISSUED (async_op=True) before the first one is waited for, the GEMM of
    chunk c + 1 is enqueued while chunk c's collective is in flight, and the handles are waited for in issue order."""
    from mojo_opset_amd import comm
    from mojo_opset_amd.comm import pipelines

    from mojo_opset_amd import switches

    os.environ["MOJO_HIP_COMM_CHUNKS"] = "4"
    switches.reload()
    log = []

    class Spy:
        def __init__(self, work, tag):
            self.work, self.tag = work, tag

        def wait(self):
            log.append(("wait", self.tag))
            return self.work.wait()

    real = {n: getattr(dist, n) for n in ("all_reduce", "reduce_scatter_tensor", "all_gather_into_tensor", "all_gather")}
    counter = {"n": 0}

    def wrap(name):
        def call(*args, **kw):
            work = real[name](*args, **kw)
            if kw.get("async_op"):
                counter["n"] += 1
                log.append(("issue", counter["n"]))
                return Spy(work, counter["n"])
            return work
        return call

    class LoggingEngine(TorchEngine):
        def __call__(self, *a, **kw):
            log.append(("gemm", None))
            return super().__call__(*a, **kw)

    for name in real:
        setattr(pipelines.dist, name, wrap(name))
    try:
        eng = LoggingEngine()
        group = dist.group.WORLD
        torch.manual_seed(3 + rank)
        m, k, n = 2048 * ws, 64, 48
        x, w = torch.randn(m, k), torch.randn(k, n) * 0.1

        def run(fn, *extra):
            log.clear()
            counter["n"] = 0
            out = fn(eng, x, w, None, True, group, *extra)
            issues = [i for i, e in enumerate(log) if e[0] == "issue"]
            waits = [i for i, e in enumerate(log) if e[0] == "wait"]
            gemms = [i for i, e in enumerate(log) if e[0] == "gemm"]
            assert len(issues) == len(waits) >= 3, log
            assert [log[i][1] for i in waits] == sorted(log[i][1] for i in waits), log      # waited for in issue order
            return out, issues, waits, gemms

        out, issues, waits, gemms = run(comm.gemm_all_reduce)
        assert max(issues) < min(waits), "all-reduce: a chunk was waited for before the last one was issued"
        assert any(issues[0] < g < issues[-1] for g in gemms), "no GEMM was enqueued while a collective was in flight"
        want = x @ w
        dist.all_reduce(want)
        torch.testing.assert_close(out, want, atol=1e-4, rtol=1e-4)
        out, issues, waits, gemms = run(comm.gemm_reduce_scatter, 0)
        assert max(issues) < min(waits) and any(issues[0] < g < issues[-1] for g in gemms)
        out, issues, waits, gemms = run(comm.all_gather_gemm, 0)
        assert max(issues) < min(waits), "all-gather: every gather is issued up front"
        assert any(waits[0] < g < waits[-1] for g in gemms), "the GEMM of chunk c must run while later gathers are in flight"
        out, issues, waits, gemms = run(comm.gemm_all2all, 0, 1)                 # row-blocked scatter: chunked like the others
        assert max(issues) < min(waits), "all-to-all: a chunk was waited for before the last one was issued"
        assert any(issues[0] < g < issues[-1] for g in gemms), "all-to-all: no GEMM was enqueued while an exchange was in flight"
        y = x @ w
        mine = [torch.empty(m // ws, n) for _ in range(ws)]
        everyone = [torch.empty_like(y) for _ in range(ws)]
        real["all_gather"](everyone, y)
        want = torch.cat([everyone[s].chunk(ws, 0)[rank] for s in range(ws)], dim=1)
        torch.testing.assert_close(out, want, atol=1e-4, rtol=1e-4)
    finally:
        for name, fn in real.items():
            setattr(pipelines.dist, name, fn)


@pytest.mark.parametrize("ws", [2, 4])
def test_pipelines_keep_several_chunks_in_flight(ws):
    run_dist(_check_chunks_in_flight, ws=ws)


@pytest.mark.parametrize("ws", [2, 4, 8])
def test_pipelines_reproduce_reference_vectors_over_gloo(ws):
    run_dist(_check_vectors, load_golden("compute_with_comm"), ws=ws)


@pytest.mark.parametrize("ws,chunks", [(2, 1), (2, 3), (4, 1), (4, 4), (4, 0), (8, 1), (8, 4)])
def test_pipelines_match_oracle_over_gloo(ws, chunks):
    """chunks = 0: the chunk count the pipelines choose themselves (payload- and world-size-aware, comm/pipelines.py)."""
    run_dist(_check_against_oracle, chunks, ws=ws)


def test_plan_row_chunks(monkeypatch):
    from mojo_opset_amd.comm import plan_row_chunks
    from mojo_opset_amd.comm.pipelines import chunk_count

    monkeypatch.delenv("MOJO_HIP_COMM_CHUNKS", raising=False)
    assert plan_row_chunks(0) == []
    assert plan_row_chunks(100) == [(0, 100)]
    c = plan_row_chunks(4096, 8192, 8)
    assert c[0][0] == 0 and c[-1][1] == 4096 and all(a[1] == b[0] for a, b in zip(c, c[1:]))
    assert all((hi - lo) % 256 == 0 for lo, hi in c[:-1])
    # the count follows the payload and the world size (it was the constant 4): a chunk's GEMM is at least one full round of
    # the chip's 256 tile slots, and a peer's share of a chunk is at least 512 KiB
    assert chunk_count(4096, 8192, 8) == 2 and chunk_count(8192, 8192, 8) == 4 and chunk_count(1024, 8192, 8) == 1     # all-reduce, config 4
    assert chunk_count(512, 8192, 8, gemm_rows_per_row=8) == 2                 # reduce-scatter at tp 8, M 4096: 2 x (8 x 256 rows)
    assert chunk_count(2048, 8192, 2, gemm_rows_per_row=2) == 2                # ... at tp 2
    assert chunk_count(65536, 8192, 8) == 8                                    # never more than MAX_CHUNKS
    assert chunk_count(4096, 256, 8) == 1                                      # a narrow product: one round needs all the rows
    assert chunk_count(512, 1280, 8, 8, 2, link_cols=8192) == 1                # all-gather feeding a narrow shard
    monkeypatch.setenv("MOJO_HIP_COMM_CHUNKS", "4")                            # forced (the fixture reloads the switches)
    assert len(plan_row_chunks(4096, 8192, 8)) == 4 and len(plan_row_chunks(1024, 8192, 8)) == 2


def test_identity_without_process_group():
    from mojo_opset_amd import comm

    eng = TorchEngine()
    x, w = torch.randn(10, 8), torch.randn(8, 6)
    for got in (comm.gemm_all_reduce(eng, x, w, None, True, None), comm.gemm_reduce_scatter(eng, x, w, None, True, None, 0),
                comm.all_gather_gemm(eng, x, w, None, True, None, 0), comm.gemm_all2all(eng, x, w, None, True, None, 0, 1)):
        torch.testing.assert_close(got, x @ w)


def _check_expert_parallel_moe(rank, ws, dp_input):
    """`MojoMoE` with ep_size = world size (core/operators/moe.py:104-128): each rank holds a slice of the experts, the
    summed partial outputs equal the single-process layer on all tokens."""
    import mojo_opset_amd as mo
    import oracle  # noqa: F401

    cls = mo.MojoMoE.get_backend_impl("torch", strict=True)
    experts, k, hidden, inter, tokens = 6, 2, 64, 48, 10 * ws
    torch.manual_seed(7)
    full = cls(num_experts=experts, top_k=k, hidden_size=hidden, intermediate_size=inter)
    for p in full.parameters():
        torch.nn.init.normal_(p, std=0.2)
    x = torch.rand(tokens, hidden)
    want = full(x)
    part = cls(num_experts=experts, top_k=k, hidden_size=hidden, intermediate_size=inter, ep_size=ws, ep_rank=rank,
               ep_group=dist.group.WORLD, dp_input=dp_input)
    assert part.num_experts_local == experts // ws and part.experts.up_proj_weight.shape[0] == experts // ws
    with torch.no_grad():
        part.gating.gate_weight.copy_(full.gating.gate_weight)
        part.experts.up_proj_weight.copy_(full.experts.up_proj_weight[part.ep_start:part.ep_end])
        part.experts.down_proj_weight.copy_(full.experts.down_proj_weight[part.ep_start:part.ep_end])
    if dp_input:
        per = tokens // ws
        got = part(x[rank * per:(rank + 1) * per].clone())
        torch.testing.assert_close(got, want[rank * per:(rank + 1) * per], atol=1e-5, rtol=1e-5)
    else:
        torch.testing.assert_close(part(x.clone()), want, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("dp_input", [False, True])
def test_moe_expert_parallel_wiring(dp_input):
    run_dist(_check_expert_parallel_moe, dp_input)


def _check_selector_host_logic(rank, ws):
    """comm/select.py without a GPU: forced switches, payload buckets, and CPU tensors never take the direct exchange."""
    from mojo_opset_amd.comm import select

    group = dist.group.WORLD
    x = torch.zeros(4, 4)
    calls = []
    os.environ.pop("MOJO_HIP_COMM_DIRECT", None)
    assert select.forced() is None
    assert select.choose(group, "gemm_all_reduce", 1 << 22, x, lambda: calls.append("d"), lambda: calls.append("r")) == "rccl"
    assert select.choose(None, "gemm_all_reduce", 1 << 22, x, None, None) == "rccl"
    assert not calls and not select.report()                  # nothing timed, nothing cached for host tensors
    from mojo_opset_amd import switches

    os.environ["MOJO_HIP_COMM_DIRECT"] = "1"
    assert select.forced() is None                            # switches are LATCHED at first use ...
    switches.reload()
    assert select.forced() == "direct"                        # ... and re-read on reload()
    os.environ["MOJO_HIP_COMM_DIRECT"] = "0"
    switches.reload()
    assert select.forced() == "rccl"
    os.environ.pop("MOJO_HIP_COMM_DIRECT", None)
    switches.reload()
    assert select.forced() is None
    assert select.bucket(1) == 1 << 20 and select.bucket((1 << 20) + 1) == 1 << 21 and select.bucket(64 << 20) == 64 << 20
    assert select._agree_max(group, "cpu", float(rank), 5.0 - rank) == [float(ws - 1), 5.0]
    # the self-test's row-copy engine honours both row maps
    eng = select._RowCopyEngine(3)
    src = torch.arange(24.0).view(8, 3)
    got = eng(src, None, None, True, rows=4, a_map=(2, 4, 1))           # rows 1, 2, 5, 6
    assert torch.equal(got, src[[1, 2, 5, 6]])
    dst = torch.zeros(8, 3)
    eng(src[:4], None, None, True, out=dst, rows=4, c_map=(2, 4, 1))
    assert torch.equal(dst[[1, 2, 5, 6]], src[:4])
    pat = select._pattern(rank, 64, 16, "cpu", torch.bfloat16)
    assert pat.abs().max() <= 8 and torch.equal(pat, pat.float().round().to(torch.bfloat16))


@pytest.mark.parametrize("ws", [2, 4])
def test_selector_host_logic(ws):
    run_dist(_check_selector_host_logic, ws=ws)
