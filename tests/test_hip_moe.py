"""GPU parity of the MoE routing ops through the C ABI.

Checks follow the reference's own tests (mojo_opset/tests/accuracy/operators/test_moe.py): gating — expert ids exact on
≥ 99.9 % of the slots and gates within 1e-2 (:141-146); dispatch — counts exact, rows exact, every bucket verified as an
unordered set (:186-210); combine — the reference asks for `mixed_tol`, this build is bit-identical (fp32 adds in the
golden's scatter order); experts — `mixed_tol` (:242-247)."""
import pytest
import torch

from conftest import load_golden
from hip_utils import DEV, hip_cls, last_launch, launches_of, run_hip_case, to_cpu, torch_cls
from mojo_opset_amd.core import check_tol_diff

pytestmark = pytest.mark.gpu

CASES = load_golden("moe")


def _of(op):
    return [pytest.param(c, id=f"{op}-{i}") for i, c in enumerate(c for c in CASES if c["op"] == op)]


def _check_buckets(sorted_hidden, per_expert, sorted_gates, token_indices, hidden, gates, ids):
    """The reference's bucket-as-a-set check (:186-210), plus: every routing slot appears exactly once."""
    n = ids.numel()
    assert int(per_expert.sum()) == n
    tok = token_indices.to(torch.int64)
    assert torch.equal(sorted_hidden, hidden[tok])
    ends = torch.cumsum(per_expert.to(torch.int64), 0).tolist()
    starts = [0] + ends[:-1]
    seen = torch.zeros(ids.shape, dtype=torch.int32)
    for e, (s, t) in enumerate(zip(starts, ends)):
        if s == t:
            continue
        rows = tok[s:t]
        match = ids[rows] == e
        assert bool(match.any(dim=-1).all())
        slot = match.to(torch.int64).argmax(dim=-1)
        assert torch.equal(sorted_gates[s:t], gates[rows, slot].unsqueeze(-1))
        seen[rows, slot] += 1
    assert bool((seen == 1).all())


@pytest.mark.parametrize("case", _of("MojoMoEGating"))
def test_gating_vectors(case):
    idx, gates = to_cpu(run_hip_case(case))
    want_idx, want_gates = case["out"]
    assert idx.dtype == torch.int32 and gates.dtype == torch.float32
    assert float((idx == want_idx).float().mean()) >= 0.999
    same = idx == want_idx
    torch.testing.assert_close(gates[same], want_gates[same], atol=1e-5, rtol=1e-4)
    torch.testing.assert_close(gates.sum(-1), torch.ones(gates.shape[0]), atol=1e-5, rtol=0)


@pytest.mark.parametrize("experts,k,hidden,tokens", [(16, 4, 1024, 64), (32, 8, 1024, 128), (64, 8, 1024, 256),
                                                     (64, 8, 1024, 1024), (8, 2, 4096, 8192), (384, 8, 3584, 128),
                                                     (256, 8, 7168, 77), (5, 5, 200, 33),
                                                     # many experts x many tokens: the MFMA route (x @ w_hi + x @ w_lo)
                                                     (384, 8, 3584, 2048), (256, 8, 7168, 600), (33, 3, 128, 515)])
def test_gating_reference_space(experts, k, hidden, tokens):
    torch.manual_seed(0)
    ref = torch_cls("MojoMoEGating")(hidden_size=hidden, num_experts=experts, top_k=k)
    torch.nn.init.normal_(ref.gate_weight, std=0.02)
    op = hip_cls("MojoMoEGating")(hidden_size=hidden, num_experts=experts, top_k=k).to(DEV)
    op.load_state_dict(ref.state_dict())
    assert op.gate_weight.dtype == torch.float32
    x = torch.rand(tokens, hidden, dtype=torch.bfloat16)
    op.forward_diff_with(ref, x.to(DEV), atol=(0, 1e-2), rtol=(0, 1e-2), ptol=(0.999, 1.0), ref_device="cpu")


@pytest.mark.parametrize("case", _of("MojoMoEDispatch"))
def test_dispatch_vectors(case):
    sorted_hidden, per_expert, sorted_gates, token_indices = to_cpu(run_hip_case(case))
    hidden, gates, ids = case["args"]
    want_hidden, want_counts, want_gates, want_tok = case["out"]
    assert torch.equal(per_expert, want_counts) and per_expert.dtype == torch.int32
    assert sorted_gates.shape == want_gates.shape and token_indices.dtype == torch.int32
    _check_buckets(sorted_hidden, per_expert, sorted_gates, token_indices, hidden, gates, ids)
    # this build's order inside a bucket is the flat slot order (stable): compare with a stable sort of the same ids
    order = torch.sort(ids.flatten().to(torch.int64), stable=True).indices
    k = ids.shape[1] if ids.numel() else 1
    assert torch.equal(token_indices.to(torch.int64), order // k)


@pytest.mark.parametrize("experts,k,hidden,tokens", [(16, 4, 1024, 64), (32, 8, 1024, 128), (64, 8, 1024, 256),
                                                     (384, 8, 3584, 128), (8, 2, 4096, 8192), (8, 2, 100, 1000), (3, 1, 8, 1)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_dispatch_reference_space(experts, k, hidden, tokens, dtype):
    torch.manual_seed(0)
    x = torch.rand(tokens, hidden, dtype=dtype)
    probs = torch.softmax(torch.randn(tokens, experts), dim=-1)
    gates, ids = torch.topk(probs, k, dim=-1)
    gates = (gates / gates.sum(-1, keepdim=True)).contiguous()
    ids = ids.to(torch.int32).contiguous()
    op = hip_cls("MojoMoEDispatch")(num_experts=experts)
    out = to_cpu(op(x.to(DEV), gates.to(DEV), ids.to(DEV)))
    want = torch_cls("MojoMoEDispatch")(num_experts=experts)(x, gates, ids)
    assert torch.equal(out[1], want[1])
    _check_buckets(*out, x, gates, ids)
    again = to_cpu(op(x.to(DEV), gates.to(DEV), ids.to(DEV)))            # deterministic: identical on a second run
    assert all(torch.equal(a, b) for a, b in zip(out, again))


def test_dispatch_rejects_wrong_dtypes():
    op = hip_cls("MojoMoEDispatch")(num_experts=4)
    x = torch.rand(4, 8, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(AssertionError):
        op(x, torch.rand(4, 2, device=DEV, dtype=torch.bfloat16), torch.zeros(4, 2, dtype=torch.int32, device=DEV))
    with pytest.raises(AssertionError):
        op(x, torch.rand(4, 2, device=DEV), torch.zeros(4, 2, dtype=torch.int64, device=DEV))


@pytest.mark.parametrize("case", _of("MojoMoECombine"))
def test_combine_vectors_bit_exact(case):
    out = to_cpu(run_hip_case(case))
    assert out.dtype == case["out"].dtype and torch.equal(out, case["out"])


@pytest.mark.parametrize("tokens,k,hidden", [(64, 4, 1024), (128, 8, 1024), (256, 8, 1024), (128, 8, 3584), (8192, 2, 4096), (7, 3, 50)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("by_gates", [True, False])
def test_combine_reference_space_bit_exact(tokens, k, hidden, dtype, by_gates):
    torch.manual_seed(0)
    n = tokens * k
    perm = torch.randperm(n)
    rows = torch.randn(n, hidden, dtype=dtype)[perm].contiguous()
    gates = torch.rand(n, 1)[perm].contiguous()
    tok = torch.arange(tokens, dtype=torch.int32).unsqueeze(1).expand(-1, k).reshape(-1)[perm].contiguous()
    buf = torch.zeros(tokens, hidden, dtype=dtype)
    want = torch_cls("MojoMoECombine")(multiply_by_gates=by_gates)(buf, rows, gates, tok)
    got = hip_cls("MojoMoECombine")(multiply_by_gates=by_gates)(buf.to(DEV), rows.to(DEV), gates.to(DEV), tok.to(DEV))
    assert torch.equal(to_cpu(got), want)


@pytest.mark.parametrize("tokens,k,hidden", [(1, 1, 64), (1000, 3, 64), (15360, 2, 64), (15361, 2, 64), (20000, 1, 32)])
@pytest.mark.parametrize("plan", ["1", "0"])
def test_combine_row_list_plan_routes_agree_with_the_oracle(tokens, k, hidden, plan, monkeypatch):
    """The row lists of the combine are built by one single-block kernel (counters in LDS, <= 15 360 tokens) or by the
    five-launch plan (more tokens, or MOJO_HIP_MOE_PLAN=0): both bit-equal to the golden, skewed routing included."""
    monkeypatch.setenv("MOJO_HIP_MOE_PLAN", plan)
    torch.manual_seed(tokens + k)
    n = tokens * k
    rows = torch.randn(n, hidden, dtype=torch.bfloat16)
    gates = torch.rand(n, 1)
    tok = torch.randint(0, tokens, (n,), dtype=torch.int32)             # some tokens get many rows, some none
    tok[: min(n // 3, 900)] = tok[0]                                    # one token owns many rows (<= 1024: the length the combine sorts in LDS)
    tok = tok[torch.randperm(n)].contiguous()
    buf = torch.zeros(tokens, hidden, dtype=torch.bfloat16)
    want = torch_cls("MojoMoECombine")()(buf, rows, gates, tok)
    got = hip_cls("MojoMoECombine")()(buf.to(DEV), rows.to(DEV), gates.to(DEV), tok.to(DEV))
    assert torch.equal(to_cpu(got), want)


@pytest.mark.parametrize("experts,k,tokens", [(1, 1, 5), (2, 2, 40000), (257, 4, 9000), (4096, 8, 3000), (64, 8, 20000)])
def test_dispatch_scan_many_blocks_and_experts(experts, k, tokens):
    """The counting sort's scan kernel (one wave per expert over the blocks' histograms, then the experts' totals): more
    than 64 blocks per expert, an odd expert count, the maximum expert count."""
    torch.manual_seed(experts)
    x = torch.rand(tokens, 16, dtype=torch.bfloat16)
    ids = torch.stack([torch.randperm(experts)[:k] for _ in range(64)]).repeat((tokens + 63) // 64, 1)[:tokens].to(torch.int32).contiguous()
    ids = ids[torch.randperm(tokens)].contiguous()
    gates = torch.rand(tokens, k)
    op = hip_cls("MojoMoEDispatch")(num_experts=experts)
    out = to_cpu(op(x.to(DEV), gates.to(DEV), ids.to(DEV)))
    want = torch_cls("MojoMoEDispatch")(num_experts=experts)(x, gates, ids)
    assert torch.equal(out[1], want[1])
    _check_buckets(*out, x, gates, ids)


def test_combine_tokens_without_rows_are_zero():
    rows = torch.randn(6, 64, dtype=torch.bfloat16)
    tok = torch.tensor([5, 5, 0, 5, 0, 9], dtype=torch.int32)
    gates = torch.rand(6, 1)
    buf = torch.full((12, 64), 7.0, dtype=torch.bfloat16)            # the buffer's CONTENT is ignored (golden: zeros_like)
    want = torch_cls("MojoMoECombine")()(buf, rows, gates, tok)
    got = to_cpu(hip_cls("MojoMoECombine")()(buf.to(DEV), rows.to(DEV), gates.to(DEV), tok.to(DEV)))
    assert torch.equal(got, want) and float(got[1].abs().sum()) == 0.0


@pytest.mark.parametrize("case", _of("MojoExperts"))
def test_experts_vectors(case):
    out = to_cpu(run_hip_case(case))
    check_tol_diff(out, case["out"], mixed_tol=True)


@pytest.mark.parametrize("experts,hidden,inter,counts", [
    (4, 256, 512, [3, 0, 5, 4]), (8, 512, 1024, [2, 1, 0, 3, 4, 0, 5, 2]), (384, 3584, 64, [2, 1, 0, 3] + [0] * 380),
    (8, 1024, 2048, [300, 17, 0, 512, 256, 1, 90, 64]),
])
def test_experts_reference_space(experts, hidden, inter, counts):
    torch.manual_seed(0)
    ref = torch_cls("MojoExperts")(num_experts=experts, hidden_size=hidden, intermediate_size=inter)
    for p in ref.parameters():
        torch.nn.init.normal_(p, std=0.02)
    ref = ref.to(torch.bfloat16)
    op = hip_cls("MojoExperts")(num_experts=experts, hidden_size=hidden, intermediate_size=inter).to(torch.bfloat16).to(DEV)
    op.load_state_dict(ref.state_dict())
    x = torch.rand(sum(counts), hidden, dtype=torch.bfloat16)
    cnt = torch.tensor(counts, dtype=torch.int32)
    op.forward_diff_with(ref, x.to(DEV), cnt.to(DEV), mixed_tol=True, ref_device="cpu")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("experts,hidden,inter,counts", [
    (4, 256, 512, [3, 0, 5, 4]), (8, 1024, 2048, [300, 17, 0, 512, 256, 1, 90, 64]), (3, 512, 1280, [700, 0, 33]),
])
def test_experts_fused_swiglu_epilogue_is_bit_identical_to_the_two_kernel_path(experts, hidden, inter, counts, dtype, monkeypatch):
    """The first projection applies SwiGLU to its accumulators (mojo_hip_group_gemm_swiglu) with the rounding points of
    GroupGemm -> SwiGLU; MOJO_HIP_EXPERTS_FUSED=0 runs the two kernels.  Same bits, and the fused entry point really ran."""
    torch.manual_seed(1)
    op = hip_cls("MojoExperts")(num_experts=experts, hidden_size=hidden, intermediate_size=inter).to(dtype).to(DEV)
    for p in op.parameters():
        torch.nn.init.normal_(p, std=0.05)
    x = torch.randn(sum(counts), hidden, dtype=dtype, device=DEV)
    cnt = torch.tensor(counts, dtype=torch.int32, device=DEV)
    act = torch.empty(x.shape[0], inter, dtype=dtype, device=DEV)
    monkeypatch.setenv("MOJO_HIP_GEMM_SKINNY", str(31 & ~2))     # (no ragged streaming form: a decode-sized case would otherwise take it)
    monkeypatch.setenv("MOJO_HIP_GEMM_TILE128", "0")             # (nor the 128-row tiles, for which the fused entry point defers to the two-kernel path)
    assert op._fused_up_swiglu(x, op.up_proj_weight.detach(), cnt, act, inter)
    fused = op(x, cnt)
    monkeypatch.setenv("MOJO_HIP_EXPERTS_FUSED", "0")
    assert not op._fused_up_swiglu(x, op.up_proj_weight.detach(), cnt, act, inter)       # the switch is seen: the two-kernel path runs
    plain = op(x, cnt)
    assert torch.equal(fused, plain)
    monkeypatch.delenv("MOJO_HIP_GEMM_SKINNY")
    monkeypatch.delenv("MOJO_HIP_EXPERTS_FUSED")
    monkeypatch.delenv("MOJO_HIP_GEMM_TILE128")
    assert torch.equal(op(x, cnt), fused)                        # whatever the library picks by itself (128-row tiles included): the same bits
    monkeypatch.setenv("MOJO_HIP_GEMM_TILE128", "1")
    assert torch.equal(op(x, cnt), fused)                        # ... and with the 128-row tiles forced


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("experts,hidden,inter,counts", [
    (4, 256, 512, [3, 0, 5, 4]), (8, 1024, 1024, [64, 64, 64, 64, 64, 64, 64, 64]), (6, 384, 640, [0, 0, 130, 1, 0, 65]),
    (64, 512, 256, None), (2, 128, 64, [1, 0]),
])
def test_experts_streaming_form_for_ragged_decode_groups_is_bit_identical(experts, hidden, inter, counts, dtype, monkeypatch):
    """At most 64 rows per expert on average (a decode step): both projections run the 64-row streaming grouped GEMM on
    ragged groups (gemm_skinny_kernel<.., RAGGED>) — empty groups, groups of more than 64 rows (several blocks), a group
    that ends on a block boundary; same bits as the tile kernels (256 or 128 rows, whichever the library picks), and against the oracle."""
    torch.manual_seed(3)
    if counts is None:
        counts = torch.randint(0, 17, (experts,)).tolist()
    ref = torch_cls("MojoExperts")(num_experts=experts, hidden_size=hidden, intermediate_size=inter)
    for p in ref.parameters():
        torch.nn.init.normal_(p, std=0.02)
    ref = ref.to(dtype)
    op = hip_cls("MojoExperts")(num_experts=experts, hidden_size=hidden, intermediate_size=inter).to(dtype).to(DEV)
    op.load_state_dict(ref.state_dict())
    assert sum(counts) <= 64 * experts
    x = torch.rand(sum(counts), hidden, dtype=dtype, device=DEV)          # (the data of the reference's own experts test)
    cnt = torch.tensor(counts, dtype=torch.int32, device=DEV)
    streamed = op(x, cnt)
    hist = launches_of(lambda: op(x, cnt))
    assert "gemm_skinny:ragged" in hist, hist                    # (a projection whose K is no multiple of 128 keeps the tile kernel)
    monkeypatch.setenv("MOJO_HIP_GEMM_SKINNY", str(31 & ~2))     # every decode-sized form but the ragged one
    tiled = op(x, cnt)
    hist = launches_of(lambda: op(x, cnt))
    assert "gemm_skinny:ragged" not in hist and ("gemm256:" in hist or "gemm128:" in hist or "gemm_generic" in hist), hist   # the OTHER kernels really ran
    assert torch.equal(streamed, tiled)
    monkeypatch.delenv("MOJO_HIP_GEMM_SKINNY")
    op.forward_diff_with(ref, x, cnt, mixed_tol=True, ref_device="cpu")


def test_moe_layer_end_to_end_matches_the_oracle_chain():
    """gating -> dispatch -> experts -> combine on the device against the same chain of oracle classes on the CPU
    (the composition `MojoMoE.forward` performs, core/operators/moe.py:86-130, ep_size = 1)."""
    torch.manual_seed(1)
    experts, k, hidden, inter, tokens = 8, 2, 512, 1024, 300
    refs = {n: torch_cls(n) for n in ("MojoMoEGating", "MojoMoEDispatch", "MojoExperts", "MojoMoECombine")}
    g_ref = refs["MojoMoEGating"](hidden_size=hidden, num_experts=experts, top_k=k)
    e_ref = refs["MojoExperts"](num_experts=experts, hidden_size=hidden, intermediate_size=inter)
    for p in list(g_ref.parameters()) + list(e_ref.parameters()):
        torch.nn.init.normal_(p, std=0.05)
    e_ref = e_ref.to(torch.bfloat16)
    g_hip = hip_cls("MojoMoEGating")(hidden_size=hidden, num_experts=experts, top_k=k).to(DEV)
    g_hip.load_state_dict(g_ref.state_dict())
    e_hip = hip_cls("MojoExperts")(num_experts=experts, hidden_size=hidden, intermediate_size=inter).to(torch.bfloat16).to(DEV)
    e_hip.load_state_dict(e_ref.state_dict())
    x = torch.rand(tokens, hidden, dtype=torch.bfloat16)

    def chain(gating, dispatch, experts_op, combine, inp):
        idx, gates = gating(inp)
        sh, counts, sg, tok = dispatch(inp, gates, idx)
        return combine(torch.zeros_like(inp), experts_op(sh, counts), sg, tok)

    want = chain(g_ref, refs["MojoMoEDispatch"](num_experts=experts), e_ref, refs["MojoMoECombine"](), x)
    got = chain(g_hip, hip_cls("MojoMoEDispatch")(num_experts=experts), e_hip, hip_cls("MojoMoECombine")(), x.to(DEV))
    check_tol_diff(to_cpu(got), want, mixed_tol=True)


@pytest.mark.parametrize("case", [pytest.param(c, id=f"MojoMoE-{i}") for i, c in enumerate(load_golden("moe_layer"))])
def test_moe_layer_vectors(case):
    """The composite operator against the reference's recorded outputs (`mixed_tol`, test_moe.py:103)."""
    out = to_cpu(run_hip_case(case))
    assert out.dtype == case["out"].dtype and out.shape == case["out"].shape
    check_tol_diff(out, case["out"], mixed_tol=True)


@pytest.mark.parametrize("experts,k,hidden,inter,tokens", [(16, 4, 1024, 2048, 64), (32, 8, 1024, 4096, 128),
                                                           (64, 8, 1024, 4096, 256), (64, 8, 1024, 4096, 1024),
                                                           (8, 2, 4096, 1024, 1)])
def test_moe_layer_reference_space(experts, k, hidden, inter, tokens):
    """The reference's own `test_moe` (test_moe.py:57-103): bf16 layer, fp32 router, normal(0.02) parameters."""
    torch.manual_seed(0)
    kw = dict(num_experts=experts, top_k=k, hidden_size=hidden, intermediate_size=inter)
    ref = torch_cls("MojoMoE")(**kw).to(torch.bfloat16)
    op = hip_cls("MojoMoE")(**kw).to(torch.bfloat16).to(DEV)
    assert type(op.gating).__name__ == "HIPMoEGating" and type(op.experts).__name__ == "HIPExperts"
    ref.gating.gate_weight.data = ref.gating.gate_weight.data.float()
    op.gating.gate_weight.data = op.gating.gate_weight.data.float()
    for p in ref.parameters():
        torch.nn.init.normal_(p, std=0.02)
    op.load_state_dict(ref.state_dict())
    x = torch.rand(tokens, hidden, dtype=torch.bfloat16)
    op.forward_diff_with(ref, x.to(DEV), mixed_tol=True, ref_device="cpu")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("k,hidden,tokens", [(2, 4096, 1000), (1, 512, 33), (8, 1024, 70), (3, 2048, 4096)])
def test_gating_eight_expert_kernel_agrees_with_general_kernel(k, hidden, tokens, dtype, monkeypatch):
    """E = 8 takes a dedicated streaming kernel (a lane owns hidden elements, not an expert); MOJO_HIP_GATING=1 runs the
    general one.  Identical expert choice (up to near-ties), gates to 1e-5, and the oracle agrees with both."""
    torch.manual_seed(5)
    op = hip_cls("MojoMoEGating")(hidden_size=hidden, num_experts=8, top_k=k).to(DEV)
    with torch.no_grad():
        op.gate_weight.normal_(std=0.05)
    x = torch.rand(tokens, hidden, dtype=dtype, device=DEV)
    monkeypatch.setenv("MOJO_HIP_GATING", "2")                   # streaming kernels only (the E = 8 specialisation where it applies)
    idx_s, g_s = op(x)
    assert last_launch() == ("moe_gating:e8" if hidden % 512 == 0 and tokens >= 32 else "moe_gating:general"), last_launch()
    monkeypatch.setenv("MOJO_HIP_GATING", "1")                   # (the fixture makes the library re-read its switches)
    idx_g, g_g = op(x)
    assert last_launch() == "moe_gating:general", last_launch()  # the general kernel really ran
    same = idx_s == idx_g
    assert float(same.float().mean()) >= 0.999
    torch.testing.assert_close(g_s[same], g_g[same], atol=1e-5, rtol=1e-4)
    torch.testing.assert_close(g_s.sum(-1), torch.ones(tokens, device=DEV), atol=1e-5, rtol=0)
    ref = torch_cls("MojoMoEGating")(hidden_size=hidden, num_experts=8, top_k=k)
    ref.load_state_dict({k_: v_.cpu() for k_, v_ in op.state_dict().items()})
    want_idx, want_g = ref(x.cpu())
    for idx_d, g_d in ((idx_s, g_s), (idx_g, g_g)):              # the oracle against BOTH kernels
        ok = idx_d.cpu() == want_idx
        assert float(ok.float().mean()) >= 0.999
        torch.testing.assert_close(g_d.cpu()[ok], want_g[ok], atol=1e-5, rtol=1e-4)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gating_mfma_route_agrees_with_vector_route(dtype, monkeypatch):
    """Same inputs through both routes of `mojo_hip_moe_gating`: identical expert choice (up to near-ties), gates to 1e-4."""
    torch.manual_seed(3)
    experts, k, hidden, tokens = 128, 6, 1024, 768
    op = hip_cls("MojoMoEGating")(hidden_size=hidden, num_experts=experts, top_k=k).to(DEV)
    with torch.no_grad():
        op.gate_weight.normal_(std=0.05)
    x = torch.rand(tokens, hidden, dtype=dtype, device=DEV)
    monkeypatch.setenv("MOJO_HIP_GATING", "4")
    idx_m, g_m = op(x)
    assert last_launch() == "moe_gating:mfma", last_launch()
    monkeypatch.setenv("MOJO_HIP_GATING", "1")
    idx_v, g_v = op(x)
    assert last_launch() == "moe_gating:general", last_launch()
    same = idx_m == idx_v
    assert float(same.float().mean()) >= 0.999
    torch.testing.assert_close(g_m[same], g_v[same], atol=1e-4, rtol=1e-3)


@pytest.mark.parametrize("tokens,experts,k,hidden", [(1, 64, 8, 4096), (7, 9, 3, 100), (64, 256, 8, 7168), (200, 384, 8, 3584),
                                                     (256, 1024, 64, 256), (33, 17, 17, 48), (64, 64, 6, 4104)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gating_few_token_kernel_matches_the_oracle(tokens, experts, k, hidden, dtype, monkeypatch):
    """The router of a decode step (<= 256 tokens, more than 8 experts) runs moe_gating_small_kernel + a workgroup-per-token
    select: against the golden (fp32 product) and against the general kernel — odd expert counts, hidden sizes that are no
    multiple of a step, the maximum expert count, fp32 activations."""
    torch.manual_seed(tokens * 131 + experts)
    ref = torch_cls("MojoMoEGating")(hidden_size=hidden, num_experts=experts, top_k=k)
    torch.nn.init.normal_(ref.gate_weight, std=0.05)
    op = hip_cls("MojoMoEGating")(hidden_size=hidden, num_experts=experts, top_k=k).to(DEV)
    op.load_state_dict(ref.state_dict())
    x = torch.rand(tokens, hidden).to(dtype)
    monkeypatch.setenv("MOJO_HIP_GATING", "3")
    idx_s, gate_s = to_cpu(op(x.to(DEV)))
    assert last_launch() == "moe_gating:small", last_launch()
    idx_w, gate_w = ref(x)
    monkeypatch.setenv("MOJO_HIP_GATING", "1")
    idx_g, gate_g = to_cpu(op(x.to(DEV)))
    assert last_launch() == "moe_gating:general", last_launch()
    # selections can only differ where two probabilities are closer than fp32 summation-order noise
    for idx_o, gate_o in ((idx_w.to(torch.int32), gate_w), (idx_g, gate_g)):
        same = (idx_s == idx_o).all(-1)
        assert same.float().mean().item() >= 0.99
        assert torch.allclose(gate_s[same], gate_o[same].float(), atol=2e-5, rtol=1e-4)
    assert idx_s.dtype == torch.int32 and gate_s.dtype == torch.float32
