"""GPU parity of `HIPGemm` (`MojoGemm`, core/operators/gemm.py:12-56) and `HIPSwiGLUMLP` (`MojoSwiGLUMLP`,
core/operators/mlp.py:7-37) through the C ABI: the reference's hooks for the dense projections and the gated MLP of a decoder
layer, which bind this backend's decode-sized forms (weight-stream GEMM with split-K, SwiGLU in the projection's epilogue) to
the `Mojo*` operator API.

Bounds: the reference's own test of `MojoGemm` (tests/accuracy/operators/test_gemm.py:34-54) uses `forward_diff_with(...,
mixed_tol=True)` (atol 2^-6 below 1, rtol 2^-6 above) and `assert_close` defaults on (1024, 4096, 4096) fp16 / bf16 with and
without bias; `MojoSwiGLUMLP` has no accuracy test in the reference — it is held here to the reference's GEMM-family bound
against the oracle plus a mean error below 1 % of the mean magnitude (every rounding point of the golden's chain is
reproduced), to two units in the last place on the reference-captured vectors, and to the separate-operator chain bit for bit
wherever both sum K in the same order.
"""
import pytest
import torch

import mojo_opset_amd as mo
from conftest import load_golden
from hip_utils import DEV, hip_cls, last_launch, launches_of, max_ulp_bf16ish, run_hip_case, to_cpu, torch_cls

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", [pytest.param(c, id=f"dense-{i}-{c['op']}") for i, c in enumerate(load_golden("dense"))])
def test_dense_vectors(case):
    """Vectors captured from the imported reference (oracle/make_golden.py gen_dense)."""
    got = to_cpu(run_hip_case(case))
    want = case["out"]
    assert got.dtype == want.dtype and got.shape == want.shape
    if want.dtype == torch.float32:
        torch.testing.assert_close(got, want, atol=1e-4, rtol=1e-4)
    else:
        # F.linear semantics reproduced (fp32 accumulation, bias in the accumulator, one rounding): what is left is the
        # summation order of two fp32 GEMMs; the MLP chains three rounded stages
        assert max_ulp_bf16ish(got, want, atol=2e-3) <= (1 if case["op"] == "MojoGemm" else 2)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("bias", [True, False])
def test_gemm_reference_test_case(dtype, bias):
    """The reference's own case and its two assertions (test_gemm.py:34-54): (1024, 4096, 4096), `forward_diff_with(...,
    mixed_tol=True)` against the torch backend and `assert_close` with torch's default bounds against `F.linear`."""
    torch.manual_seed(0)
    m, k, n = 1024, 4096, 4096
    ref = torch_cls("MojoGemm")(k, n, bias=bias, dtype=dtype)
    op = hip_cls("MojoGemm")(k, n, bias=bias, dtype=dtype).to(DEV)
    assert set(op.state_dict()) == ({"weight", "bias"} if bias else {"weight"})
    op.load_state_dict({k_: v.to(DEV) for k_, v in ref.state_dict().items()})
    x = torch.randn(m, k, dtype=dtype)
    op.forward_diff_with(ref, x.to(DEV), mixed_tol=True, ref_device="cpu")
    want = torch.nn.functional.linear(x, ref.weight, ref.bias)
    got = to_cpu(op(x.to(DEV)))
    torch.testing.assert_close(got, want.detach())
    assert max_ulp_bf16ish(got, want.detach(), atol=2e-2 if dtype == torch.bfloat16 else 4e-3) <= 1


def test_gemm_constructor_contract_and_weight_form():
    """`MojoGemm(weight=w)` wraps the tensor (no bias); mixing both forms, a non-2-D weight or missing sizes raise ValueError
    (gemm.py:22-34)."""
    w = (torch.randn(48, 128) * 0.1).to(torch.bfloat16).to(DEV)
    op = hip_cls("MojoGemm")(weight=w)
    assert op.in_features == 128 and op.out_features == 48 and op.bias is None and op.weight.data_ptr() == w.data_ptr()
    x = torch.randn(9, 128).to(torch.bfloat16)
    want = torch.nn.functional.linear(x, w.cpu())
    assert max_ulp_bf16ish(to_cpu(op(x.to(DEV))), want, atol=2e-3) <= 1
    for bad in (dict(in_features=8, weight=w), dict(weight=w[0]), dict(in_features=8), dict()):
        with pytest.raises(ValueError):
            hip_cls("MojoGemm")(**bad)
    assert isinstance(mo.MojoGemm(8, 4, device=DEV), hip_cls("MojoGemm"))            # the default backend on a ROCm host


@pytest.mark.parametrize("m,inp,hidden,outp", [(64, 4096, 14336, 4096), (1, 512, 1016, 256), (16, 1024, 2048, 1024), (300, 512, 1024, 384)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_swiglu_mlp_takes_the_fused_forms_and_matches_the_chain_and_the_oracle(m, inp, hidden, outp, dtype):
    """Decode-sized rows: ONE launch for fc1 + SwiGLU (`gemm_skinny:glu`), then the down projection.  The result equals, bit for
    bit, the chain of separate operators (`HIPGemm` -> `HIPSwiGLU` on the two halves -> `HIPGemm`), and the oracle within two
    units in the last place (three rounded stages; Llama-3-8B's MLP shape first)."""
    torch.manual_seed(m + hidden)
    ref = torch_cls("MojoSwiGLUMLP")(inp, outp, hidden).to(dtype)
    with torch.no_grad():
        ref.fc1.weight.normal_(std=0.03)
        ref.fc2.weight.normal_(std=0.03)
    op = hip_cls("MojoSwiGLUMLP")(inp, outp, hidden).to(dtype).to(DEV)
    assert set(op.state_dict()) == {"fc1.weight", "fc2.weight"}
    op.load_state_dict({k: v.to(DEV) for k, v in ref.state_dict().items()})
    x = torch.randn(m, inp).to(dtype)
    hist = launches_of(lambda: op(x.to(DEV)))
    got = op(x.to(DEV))
    if m <= 64:
        assert "gemm_skinny:glu" in hist, hist                   # the fused epilogue really ran
    fc1 = hip_cls("MojoGemm")(weight=op.fc1.weight.detach())
    fc2 = hip_cls("MojoGemm")(weight=op.fc2.weight.detach())
    a = fc1(x.to(DEV))
    fc1_form = last_launch()
    chain = fc2(hip_cls("MojoSwiGLU")()(a[:, :hidden], a[:, hidden:]))
    want = ref(x).detach()
    scale = float(want.float().abs().mean())
    if "splitk1" in fc1_form or fc1_form.startswith("gemm256:") and ":splitk" not in fc1_form:
        assert torch.equal(got, chain), fc1_form                  # same fp32 summation order in both: the same bits
    else:
        # the stand-alone projection cuts K into slices (decode-sized rows, few column tiles) where the fused kernel sums K
        # in one pass: another fp32 order, so a last-place flip of a few activations — far below the output's own ulp
        assert float((got.float() - chain.float()).abs().max()) <= 0.02 * scale + 2.0 ** -8 * float(want.float().abs().max()), fc1_form
    # against the oracle: the reference's GEMM-family bound (mixed_tol: atol 2^-6 below 1, rtol 2^-6 above, utils/acc.py:40-44)
    # and, far tighter, a mean error below 1 % of the mean magnitude (three rounded stages, each reproduced)
    mo.check_tol_diff(to_cpu(got), want, mixed_tol=True)
    assert float((to_cpu(got).float() - want.float()).abs().mean()) <= 0.01 * scale
    x3 = x.reshape(1, m, inp) if m > 1 else x.reshape(1, 1, inp)
    assert torch.equal(op(x3.to(DEV)).reshape(m, outp), got)      # leading dimensions are flattened like nn.Linear's


def test_swiglu_mlp_fp32_and_graph_capture():
    """fp32 inputs run projection -> `mojo_hip_swiglu_rows` -> projection; a decode-sized bf16 call captures into a HIP graph
    (no host sync, no allocation outside torch's allocator) and replays on new inputs."""
    torch.manual_seed(3)
    ref = torch_cls("MojoSwiGLUMLP")(64, 32, 72)
    op = hip_cls("MojoSwiGLUMLP")(64, 32, 72).to(DEV)
    op.load_state_dict({k: v.to(DEV) for k, v in ref.state_dict().items()})
    x = torch.randn(11, 64)
    torch.testing.assert_close(to_cpu(op(x.to(DEV))), ref(x).detach(), atol=1e-4, rtol=1e-4)
    op16 = hip_cls("MojoSwiGLUMLP")(512, 512, 1024).to(torch.bfloat16).to(DEV)
    xs = torch.zeros(32, 512, dtype=torch.bfloat16, device=DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        op16(xs)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = op16(xs)
    for seed in (1, 2):
        xs.copy_(torch.randn(32, 512, generator=torch.Generator().manual_seed(seed)).to(torch.bfloat16))
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, op16(xs))
    assert last_launch()


@pytest.mark.parametrize("m,inp,hidden", [(512, 1024, 14336), (2048, 4096, 14336), (300, 512, 24576)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gemm_swiglu_prefill_rows_take_the_fused_tiles_and_give_the_unfused_bits(m, inp, hidden, dtype):
    """More than 128 rows and enough fused tiles to fill the chip: `mojo_hip_gemm_swiglu` runs the 256 x 256 kernel's
    fused-SwiGLU epilogue with one group (`gemm256:...:glu`; the [M, 2 hidden] product never exists).  A/B against the unfused
    route (MOJO_HIP_GEMM_SKINNY without bit 4: the product, then `mojo_hip_swiglu_rows`), both forms asserted: the fused
    epilogue rounds where the two launches round, so the outputs are the same bits; and both match the oracle's MLP front half."""
    from hip_utils import switch_env
    from mojo_opset_amd.backends.hip.operators.gemm import dense_gemm_swiglu
    torch.manual_seed(m + hidden)
    x = torch.randn(m, inp, device=DEV).to(dtype)
    w = (torch.randn(2 * hidden, inp, device=DEV) * 0.03).to(dtype)
    hist = launches_of(lambda: dense_gemm_swiglu(x, w))
    fused = dense_gemm_swiglu(x, w)
    assert ":glu" in hist and "gemm256:" in hist, hist
    with switch_env(MOJO_HIP_GEMM_SKINNY="27"):
        hist2 = launches_of(lambda: dense_gemm_swiglu(x, w))
        unfused = dense_gemm_swiglu(x, w)
    assert ":glu" not in hist2, hist2
    assert torch.equal(fused, unfused)
    a = torch.nn.functional.linear(x.float(), w.float()).to(dtype).float()           # the golden's rounding points
    want = (torch.nn.functional.silu(a[:, :hidden]).to(dtype).float() * a[:, hidden:]).to(dtype)
    # (three rounded stages: where the rounded gate or up value flips its last place — a different fp32 summation order is
    # enough — silu(gate) * up moves by up to two of the factors' last places: 2^-7 each; over 29 M outputs a handful sit at 2 %.
    # Twice the reference's GEMM-family tolerance per element, and a mean error below 1 % of the mean magnitude)
    err = (fused.float() - want.float()).abs()
    assert bool((err <= 2.0 ** -5 * want.float().abs() + 2.0 ** -5).all())
    assert float(err.mean()) <= 0.01 * float(want.float().abs().mean())


@pytest.mark.parametrize("layout", ["NK", "KN"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,k,n", [(2048, 512, 2048), (4096, 1024, 1536), (1000, 256, 4096)])
def test_gemm_with_bias_leaves_through_the_row_staged_epilogue_with_the_direct_stores_bits(m, k, n, dtype, layout):
    """Chip-filling products WITH a bias on the 256 x 256 kernel: the row-staged epilogue now takes the bias from LDS (fetched
    before the K loop) instead of sending the tile through the direct 8-byte stores.  A/B against MOJO_HIP_GEMM_STAGE_ROWS=0
    (direct stores, bias fetched in the epilogue), both forms asserted: the same bits; small-integer data: equal to the golden's
    rounding exactly (`F.linear`: one rounding; `x @ w + b`: two)."""
    from hip_utils import switch_env
    from mojo_opset_amd.backends.hip.operators.gemm import dense_gemm
    torch.manual_seed(m + n)
    trans = layout == "KN"
    x = torch.randint(-4, 5, (m, k)).to(dtype).to(DEV)
    w = torch.randint(-4, 5, (n, k)).to(dtype).to(DEV)
    b = torch.randint(-8, 9, (n,)).to(dtype).to(DEV)
    if trans:
        w = w.t().contiguous()
        want = ((x.float() @ w.float()).to(dtype).float() + b.float()).to(dtype)
    else:
        want = torch.nn.functional.linear(x.float(), w.float(), b.float()).to(dtype)
    with switch_env(MOJO_HIP_GEMM_TILE128="0"):
        staged = dense_gemm(x, w, b, trans)
        assert last_launch().startswith("gemm256:staged"), last_launch()
        with switch_env(MOJO_HIP_GEMM_STAGE_ROWS="0"):
            direct = dense_gemm(x, w, b, trans)
            assert last_launch().startswith("gemm256:direct"), last_launch()
        xr = torch.randn(m, k, device=DEV).to(dtype)
        br = torch.randn(n, device=DEV).to(dtype)
        wr = (torch.randn(k, n, device=DEV) * 0.05).to(dtype) if trans else (torch.randn(n, k, device=DEV) * 0.05).to(dtype)
        staged_r = dense_gemm(xr, wr, br, trans)
        with switch_env(MOJO_HIP_GEMM_STAGE_ROWS="0"):
            direct_r = dense_gemm(xr, wr, br, trans)
    assert torch.equal(staged, want) and torch.equal(direct, want)
    assert torch.equal(staged_r, direct_r)
