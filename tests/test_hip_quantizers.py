"""GPU parity of MojoDynamicQuant / MojoResidualAddRMSNormQuant through the C ABI.

Reference tolerances: dynamic quant atol=(1, 2e-3) (test_quantize.py:198), norm+quant atol=(2, 1e-2, 1e-2)
(test_normalization.py:524-530).  This build: the dynamic quantiser is bit-identical (same IEEE operations); the fused
norm differs from the golden only through the fp32 row statistic (summation order), which can move a value that sits on a
rounding boundary by one step."""
import pytest
import torch

from conftest import load_golden
from hip_utils import DEV, hip_cls, run_hip_case, to_cpu, torch_cls

pytestmark = pytest.mark.gpu

CASES = load_golden("quantizers")


def _of(op):
    return [pytest.param(c, id=f"{op}-{i}") for i, c in enumerate(c for c in CASES if c["op"] == op)]


@pytest.mark.parametrize("case", _of("MojoDynamicQuant"))
def test_dynamic_quant_vectors_bit_exact(case):
    q, scale = to_cpu(run_hip_case(case))
    want_q, want_scale = case["out"]
    assert q.dtype == torch.int8 and scale.dtype == torch.float32 and scale.shape == want_scale.shape
    assert torch.equal(scale, want_scale)
    assert torch.equal(q, want_q)


@pytest.mark.parametrize("shape", [(1, 128), (8, 128), (17, 320), (24, 512), (48, 1536), (64, 2048), (3, 129), (7, 257),
                                   (4096, 7168), (3, 5, 96), (2, 40000)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("smooth", [True, False])
def test_dynamic_quant_reference_space_bit_exact(shape, dtype, smooth):
    torch.manual_seed(0)
    x = torch.randn(shape, dtype=dtype)
    ref = torch_cls("MojoDynamicQuant")(input_size=shape[-1] if smooth else None)
    op = hip_cls("MojoDynamicQuant")(input_size=shape[-1] if smooth else None)
    if smooth:
        inv = 1.0 / (torch.rand(shape[-1]) + 0.1)
        with torch.no_grad():
            ref.inv_smooth_scale.copy_(inv)
        op = op.to(DEV)
        op.load_state_dict(ref.state_dict())
    want_q, want_scale = ref(x)
    q, scale = to_cpu(op(x.to(DEV)))
    assert torch.equal(scale, want_scale) and torch.equal(q, want_q)


def test_dynamic_quant_rejects_other_dtypes():
    with pytest.raises(NotImplementedError):
        hip_cls("MojoDynamicQuant")(input_size=8, quant_dtype=torch.float8_e4m3fn)


def _check_norm_quant(got, want, quant_dtype):
    q, res, scale = got
    want_q, want_res, want_scale = want
    assert q.dtype == quant_dtype and res.dtype == want_res.dtype and scale.shape == want_scale.shape
    torch.testing.assert_close(scale, want_scale, atol=0, rtol=1e-5)
    if want_res.dtype == torch.float32:                       # post: the fp32 normed tensor
        torch.testing.assert_close(res, want_res, atol=1e-5, rtol=1e-5)
    else:                                                     # pre: hidden + residual, one rounding
        assert torch.equal(res, want_res)
    diff = (q.float() - want_q.float()).abs()
    step = 1.0 if quant_dtype == torch.int8 else 32.0         # fp8: integers up to 448 are spaced by up to 32
    assert float(diff.max()) <= step
    assert float((diff > 0).float().mean()) <= 2e-3


@pytest.mark.parametrize("case", _of("MojoResidualAddRMSNormQuant"))
def test_rmsnorm_quant_vectors(case):
    _check_norm_quant(to_cpu(run_hip_case(case)), case["out"], case["ctor"]["kwargs"]["quant_dtype"])


@pytest.mark.parametrize("shape", [(32, 1024), (64, 8192), (2, 256), (4096, 7168), (5, 1000), (3, 20000)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("norm_pos", ["pre", "post"])
@pytest.mark.parametrize("quant_dtype", [torch.int8, torch.float8_e4m3fn])
def test_rmsnorm_quant_reference_space(shape, dtype, norm_pos, quant_dtype):
    torch.manual_seed(0)
    x, r = torch.randn(shape, dtype=dtype), torch.randn(shape, dtype=dtype)
    ref = torch_cls("MojoResidualAddRMSNormQuant")(norm_size=shape[-1], norm_pos=norm_pos, quant_dtype=quant_dtype)
    with torch.no_grad():
        ref.weight.copy_(torch.randn(shape[-1]))
    op = hip_cls("MojoResidualAddRMSNormQuant")(norm_size=shape[-1], norm_pos=norm_pos, quant_dtype=quant_dtype).to(DEV)
    op.load_state_dict(ref.state_dict())
    smooth = torch.rand(shape[-1]) + 0.5 if shape[0] % 2 else None
    want = ref(x, r, smooth)
    got = to_cpu(op(x.to(DEV), r.to(DEV), None if smooth is None else smooth.to(DEV)))
    _check_norm_quant(got, want, quant_dtype)
    # the reference's own acceptance test
    op.forward_diff_with(ref, x.to(DEV), r.to(DEV), atol=(2 if quant_dtype == torch.int8 else 64, 1e-2, 1e-2),
                         rtol=(0, 1e-2, 1e-2), ref_device="cpu")


def test_quantised_activations_feed_quant_gemm():
    """dynamic quant -> int8 GEMM reproduces the float product to quantisation accuracy (the pipeline the two ops form)."""
    torch.manual_seed(0)
    m, k, n = 64, 1024, 512
    x = torch.randn(m, k, dtype=torch.bfloat16)
    w = torch.randn(k, n) * 0.05
    w_scale = (w.abs().amax(0) / 127).to(torch.bfloat16)
    w_q = torch.clamp(torch.round(w / w_scale.float()), -128, 127).to(torch.int8)
    quant = hip_cls("MojoDynamicQuant")()
    gemm = hip_cls("MojoQuantGemm")(k, n, output_dtype=torch.float32).to(DEV)
    with torch.no_grad():
        gemm.weight.copy_(w_q)
        gemm.weight_scale.copy_(w_scale)
    x_q, s = quant(x.to(DEV))
    got = to_cpu(gemm(x_q, s.reshape(-1)))
    want = x.float() @ w
    assert float((got - want).abs().max()) < 0.05 * float(want.abs().max())
