"""Known-answer tests the reference ships for this path, restated against the oracle.

* quant gemm == exact integer formula at atol=rtol=0
  (`mojo_opset/tests/accuracy/operators/test_gemm.py:95-111`, registered-buffer state dict :133-146)
* swiglu_limit clamps before the activation (`test_activation.py:69-79`)
* MLA attention-sink softmax closed form (`test_attention.py:1055-1077`), sink optional (:1079-…)
"""
import math

import pytest
import torch

import mojo_opset_amd as mo
import oracle  # noqa: F401  (registers the Torch* classes)
from oracle import quant_gemm_formula


def _quantize(x):
    scale = x.abs().amax(dim=-1).clamp_min(1e-8) / 127.0
    return torch.clamp(torch.round(x / scale.unsqueeze(-1)), -128, 127).to(torch.int8), scale


@pytest.mark.parametrize("m,k,n", [(1, 4096, 256), (32, 4096, 172), (128, 2048, 64), (64, 4096, 96)])
@pytest.mark.parametrize("trans_weight", [False, True])
@pytest.mark.parametrize("odt", [torch.bfloat16, torch.float16, torch.float32])
def test_quant_gemm_equals_integer_formula(m, k, n, trans_weight, odt):
    torch.manual_seed(0)
    xq, xs = _quantize(torch.randn(m, k))
    wq, ws = _quantize(torch.randn(n, k))
    op = mo.MojoQuantGemm.get_backend_impl("torch", strict=True)(
        in_features=k, out_features=n, output_dtype=odt, trans_weight=trans_weight)
    op.weight.copy_(wq if trans_weight else wq.t())
    op.weight_scale.copy_(ws.to(torch.bfloat16))
    out = op(xq, xs)
    expect = quant_gemm_formula(xq, wq.t(), xs, ws.to(torch.bfloat16), odt)
    torch.testing.assert_close(out, expect, atol=0, rtol=0)
    assert set(op.state_dict()) == {"weight", "weight_scale"}


def test_quant_gemm_rejects_bad_shapes():
    op = mo.MojoQuantGemm.get_backend_impl("torch")(in_features=8, out_features=4)
    with pytest.raises(ValueError):
        op(torch.zeros(2, 3, 8, dtype=torch.int8), torch.ones(2))
    with pytest.raises(ValueError):
        op(torch.zeros(2, 7, dtype=torch.int8), torch.ones(2))


def test_swiglu_limit_reference():
    limit = 1.5
    gate = torch.tensor([[-3.0, -0.5, 0.25, 2.0, 9.0]])
    up = torch.tensor([[-4.0, -1.0, 0.5, 1.75, 6.0]])
    op = mo.MojoSwiGLU.get_backend_impl("torch")(swiglu_limit=limit)
    expect = torch.nn.functional.silu(gate.clamp(max=limit)) * up.clamp(-limit, limit)
    torch.testing.assert_close(op(gate, up), expect, atol=0, rtol=0)


def test_mla_attn_sink_closed_form():
    torch.manual_seed(1)
    h, nope, rope, vd, r, page = 4, 8, 4, 8, 6, 4
    op = mo.MojoPagedDecodeMLA.get_backend_impl("torch")(h, nope, rope, vd, r, use_attn_sink=True)
    with torch.no_grad():
        op.kv_b_proj.copy_(torch.randn_like(op.kv_b_proj))
        op.attn_sink.copy_(torch.tensor([0.0, 1.0, -1.0, 2.0]))
    lens = [7]
    ckv = torch.randn(3, 1, page, r)
    kpe = torch.randn(3, 1, page, rope)
    table = torch.tensor([[2, 0]], dtype=torch.int32)
    q = torch.randn(1, h, nope + rope)
    out = op(q, ckv, kpe, torch.tensor(lens, dtype=torch.int32), table)

    c = torch.cat([ckv[2, 0], ckv[0, 0, :3]])
    pe = torch.cat([kpe[2, 0], kpe[0, 0, :3]])
    kv = (c @ op.kv_b_proj.T).view(7, h, nope + vd)
    k = torch.cat([kv[..., :nope], pe[:, None, :].expand(-1, h, -1)], -1)
    s = torch.einsum("hd,shd->hs", q[0], k) / math.sqrt(nope + rope)
    e = torch.exp(s - s.max(-1, keepdim=True).values)
    sink = torch.exp(op.attn_sink.detach()[:, None] - s.max(-1, keepdim=True).values)
    p = e / (e.sum(-1, keepdim=True) + sink)
    expect = torch.einsum("hs,shd->hd", p, kv[..., nope:])
    torch.testing.assert_close(out[0], expect, atol=1e-5, rtol=1e-5)


def test_mla_attn_sink_parameter_is_optional():
    op = mo.MojoPagedDecodeMLA.get_backend_impl("torch")(4, 8, 4, 8, 6)
    assert not hasattr(op, "attn_sink")
    op = mo.MojoPagedPrefillMLA.get_backend_impl("torch")(4, 8, 4, 8, 6, use_attn_sink=True)
    assert op.attn_sink.dtype == torch.float32 and op.attn_sink.shape == (4,)


def test_fp8_quant_gemm_oracle_matches_float64_formula():
    """fp8 extension: parity unpinned (no reference); oracle vs an independent fp64 formula."""
    torch.manual_seed(2)
    m, k, n = 16, 256, 48
    x = (torch.randn(m, k)).to(torch.float8_e4m3fn)
    w = (torch.randn(k, n)).to(torch.float8_e4m3fn)
    op = mo.MojoQuantGemm.get_backend_impl("torch")(k, n, output_dtype=torch.float32,
                                                    quant_dtype=torch.float8_e4m3fn,
                                                    weight_dtype=torch.float8_e4m3fn)
    op.weight.copy_(w)
    op.weight_scale.copy_(torch.rand(n).to(torch.bfloat16))
    s = torch.rand(m)
    expect = quant_gemm_formula(x, w, s, op.weight_scale, torch.float32)
    torch.testing.assert_close(op(x, s), expect, atol=1e-4, rtol=1e-4)
