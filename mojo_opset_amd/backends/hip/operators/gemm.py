"""HIP<Op> classes of the GEMM family."""
import torch

from ....core.operators.gemm import MojoGemm, MojoGroupGemm, MojoQuantGemm
from ....core.operators.mlp import MojoSwiGLUMLP
from .. import lib as L

_ROCM = ["rocm"]


def dense_gemm(x: torch.Tensor, weight: torch.Tensor, bias, trans_weight: bool) -> torch.Tensor:
    """``x @ weight (+ bias)`` for ``weight [K,N]`` (trans_weight) or ``[N,K]`` (F.linear layout) through
    `mojo_hip_gemm`; leading dimensions of ``x`` are flattened.  Shared by the GEMM+collective ops."""
    L.require_cuda(x, weight, bias)
    if weight.dim() != 2 or x.dtype != weight.dtype or (bias is not None and bias.dtype != x.dtype):
        raise NotImplementedError("hip gemm: 2-D weight and one common dtype required")
    k = x.shape[-1]
    if trans_weight:
        assert weight.shape[0] == k, "input K must match weight K"
        n = weight.shape[1]
    else:
        assert weight.shape[1] == k, "input K must match weight K"
        n = weight.shape[0]
    if weight.stride(0) != 1 and weight.stride(1) != 1:
        weight = weight.contiguous()
    w_k, w_n = (weight.stride(0), weight.stride(1)) if trans_weight else (weight.stride(1), weight.stride(0))
    x2 = x.reshape(-1, k)
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    m = x2.shape[0]
    out = torch.empty(m, n, dtype=x.dtype, device=x.device)
    ws = torch.empty(L.load().mojo_hip_gemm_workspace_bytes(m, k, n), dtype=torch.uint8, device=x.device)
    L.check(L.load().mojo_hip_gemm(L.ptr(x2), L.ptr(weight), L.ptr(None if bias is None else bias.contiguous()),
                                   L.ptr(out), m, k, n, x2.stride(0), n, w_k, w_n, L.dtype_code(x.dtype), L.ptr(ws),
                                   ws.numel(), L.stream_of(x2)), "hip gemm")
    return out.reshape(*x.shape[:-1], n)


def dense_gemm_swiglu(x: torch.Tensor, weight_gate_up: torch.Tensor) -> torch.Tensor:
    """``silu(x @ Wg.T) * (x @ Wu.T)`` for ``weight_gate_up = [Wg; Wu]`` of shape ``[2*inter, K]`` (F.linear layout, the
    fused gate|up projection of a gated MLP; op chain core/operators/moe.py:441-445, activation.py:38-66) through
    `mojo_hip_gemm_swiglu`.  Same bits as ``dense_gemm`` followed by ``HIPSwiGLU`` on the two halves; at <= 64 rows it is
    one launch and the ``[M, 2*inter]`` product never exists."""
    L.require_cuda(x, weight_gate_up)
    w = weight_gate_up
    if w.dim() != 2 or x.dtype != w.dtype or w.shape[0] % 2 or x.dtype not in (torch.bfloat16, torch.float16):
        raise NotImplementedError("hip gemm_swiglu: one 16-bit dtype and a [2*inter, K] weight required")
    k, inter = x.shape[-1], w.shape[0] // 2
    assert w.shape[1] == k, "input K must match weight K"
    if w.stride(1) != 1:
        w = w.contiguous()
    x2 = x.reshape(-1, k)
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    m = x2.shape[0]
    out = torch.empty(m, inter, dtype=x.dtype, device=x.device)
    lib = L.load()
    ws = torch.empty(lib.mojo_hip_gemm_swiglu_workspace_bytes(m, k, inter), dtype=torch.uint8, device=x.device)
    L.check(lib.mojo_hip_gemm_swiglu(L.ptr(x2), L.ptr(w), L.ptr(out), m, k, inter, x2.stride(0), inter, w.stride(0),
                                     L.dtype_code(x.dtype), L.ptr(ws), ws.numel(), L.stream_of(x2)), "hip gemm_swiglu")
    return out.reshape(*x.shape[:-1], inter)


def dense_gemm_residual_rmsnorm(x: torch.Tensor, weight: torch.Tensor, bias, residual, norm_weight: torch.Tensor,
                                eps: float, trans_weight: bool = False):
    """``(RMSNorm(x @ W (+ bias) + residual) * norm_weight, x @ W (+ bias) + residual)`` — a projection followed by
    `MojoResidualAddRMSNorm(norm_pos="pre")` (core/operators/normalization.py:308-362) through
    `mojo_hip_gemm_residual_rmsnorm`.  Same bits as ``dense_gemm`` followed by ``HIPResidualAddRMSNorm``; when the
    projection is cut along K (decode-sized M) the K-slice sums feed the norm kernel directly."""
    L.require_cuda(x, weight, bias, residual, norm_weight)
    if weight.dim() != 2 or x.dtype != weight.dtype or x.dtype not in (torch.bfloat16, torch.float16):
        raise NotImplementedError("hip gemm_residual_rmsnorm: 2-D weight and one 16-bit dtype required")
    k = x.shape[-1]
    n = weight.shape[1] if trans_weight else weight.shape[0]
    assert (weight.shape[0] if trans_weight else weight.shape[1]) == k, "input K must match weight K"
    if weight.stride(0) != 1 and weight.stride(1) != 1:
        weight = weight.contiguous()
    w_k, w_n = (weight.stride(0), weight.stride(1)) if trans_weight else (weight.stride(1), weight.stride(0))
    x2 = x.reshape(-1, k)
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    m = x2.shape[0]
    res2 = None if residual is None else residual.reshape(m, n).contiguous()
    normed = torch.empty(m, n, dtype=x.dtype, device=x.device)
    summed = torch.empty(m, n, dtype=x.dtype, device=x.device) if residual is not None else None
    lib = L.load()
    ws = torch.empty(lib.mojo_hip_gemm_residual_rmsnorm_workspace_bytes(m, k, n), dtype=torch.uint8, device=x.device)
    L.check(lib.mojo_hip_gemm_residual_rmsnorm(
        L.ptr(x2), L.ptr(weight), L.ptr(None if bias is None else bias.contiguous()), L.ptr(res2),
        L.ptr(norm_weight.to(x.dtype).contiguous()), L.ptr(normed), L.ptr(summed), None, m, k, n, x2.stride(0), w_k, w_n,
        L.dtype_code(x.dtype), float(eps), L.ptr(ws), ws.numel(), L.stream_of(x2)), "hip gemm_residual_rmsnorm")
    shape = (*x.shape[:-1], n)
    return normed.reshape(shape), (None if summed is None else summed.reshape(shape))


def qkv_rope_store(x: torch.Tensor, weight_qkv: torch.Tensor, bias, cos: torch.Tensor, sin: torch.Tensor,
                   key_cache: torch.Tensor, value_cache: torch.Tensor, block_table: torch.Tensor,
                   context_kv_lens: torch.Tensor, q_heads: int, kv_heads: int) -> torch.Tensor:
    """One decode step in front of the attention: ``qkv = x @ Wqkv.T (+ bias)`` (heads in q | k | v order), `MojoApplyRoPE`
    on q and k (``cos`` / ``sin`` [B, D], rotate-half over the whole head), `MojoStorePagedKVCache` of the rotated k and of
    v at position ``context_kv_lens[b]``; returns the rotated q ``[B, Hq, D]``.  Same bits as ``dense_gemm`` -> ``HIPApplyRoPE``
    -> ``HIPStorePagedKVCache`` (decode mode); through `mojo_hip_qkv_rope_store` the chain is two launches."""
    L.require_cuda(x, weight_qkv, bias, cos, sin, key_cache, value_cache, block_table, context_kv_lens)
    if x.dtype not in (torch.bfloat16, torch.float16) or not (x.dtype == weight_qkv.dtype == key_cache.dtype == value_cache.dtype):
        raise NotImplementedError("hip qkv_rope_store: one 16-bit dtype for activations, weights and caches required")
    n_blocks, hkv, page, d = key_cache.shape
    assert hkv == kv_heads and value_cache.shape == key_cache.shape and key_cache.stride() == value_cache.stride()
    assert key_cache.stride(-1) == 1
    b, k = x.shape
    n = (q_heads + 2 * kv_heads) * d
    assert weight_qkv.shape == (n, k), "weight must be [(Hq + 2 Hkv) * D, K]"
    assert cos.shape == (b, d) and sin.shape == (b, d), "cos / sin: one [D] row per sequence (rope over the whole head)"
    w = weight_qkv if weight_qkv.stride(1) == 1 else weight_qkv.contiguous()
    x2 = x if x.stride(1) == 1 else x.contiguous()
    cos = cos.float()
    sin = sin.float()
    if cos.stride(1) != 1 or sin.stride() != cos.stride():
        cos, sin = cos.contiguous(), sin.contiguous()
    table = block_table if block_table.stride(1) == 1 else block_table.contiguous()
    ctx = context_kv_lens.to(torch.int32).contiguous()
    q_out = torch.empty(b, q_heads, d, dtype=x.dtype, device=x.device)
    lib = L.load()
    ws = torch.empty(lib.mojo_hip_qkv_rope_store_workspace_bytes(b, k, n), dtype=torch.uint8, device=x.device)
    L.check(lib.mojo_hip_qkv_rope_store(
        L.ptr(x2), L.ptr(w), L.ptr(None if bias is None else bias.contiguous()), L.ptr(cos), L.ptr(sin), cos.stride(0),
        L.ptr(q_out), L.ptr(key_cache), L.ptr(value_cache), L.ptr(table), table.stride(0), table.shape[1], L.ptr(ctx), b, k,
        q_heads, kv_heads, d, x2.stride(0), w.stride(0), n_blocks, page, key_cache.stride(0), key_cache.stride(1),
        key_cache.stride(2), L.dtype_code(x.dtype), L.ptr(ws), ws.numel(), L.stream_of(x2)), "hip qkv_rope_store")
    return q_out


class HIPGemm(MojoGemm):
    """`MojoGemm` (core/operators/gemm.py:12-56) on `mojo_hip_gemm`: ``F.linear(input, weight, bias)``.  Decode-sized inputs take
    the weight-stream kernels with split-K (csrc/gemm_skinny.hip), mid-size ones the 128-row tiles (csrc/gemm_tile128.hip), larger ones the 256 x 256 tile kernel; the bias joins the fp32
    accumulator and the sum is rounded ONCE to the storage type, as the golden's ``F.linear(x, w, b)`` does (fixtures:
    tests/golden, `tests/test_hip_dense.py`)."""

    supported_platforms_list = _ROCM

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        w = self.weight.detach()
        b = None if self.bias is None else self.bias.detach()
        return dense_gemm(input, w, b, False)


class HIPSwiGLUMLP(MojoSwiGLUMLP):
    """`MojoSwiGLUMLP` (core/operators/mlp.py:7-37): ``fc2(silu(a1) * a2)``, ``a1, a2 = fc1(x).chunk(2, -1)``.  16-bit inputs
    run `mojo_hip_gemm_swiglu` (at <= 64 rows ONE launch whose epilogue applies SwiGLU to the accumulators — the
    ``[M, 2 * hidden]`` product never exists; same rounding points as projection -> `MojoSwiGLU`) followed by `mojo_hip_gemm`;
    fp32 inputs run the projection, `mojo_hip_swiglu_rows` on its two halves, and the second projection.  ``fc1.weight``
    ``[2 * hidden, in]`` is used in place: its first ``hidden`` rows are the gate, the rest the up projection — the layout
    the fused kernel wants, so `state_dict` keys and layouts stay the golden's."""

    supported_platforms_list = _ROCM

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        w1, w2 = self.fc1.weight.detach(), self.fc2.weight.detach()
        if x.dtype in (torch.bfloat16, torch.float16) and x.dtype == w1.dtype == w2.dtype:
            return dense_gemm(dense_gemm_swiglu(x, w1), w2, None, False)
        a = dense_gemm(x, w1, None, False)
        hidden = w1.shape[0] // 2
        a2d = a.reshape(-1, 2 * hidden)
        act = torch.empty(a2d.shape[0], hidden, dtype=a.dtype, device=a.device)
        L.check(L.load().mojo_hip_swiglu_rows(L.ptr(a2d), L.c_void_p(a2d.data_ptr() + hidden * a2d.element_size()), L.ptr(act),
                                              a2d.shape[0], hidden, 2 * hidden, 2 * hidden, hidden, L.dtype_code(a.dtype), 0.0,
                                              L.stream_of(a2d)), "HIPSwiGLUMLP swiglu")
        return dense_gemm(act.reshape(*a.shape[:-1], hidden), w2, None, False)


class HIPGroupGemm(MojoGroupGemm):
    supported_platforms_list = _ROCM

    def forward(self, input: torch.Tensor, group_list: torch.Tensor) -> torch.Tensor:
        self.check_call_contract(input, group_list)
        weight = self.weight
        L.require_cuda(input, weight)
        if input.dtype != weight.dtype:
            raise NotImplementedError("HIPGroupGemm: input and weight must share one dtype")
        if group_list.dtype not in (torch.int32, torch.int64):
            raise NotImplementedError("HIPGroupGemm: group_list must be int32 or int64")
        counts = group_list.to(input.device, non_blocking=True).contiguous()   # stays on the device: no sync
        x = input if input.is_contiguous() else input.contiguous()
        w = weight if weight.is_contiguous() else weight.contiguous()
        groups = w.shape[0]
        m, k = x.shape
        n = w.shape[1] if self.trans_weight else w.shape[2]
        out = torch.empty(m, n, dtype=x.dtype, device=x.device)
        lib = L.load()
        ws = torch.empty(lib.mojo_hip_group_gemm_workspace_bytes(groups), dtype=torch.uint8, device=x.device)
        L.check(lib.mojo_hip_group_gemm(L.ptr(x), L.ptr(w), L.ptr(out), L.ptr(counts),
                                        1 if counts.dtype == torch.int64 else 0, m, k, n, groups,
                                        1 if self.trans_weight else 0, L.dtype_code(x.dtype), L.ptr(ws), ws.numel(),
                                        L.stream_of(x)), "HIPGroupGemm")
        return out


class HIPQuantGemm(MojoQuantGemm):
    supported_platforms_list = _ROCM

    def forward(self, input: torch.Tensor, input_scale: torch.Tensor) -> torch.Tensor:
        self.check_call_contract(input, input_scale)
        weight = self.weight
        L.require_cuda(input, input_scale, weight, self.weight_scale)
        if input.dtype != weight.dtype or input.dtype not in (torch.int8, torch.float8_e4m3fn):
            raise NotImplementedError("HIPQuantGemm: int8 (or fp8-e4m3) activations and weights of one dtype required")
        m, k = input.shape
        n = self.out_features
        if input_scale.numel() != m:
            raise ValueError(f"input_scale must have one entry per row, got {tuple(input_scale.shape)} for M={m}")
        x = input if input.is_contiguous() else input.contiguous()
        w = weight if weight.is_contiguous() else weight.contiguous()
        s_in = input_scale.reshape(-1).to(torch.float32).contiguous()
        s_w = self.weight_scale.to(torch.bfloat16).contiguous()
        out = torch.empty(m, n, dtype=self.output_dtype, device=x.device)
        lib = L.load()
        ws = torch.empty(lib.mojo_hip_quant_gemm_workspace_bytes(m, k, n), dtype=torch.uint8, device=x.device)
        L.check(lib.mojo_hip_quant_gemm(L.ptr(x), L.ptr(w), L.ptr(s_in), L.ptr(s_w), L.ptr(out), m, k, n,
                                             1 if self.trans_weight else 0, L.dtype_code(x.dtype),
                                             L.dtype_code(self.output_dtype), L.ptr(ws), ws.numel(), L.stream_of(x)),
                "HIPQuantGemm")
        return out
