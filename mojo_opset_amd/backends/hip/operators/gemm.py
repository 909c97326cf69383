"""HIP<Op> classes of the GEMM family."""
import torch

from ....core.operators.gemm import MojoGroupGemm, MojoQuantGemm
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
