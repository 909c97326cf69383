"""HIP<Op> classes of the per-token activation quantisers."""
from typing import Optional

import torch

from ....core.operators.quantize import MojoDynamicQuant, MojoResidualAddRMSNormQuant
from .. import lib as L

_ROCM = ["rocm"]


def _dense(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


def _fp32_vector(t: Optional[torch.Tensor], dim: int, what: str):
    if t is None:
        return None
    if t.dtype != torch.float32:
        raise NotImplementedError(f"hip quantiser: {what} must be float32 (the reference pins it with force_dtype), got {t.dtype}")
    t = _dense(t.detach()).reshape(-1)
    if t.numel() != dim:
        raise ValueError(f"{what} has {t.numel()} elements, the last input dimension is {dim}")
    return t


class HIPDynamicQuant(MojoDynamicQuant):
    supported_platforms_list = _ROCM

    def forward(self, input: torch.Tensor):
        if input.dim() < 1:
            raise ValueError("input must have at least one dimension.")
        L.require_cuda(input, self.inv_smooth_scale)
        x = _dense(input)
        dim = x.shape[-1]
        rows = x.numel() // dim if dim else 0
        inv = _fp32_vector(self.inv_smooth_scale, dim, "inv_smooth_scale")
        out = torch.empty(x.shape, dtype=torch.int8, device=x.device)
        scale = torch.empty(*x.shape[:-1], 1, dtype=torch.float32, device=x.device)
        L.check(L.load().mojo_hip_dynamic_quant(L.ptr(x), L.ptr(inv), L.ptr(out), L.ptr(scale), rows, dim,
                                                L.dtype_code(x.dtype), L.stream_of(x)), "HIPDynamicQuant")
        return out, scale


class HIPResidualAddRMSNormQuant(MojoResidualAddRMSNormQuant):
    supported_platforms_list = _ROCM

    def forward(self, hidden_state: torch.Tensor, residual: torch.Tensor, smooth_scale: Optional[torch.Tensor] = None):
        L.require_cuda(hidden_state, residual, smooth_scale, self.weight)
        if residual.shape != hidden_state.shape or residual.dtype != hidden_state.dtype:
            raise NotImplementedError("HIPResidualAddRMSNormQuant: hidden_state and residual must share shape and dtype")
        h, r = _dense(hidden_state), _dense(residual)
        dim = h.shape[-1]
        rows = h.numel() // dim if dim else 0
        weight = _fp32_vector(self.weight, dim, "weight")
        smooth = _fp32_vector(smooth_scale, dim, "smooth_scale")
        dev = h.device
        out = torch.empty(h.shape, dtype=self.quant_dtype, device=dev)
        scale = torch.empty(*h.shape[:-1], 1, dtype=torch.float32, device=dev)
        pre = self.norm_pos == "pre"
        summed = torch.empty_like(h) if pre else None
        normed = None if pre else torch.empty(h.shape, dtype=torch.float32, device=dev)
        L.check(L.load().mojo_hip_residual_add_rmsnorm_quant(
            L.ptr(h), L.ptr(r), L.ptr(weight), L.ptr(smooth), L.ptr(out), L.ptr(summed), L.ptr(normed), L.ptr(scale), rows, dim,
            L.dtype_code(h.dtype), L.dtype_code(self.quant_dtype), float(self.q_min), float(self.variance_epsilon),
            L.stream_of(h)), "HIPResidualAddRMSNormQuant")
        return out, (summed if pre else normed), scale


__all__ = ["HIPDynamicQuant", "HIPResidualAddRMSNormQuant"]
