"""API classes of the activation quantisers that feed `MojoQuantGemm`.

Follows `mojo_opset/core/operators/quantize.py:120-172` (`MojoDynamicQuant`) and
`mojo_opset/core/operators/normalization.py:434-533` (`MojoResidualAddRMSNormQuant`).
"""
from typing import Optional

import torch

from ..operator import MojoOperator


class MojoDynamicQuant(MojoOperator):
    """forward(input [*, K]) -> (int8 [*, K], fp32 scale [*, 1]): per-token symmetric quantisation,
    ``scale = max(amax|x * inv_smooth_scale|, 1e-12) / 127`` (``1.0`` where that is below 1e-6),
    ``q = clamp(round_half_even(x / scale), -128, 127)``.  ``inv_smooth_scale [K]`` fp32 is optional (``input_size=None``)."""

    def __init__(self, input_size: Optional[int] = None, quant_dtype: torch.dtype = torch.int8, **kwargs):
        super().__init__(**kwargs)
        self.input_size = input_size
        if input_size is None:
            self.register_parameter("inv_smooth_scale", None)
        else:
            self.inv_smooth_scale = torch.nn.Parameter(torch.empty(input_size, **self.tensor_factory_kwargs))
            setattr(self.inv_smooth_scale, "force_dtype", torch.float32)
        self.quant_dtype = quant_dtype
        if quant_dtype != torch.int8:
            raise NotImplementedError(f"Unsupported quant_dtype: {quant_dtype}, expected torch.int8.")
        self.q_max = 127
        self.q_min = -128

    def extra_repr(self) -> str:
        return f"input_size={self.input_size}, quant_dtype={self.quant_dtype}"


class MojoResidualAddRMSNormQuant(MojoOperator):
    """forward(hidden_state, residual, smooth_scale=None) -> (quant_output, residual_out, scale [*, 1]).

    ``pre``: ``residual_out = hidden + residual`` (input dtype); ``post``: ``residual_out`` is the fp32 normed tensor.
    ``normed = rms_norm(sum.float(), weight, eps)`` stays fp32; ``scale = max(amax|normed * smooth|, 1e-12) / q_max``;
    ``q = clamp(round_half_even(normed / scale), q_min, q_max)`` cast to int8 or float8_e4m3fn.
    """

    def __init__(self, norm_size: int, eps: float = 1e-5, norm_pos: str = "pre", quant_dtype: torch.dtype = torch.int8,
                 symmetric: bool = True, **kwargs):
        super().__init__(**kwargs)
        if norm_pos not in ("pre", "post"):
            raise ValueError("norm_pos should be 'pre' or 'post'")
        self.norm_size = norm_size
        self.variance_epsilon = float(eps)
        self.norm_pos = norm_pos
        self.weight = torch.nn.Parameter(torch.empty(norm_size, **self.tensor_factory_kwargs))
        self.quant_dtype = quant_dtype
        self.symmetric = symmetric
        if quant_dtype == torch.int8:
            self.q_max = 127
            self.q_min = -128 if symmetric else 0
        elif quant_dtype == torch.float8_e4m3fn:
            self.q_max = torch.finfo(torch.float8_e4m3fn).max
            self.q_min = -torch.finfo(torch.float8_e4m3fn).max
        else:
            raise NotImplementedError(
                f"Unsupported quant_dtype: {quant_dtype}, "
                f"expected torch.int8 or torch.float8_e4m3fn"
            )

    def extra_repr(self) -> str:
        return (
            f"norm_size={self.norm_size}, variance_epsilon={self.variance_epsilon}, "
            f"norm_pos={self.norm_pos!r}, quant_dtype={self.quant_dtype}, "
            f"symmetric={self.symmetric}"
        )


__all__ = ["MojoDynamicQuant", "MojoResidualAddRMSNormQuant"]
