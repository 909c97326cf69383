"""API classes `MojoGemm` / `MojoGroupGemm` / `MojoQuantGemm` (`MojoGemm` is the reference's hook for the
decode-sized dense products).

Follows `mojo_opset/core/operators/gemm.py` (:12-56, :59-124, :127-231).
"""
import math
from typing import Optional, Union

import torch
import torch.nn as nn

from ..operator import MojoOperator


class MojoGemm(MojoOperator):
    """forward(input [..., in_features]) -> [..., out_features] = ``F.linear(input, weight, bias)``.

    Either ``(in_features, out_features, bias=True)`` — Parameters ``weight [out, in]`` and ``bias [out]`` created with the
    tensor factory kwargs and initialised like ``nn.Linear`` — or ``weight=<2-D tensor>`` alone (wrapped as the Parameter,
    no bias).  ValueErrors as the reference raises them (`gemm.py:22-34`)."""

    def __init__(self, in_features: Optional[int] = None, out_features: Optional[int] = None, bias: bool = True,
                 weight: Optional[torch.Tensor] = None, **kwargs):
        super().__init__(**kwargs)
        if weight is not None:
            if in_features is not None or out_features is not None:
                raise ValueError("Provide either weight or in_features/out_features, not both.")
            if weight.dim() != 2:
                raise ValueError(f"weight must be 2D, got shape {tuple(weight.shape)}.")
            self.out_features, self.in_features = weight.shape
            self.weight = nn.Parameter(weight)
            self.register_parameter("bias", None)
            return
        if in_features is None or out_features is None:
            raise ValueError("in_features and out_features are required when weight is not provided.")
        self.in_features = in_features
        self.out_features = out_features
        self.weight = nn.Parameter(torch.empty((out_features, in_features), **self.tensor_factory_kwargs))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_features, **self.tensor_factory_kwargs))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self) -> None:
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = nn.init._calculate_fan_in_and_fan_out(self.weight)
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            nn.init.uniform_(self.bias, -bound, bound)

    def extra_repr(self) -> str:
        return f"in_features={self.in_features}, out_features={self.out_features}, bias={self.bias is not None}"


class MojoGroupGemm(MojoOperator):
    """forward(input [M,K], group_list [G] row counts) -> [M,N].

    ``weight`` is a plain attribute (not a Parameter): ``[G,K,N]``, or ``[G,N,K]`` when
    ``trans_weight`` is set.  Rows ``[s_g, e_g)`` of the input are multiplied by ``weight[g]``.
    """

    def __init__(self, weight, trans_weight=False):
        super().__init__()
        self.weight = weight
        self.trans_weight = trans_weight

    def check_call_contract(self, input, group_list):
        assert input.dim() == 2, "input must be 2D"
        assert self.weight.dim() == 3, "weight must be 3D"
        groups = group_list.numel()
        assert self.weight.size(0) == groups, "weight group count must match group_list length"
        k_w = self.weight.shape[2] if self.trans_weight else self.weight.shape[1]
        assert k_w == input.shape[1], "K of input should be equal to K of self.weight."

    def extra_repr(self) -> str:
        w = self.weight if isinstance(self.weight, torch.Tensor) else None
        return (f"weight_shape={tuple(w.shape) if w is not None else None}, "
                f"weight_dtype={w.dtype if w is not None else None}, trans_weight={self.trans_weight}")


class MojoQuantGemm(MojoOperator):
    """forward(input i8 [M,K], input_scale [M] or [M,1]) -> [M,N] in ``output_dtype``:
    ``(input @ weight) * input_scale[m] * weight_scale[n]`` with exact integer accumulation.

    Buffers: ``weight`` ``(K,N)`` (``(N,K)`` if ``trans_weight``) and bf16 ``weight_scale [N]``.
    The reference accepts int8 only (:171,:173).  This build additionally accepts
    ``torch.float8_e4m3fn`` for both (BASELINE config 5); that path has no reference
    implementation — **parity unpinned**, pinned only by this repo's own CPU restatement.
    """

    _QUANT_DTYPES = (torch.int8, torch.float8_e4m3fn)

    def __init__(self, in_features: int, out_features: int, output_dtype: torch.dtype = torch.bfloat16,
                 trans_weight: bool = False, quant_dtype: torch.dtype = torch.int8,
                 weight_dtype: Union[str, torch.dtype] = torch.int8, **kwargs):
        super().__init__(**kwargs)
        self.in_features = in_features
        self.out_features = out_features
        self.weight_shape = (out_features, in_features) if trans_weight else (in_features, out_features)
        self.quant_dtype = quant_dtype
        assert quant_dtype in self._QUANT_DTYPES, (
            f"GemmDequant only support int8 (and fp8-e4m3 in this build) quantization yet, but get {quant_dtype=}"
        )
        self.weight_dtype = weight_dtype
        assert weight_dtype == quant_dtype, (
            f"GemmDequant weight dtype must equal the activation quant dtype, but get {weight_dtype=}"
        )
        self.register_buffer("weight", torch.empty(self.weight_shape, **{**self.tensor_factory_kwargs, "dtype": quant_dtype}))
        self.register_buffer("weight_scale", torch.empty(out_features, **{**self.tensor_factory_kwargs, "dtype": torch.bfloat16}))
        self.output_dtype = output_dtype
        self.trans_weight = trans_weight

    def check_call_contract(self, input, input_scale):
        if input.dim() != 2:
            raise ValueError(f"input must be 2D, got shape {tuple(input.shape)}.")
        if self.weight.dim() != 2:
            raise ValueError(f"weight must be 2D, got shape {tuple(self.weight.shape)}.")
        if input.shape[-1] != self.in_features:
            raise ValueError(f"input K {input.shape[-1]} must match weight K {self.in_features}.")
        if self.weight_scale.shape != (self.out_features,):
            raise ValueError(
                f"weight_scale shape {tuple(self.weight_scale.shape)} must match output dim {(self.out_features,)}."
            )

    def extra_repr(self) -> str:
        return (f"in_features={self.in_features}, out_features={self.out_features}, "
                f"output_dtype={self.output_dtype}, trans_weight={self.trans_weight}, "
                f"quant_dtype={self.quant_dtype}, weight_dtype={self.weight_dtype}")
