"""API classes `MojoRMSNorm` / `MojoResidualAddRMSNorm` and `MojoRMSNormInplace` (§8 f3).

Follows `mojo_opset/core/operators/normalization.py` (:71-111, :308-362): one learnable
``weight [norm_size]`` created with the factory kwargs; ``norm_pos`` in {"pre","post"}; and
`mojo_opset/experimental/operators/normalization.py:95-140` for the in-place variant.
"""
import torch

from ..operator import MojoOperator


class MojoRMSNorm(MojoOperator):
    """forward(hidden_state [..., D]) -> same shape/dtype."""

    def __init__(self, norm_size: int, eps: float = 1e-5, **kwargs):
        super().__init__(**kwargs)
        self.norm_size = norm_size
        self.weight = torch.nn.Parameter(torch.empty(norm_size, **self.tensor_factory_kwargs))
        self.variance_epsilon = eps

    def extra_repr(self) -> str:
        return f"norm_size={self.norm_size!r}, variance_epsilon={self.variance_epsilon!r}"


class MojoRMSNormInplace(MojoOperator):
    """forward(hidden_state [..., D]) -> rms_norm(hidden_state); with ``inplace=True`` the result is written back into
    ``hidden_state`` and that same tensor is returned (`experimental/operators/normalization.py:95-140`; the q/k-norm in
    front of RoPE in the Qwen3 stack, `modeling/qwen3/mojo_qwen3_dense.py:229-233`)."""

    def __init__(self, norm_size: int, eps: float = 1e-5, inplace: bool = False, **kwargs):
        super().__init__(**kwargs)
        self.norm_size = norm_size
        self.weight = torch.nn.Parameter(torch.empty(norm_size, **self.tensor_factory_kwargs))
        self.variance_epsilon = eps
        self.inplace = inplace

    def extra_repr(self) -> str:
        return f"norm_size={self.norm_size!r}, variance_epsilon={self.variance_epsilon!r}, inplace={self.inplace!r}"


class MojoResidualAddRMSNorm(MojoOperator):
    """forward(hidden_state, residual) -> (normed, residual_out).

    ``pre``: ``residual_out = hidden + residual`` (rounded to the input dtype) and
    ``normed = rms_norm(residual_out)``; ``post``: ``normed = rms_norm(hidden + residual)`` and
    ``residual_out = normed``.
    """

    def __init__(self, norm_size: int, eps: float = 1e-05, norm_pos: str = "pre", **kwargs):
        super().__init__(**kwargs)
        if norm_pos not in ("pre", "post"):
            raise ValueError("norm_pos should be 'pre' or 'post'")
        self.norm_size = norm_size
        self.variance_epsilon = float(eps)
        self.weight = torch.nn.Parameter(torch.empty(norm_size, **self.tensor_factory_kwargs))
        self.norm_pos = norm_pos

    def extra_repr(self) -> str:
        return (f"norm_size={self.norm_size!r}, variance_epsilon={self.variance_epsilon!r}, "
                f"norm_pos={self.norm_pos!r}")
