"""API class `MojoSwiGLU` (`mojo_opset/core/operators/activation.py:38-66`)."""
from ..operator import MojoOperator


class MojoSwiGLU(MojoOperator):
    """forward(gate_out, up_out) -> silu(gate_out) * up_out.

    ``swiglu_limit > 0`` first clamps ``up_out`` to ``[-L, L]`` and ``gate_out`` to ``<= L``.
    """

    def __init__(self, swiglu_limit: float = 0.0, **kwargs):
        super().__init__(**kwargs)
        self.swiglu_limit = swiglu_limit

    def extra_repr(self) -> str:
        return f"swiglu_limit={self.swiglu_limit!r}"
