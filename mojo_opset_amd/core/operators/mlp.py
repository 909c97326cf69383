"""API class `MojoSwiGLUMLP` — the reference's hook for a gated MLP (`mojo_opset/core/operators/mlp.py:7-37`): the decode-sized
fused form of this backend (gate|up projection with SwiGLU in its epilogue, then the down projection) binds to it."""
import torch

from ..operator import MojoOperator


class MojoSwiGLUMLP(MojoOperator):
    """forward(x [..., input_size]) -> [..., output_size]: ``fc2(silu(a1) * a2)`` with ``a1, a2 = fc1(x).chunk(2, -1)``.

    ``fc1 = Linear(input_size, 2 * hidden_size, bias=False)`` (gate rows first, then up rows), ``fc2 = Linear(hidden_size,
    output_size, bias=False)``; state_dict keys ``fc1.weight`` / ``fc2.weight``."""

    def __init__(self, input_size: int, output_size: int, hidden_size: int):
        super().__init__()
        self.fc1 = torch.nn.Linear(input_size, hidden_size * 2, bias=False)
        self.fc2 = torch.nn.Linear(hidden_size, output_size, bias=False)

    def extra_repr(self) -> str:
        return f"input_size={self.fc1.in_features}, output_size={self.fc2.out_features}, hidden_size={self.fc2.in_features}"
