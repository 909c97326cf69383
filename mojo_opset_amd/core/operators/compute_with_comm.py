"""API classes of the four GEMM+collective ops.

Follows `mojo_opset/core/operators/compute_with_comm.py` (`MojoGemmAllReduce` :57-116,
`MojoAllGatherGemm` :119-184, `MojoGemmAll2All` :187-261, `MojoGemmReduceScatter` :264-340).
``trans_weight=False`` means ``weight [N, K_local]`` (`F.linear` layout); ``True`` means
``[K_local, N]``.  ``weight``/``bias`` are plain attributes.  With `torch.distributed`
uninitialised every collective is the identity.
"""
from typing import Optional

import torch
import torch.distributed as dist

from ..operator import MojoOperator


def is_dist_initialized() -> bool:
    return dist.is_available() and dist.is_initialized()


class _GemmCommBase:
    def _init_gemm_comm(self, weight, bias, trans_weight, process_group):
        if not isinstance(trans_weight, bool):
            raise TypeError("trans_weight must be bool.")
        self.weight = weight
        self.bias = bias
        self.trans_weight = trans_weight
        self.process_group = process_group

    def _group(self):
        if self.process_group is not None:
            return self.process_group
        return dist.distributed_c10d._get_default_group()

    def _base_repr(self) -> str:
        w = tuple(self.weight.shape) if isinstance(self.weight, torch.Tensor) else None
        return f"weight_shape={w}, has_bias={self.bias is not None}, trans_weight={self.trans_weight}"

    def extra_repr(self) -> str:
        return self._base_repr()


class MojoGemmAllReduce(_GemmCommBase, MojoOperator):
    """forward(input [*, K_local]) -> allreduce_sum(input @ W (+ bias)) [*, N].
    Golden semantics: the bias is added on every rank *before* the reduce (:106-110)."""

    def __init__(self, weight, bias: Optional[torch.Tensor] = None, trans_weight: bool = False, process_group=None):
        super().__init__()
        self._init_gemm_comm(weight, bias, trans_weight, process_group)


class MojoAllGatherGemm(_GemmCommBase, MojoOperator):
    """forward(input local shard) -> allgather(input, gather_dim) @ W (+ bias)."""

    def __init__(self, weight, bias: Optional[torch.Tensor] = None, trans_weight: bool = False,
                 process_group=None, gather_dim: int = 0):
        super().__init__()
        self._init_gemm_comm(weight, bias, trans_weight, process_group)
        self.gather_dim = gather_dim

    def extra_repr(self) -> str:
        return f"{self._base_repr()}, gather_dim={self.gather_dim}"


class MojoGemmAll2All(_GemmCommBase, MojoOperator):
    """forward(input) -> cat(all_to_all(chunk(input @ W (+bias), ws, scatter_dim)), gather_dim)."""

    def __init__(self, weight, bias: Optional[torch.Tensor] = None, trans_weight: bool = False,
                 process_group=None, scatter_dim: int = 0, gather_dim: int = 1):
        super().__init__()
        self._init_gemm_comm(weight, bias, trans_weight, process_group)
        self.scatter_dim = scatter_dim
        self.gather_dim = gather_dim

    def extra_repr(self) -> str:
        return f"{self._base_repr()}, scatter_dim={self.scatter_dim}, gather_dim={self.gather_dim}"


class MojoGemmReduceScatter(_GemmCommBase, MojoOperator):
    """forward(input) -> this rank's ``scatter_dim`` chunk of sum_ranks(input @ W (+ bias))."""

    def __init__(self, weight, bias: Optional[torch.Tensor] = None, trans_weight: bool = False,
                 process_group=None, scatter_dim: int = 0):
        super().__init__()
        self._init_gemm_comm(weight, bias, trans_weight, process_group)
        self.scatter_dim = scatter_dim

    def extra_repr(self) -> str:
        return f"{self._base_repr()}, scatter_dim={self.scatter_dim}"
