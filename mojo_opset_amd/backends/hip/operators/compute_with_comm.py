"""HIP<Op> classes of the GEMM + collective operators: the C-ABI GEMM plugged into `mojo_opset_amd.comm`."""

import torch

from ....comm import GemmEngine, all_gather_gemm, gemm_all2all, gemm_all_reduce, gemm_reduce_scatter
from ....comm import peer, select
from ....core.operators.compute_with_comm import (MojoAllGatherGemm, MojoGemmAll2All, MojoGemmAllReduce,
                                                  MojoGemmReduceScatter, is_dist_initialized)
from .. import lib as L

_ROCM = ["rocm"]


class HipGemmEngine(GemmEngine):
    """`mojo_hip_gemm_rowmap` on the current stream."""

    def __call__(self, x, weight, bias, trans_weight, *, out=None, rows=None, a_map=None, c_map=None):
        L.require_cuda(x, weight, bias, out)
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
        if x.stride(-1) != 1:
            x = x.contiguous()
        rows = x.shape[0] if rows is None else rows
        if out is None:
            out = torch.empty(rows, n, dtype=x.dtype, device=x.device)
        assert out.dim() == 2 and out.shape[1] == n and out.stride(1) == 1 and out.dtype == x.dtype
        ws = torch.empty(L.load().mojo_hip_gemm_workspace_bytes(rows, k, n), dtype=torch.uint8, device=x.device)
        amap = None if a_map is None else L.strides3(*a_map)
        cmap = None if c_map is None else L.strides3(*c_map)
        L.check(L.load().mojo_hip_gemm_rowmap(
            L.ptr(x), L.ptr(weight), L.ptr(None if bias is None else bias.contiguous()), L.ptr(out), rows, k, n,
            x.stride(0), out.stride(0), w_k, w_n, amap, cmap, L.dtype_code(x.dtype), L.ptr(ws), ws.numel(),
            L.stream_of(x)), "hip gemm")
        return out


_ENGINE = HipGemmEngine()


def _group_of(op):
    return op._group() if is_dist_initialized() else None


def _exchange(group, op_name, payload_bytes, x2, direct_fn, pipeline_fn):
    """Run the operator on the exchange `comm.select` picks for this (group, operator, payload): the direct peer exchange or
    the collective-library pipeline (forced by MOJO_HIP_COMM_DIRECT=1/0, otherwise self-tested and timed once per key)."""
    if select.choose(group, op_name, payload_bytes, x2, direct_fn, pipeline_fn) == "direct":
        return direct_fn()
    return pipeline_fn()


class HIPGemmAllReduce(MojoGemmAllReduce):
    supported_platforms_list = _ROCM

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        group = _group_of(self)
        if group is not None and input.is_cuda:
            x2 = input.reshape(-1, input.shape[-1])
            x2 = x2 if x2.stride(-1) == 1 else x2.contiguous()
            n = _ENGINE.out_features(self.weight, self.trans_weight)
            if peer.direct_supported(x2, n, x2.shape[0]):
                return _exchange(
                    group, "gemm_all_reduce", x2.shape[0] * n * x2.element_size(), x2,
                    lambda: peer.gemm_all_reduce_direct(_ENGINE, x2, self.weight, self.bias, self.trans_weight, group).reshape(*input.shape[:-1], n),
                    lambda: gemm_all_reduce(_ENGINE, input, self.weight, self.bias, self.trans_weight, group))
        return gemm_all_reduce(_ENGINE, input, self.weight, self.bias, self.trans_weight, group)


class HIPAllGatherGemm(MojoAllGatherGemm):
    supported_platforms_list = _ROCM

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        group = _group_of(self)
        if group is not None and input.is_cuda and self.gather_dim % input.dim() == 0 and input.dim() >= 2:
            import torch.distributed as dist

            x2 = input.reshape(-1, input.shape[-1])
            x2 = x2 if x2.stride(-1) == 1 and x2.stride(0) == x2.shape[1] else x2.contiguous()
            n = _ENGINE.out_features(self.weight, self.trans_weight)
            if x2.shape[0] > 0 and (x2.shape[1] * x2.element_size()) % 16 == 0 and x2.dtype in (torch.bfloat16, torch.float16, torch.float32):
                shape = list(input.shape[:-1]) + [n]
                shape[0] *= dist.get_world_size(group)
                return _exchange(
                    group, "all_gather_gemm", x2.numel() * x2.element_size() * dist.get_world_size(group), x2,
                    lambda: peer.all_gather_gemm_direct(_ENGINE, x2, self.weight, self.bias, self.trans_weight, group).reshape(shape),
                    lambda: all_gather_gemm(_ENGINE, input, self.weight, self.bias, self.trans_weight, group, self.gather_dim))
        return all_gather_gemm(_ENGINE, input, self.weight, self.bias, self.trans_weight, group, self.gather_dim)


class HIPGemmAll2All(MojoGemmAll2All):
    supported_platforms_list = _ROCM

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return gemm_all2all(_ENGINE, input, self.weight, self.bias, self.trans_weight, _group_of(self),
                            self.scatter_dim, self.gather_dim)


class HIPGemmReduceScatter(MojoGemmReduceScatter):
    supported_platforms_list = _ROCM

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        group = _group_of(self)
        if group is not None and input.is_cuda and self.scatter_dim % input.dim() == 0:
            import torch.distributed as dist

            ws = dist.get_world_size(group)
            x2 = input.reshape(-1, input.shape[-1])
            x2 = x2 if x2.stride(-1) == 1 else x2.contiguous()
            n = _ENGINE.out_features(self.weight, self.trans_weight)
            if input.shape[0] % ws == 0 and peer.direct_supported(x2, n, x2.shape[0] // ws):
                shape = list(input.shape[:-1]) + [n]
                shape[0] //= ws
                return _exchange(
                    group, "gemm_reduce_scatter", x2.shape[0] * n * x2.element_size(), x2,
                    lambda: peer.gemm_reduce_scatter_direct(_ENGINE, x2, self.weight, self.bias, self.trans_weight, group).reshape(shape),
                    lambda: gemm_reduce_scatter(_ENGINE, input, self.weight, self.bias, self.trans_weight, group, self.scatter_dim))
        return gemm_reduce_scatter(_ENGINE, input, self.weight, self.bias, self.trans_weight, group, self.scatter_dim)
