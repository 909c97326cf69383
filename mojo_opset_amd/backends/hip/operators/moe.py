"""HIP<Op> classes of the MoE routing ops: gating, dispatch, experts, combine.
No host synchronisation anywhere: token counts stay on the device and feed the grouped GEMM as they are."""

import torch

from ....core.operators.moe import MojoExperts, MojoMoE, MojoMoECombine, MojoMoEDispatch, MojoMoEGating
from .... import switches
from .. import lib as L

_ROCM = ["rocm"]


def _dense(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


class HIPMoEGating(MojoMoEGating):
    supported_platforms_list = _ROCM

    def forward(self, hidden_states: torch.Tensor):
        self.check_call_contract(hidden_states)
        w = self.gate_weight.detach()
        L.require_cuda(hidden_states, w)
        x, w = _dense(hidden_states), _dense(w)
        tokens, hidden = x.shape
        experts = w.shape[1]
        idx = torch.empty(tokens, self.top_k, dtype=torch.int32, device=x.device)
        gates = torch.empty(tokens, self.top_k, dtype=torch.float32, device=x.device)
        lib = L.load()
        code = L.dtype_code(x.dtype)
        ws = torch.empty(lib.mojo_hip_moe_gating_workspace_bytes(tokens, hidden, experts, code), dtype=torch.uint8, device=x.device)
        L.check(lib.mojo_hip_moe_gating(L.ptr(x), L.ptr(w), L.ptr(idx), L.ptr(gates), tokens, hidden, experts, self.top_k, code,
                                        L.ptr(ws), ws.numel(), L.stream_of(x)), "HIPMoEGating")
        return idx, gates


class HIPMoEDispatch(MojoMoEDispatch):
    supported_platforms_list = _ROCM

    def forward(self, hidden_states: torch.Tensor, top_k_gates: torch.Tensor, top_k_indices: torch.Tensor):
        self.check_call_contract(hidden_states, top_k_gates, top_k_indices)
        L.require_cuda(hidden_states, top_k_gates, top_k_indices)
        if hidden_states.dim() != 2 or top_k_indices.dim() != 2 or top_k_gates.shape != top_k_indices.shape:
            raise NotImplementedError("HIPMoEDispatch: hidden_states [T,H], top_k_gates / top_k_indices [T,k] expected")
        x, g, ids = _dense(hidden_states), _dense(top_k_gates), _dense(top_k_indices)
        tokens, hidden = x.shape
        k = ids.shape[1]
        assert ids.shape[0] == tokens
        n = tokens * k
        dev = x.device
        sorted_hidden = torch.empty(n, hidden, dtype=x.dtype, device=dev)
        per_expert = torch.empty(self.num_experts, dtype=torch.int32, device=dev)
        sorted_gates = torch.empty(n, 1, dtype=torch.float32, device=dev)
        token_indices = torch.empty(n, dtype=torch.int32, device=dev)
        lib = L.load()
        ws = torch.empty(lib.mojo_hip_moe_dispatch_workspace_bytes(n, self.num_experts), dtype=torch.uint8, device=dev)
        L.check(lib.mojo_hip_moe_dispatch(L.ptr(x), L.ptr(g), L.ptr(ids), L.ptr(sorted_hidden), L.ptr(per_expert),
                                          L.ptr(sorted_gates), L.ptr(token_indices), tokens, hidden, k, self.num_experts,
                                          L.dtype_code(x.dtype), L.ptr(ws), ws.numel(), L.stream_of(x)), "HIPMoEDispatch")
        return sorted_hidden, per_expert, sorted_gates, token_indices


class HIPMoECombine(MojoMoECombine):
    supported_platforms_list = _ROCM

    def forward(self, output_buffer: torch.Tensor, expert_outputs: torch.Tensor, sorted_gates: torch.Tensor,
                token_indices: torch.Tensor) -> torch.Tensor:
        L.require_cuda(output_buffer, expert_outputs, sorted_gates, token_indices)
        if output_buffer.dim() != 2 or expert_outputs.dim() != 2 or expert_outputs.shape[1] != output_buffer.shape[1]:
            raise NotImplementedError("HIPMoECombine: output_buffer [T,H] and expert_outputs [N,H] expected")
        rows_t = _dense(expert_outputs)
        n, hidden = rows_t.shape
        tokens = output_buffer.shape[0]
        tok = _dense(token_indices.to(torch.int32))
        assert tok.numel() == n
        gates = None
        if self.multiply_by_gates:
            gates = _dense(sorted_gates.to(torch.float32)).reshape(-1)
            assert gates.numel() == n, "sorted_gates must hold one value per expert-output row"
        out = torch.empty(tokens, hidden, dtype=rows_t.dtype, device=rows_t.device)
        lib = L.load()
        ws = torch.empty(lib.mojo_hip_moe_combine_workspace_bytes(tokens, n), dtype=torch.uint8, device=rows_t.device)
        L.check(lib.mojo_hip_moe_combine(L.ptr(rows_t), L.ptr(gates), L.ptr(tok), L.ptr(out), tokens, n, hidden,
                                         L.dtype_code(rows_t.dtype), L.ptr(ws), ws.numel(), L.stream_of(rows_t)),
                "HIPMoECombine")
        return out


class HIPExperts(MojoExperts):
    """grouped GEMM -> SwiGLU on the two halves in place -> grouped GEMM, with the reference's weight layouts
    (`up_proj_weight [E, 2I, H]`, `down_proj_weight [E, H, I]`: both ``[G, N, K]``)."""

    supported_platforms_list = _ROCM

    def _group_gemm(self, x, w, counts):
        m, k = x.shape
        groups, n = w.shape[0], w.shape[1]
        out = torch.empty(m, n, dtype=x.dtype, device=x.device)
        lib = L.load()
        ws = torch.empty(lib.mojo_hip_group_gemm_workspace_bytes(groups), dtype=torch.uint8, device=x.device)
        L.check(lib.mojo_hip_group_gemm(L.ptr(x), L.ptr(w), L.ptr(out), L.ptr(counts),
                                        1 if counts.dtype == torch.int64 else 0, m, k, n, groups, 1,
                                        L.dtype_code(x.dtype), L.ptr(ws), ws.numel(), L.stream_of(x)), "HIPExperts gemm")
        return out

    def _fused_up_swiglu(self, x, w, counts, act, inter) -> bool:
        """First projection with the SwiGLU applied to the accumulators (same rounding points as the two-kernel path: the
        [M, 2I] product never goes to HBM).  False when the shape is outside the fused kernel's preconditions."""
        if switches.get("MOJO_HIP_EXPERTS_FUSED", "1") == "0" or x.dtype not in (torch.bfloat16, torch.float16) or x.shape[0] == 0:
            return False
        # A decode step (at most 64 rows per expert on average): the projections are weight streams, and the grouped GEMM
        # has a 64-row streaming form for ragged groups (csrc/gemm_skinny.hip, RAGGED) that the fused 256-row tile kernel
        # cannot use; the [M, 2I] round trip of the two-kernel path is a few hundred KB there.
        if (x.shape[0] <= 64 * w.shape[0] and x.shape[1] % 128 == 0 and w.shape[1] % 64 == 0 and w.shape[0] >= 2
                and switches.get_int("MOJO_HIP_GEMM_SKINNY", 31) & 2):        # (bit 2 = the ragged skinny form, csrc/gemm.h)
            return False
        lib = L.load()
        groups = w.shape[0]
        ws = torch.empty(lib.mojo_hip_group_gemm_workspace_bytes(groups), dtype=torch.uint8, device=x.device)
        rc = lib.mojo_hip_group_gemm_swiglu(L.ptr(x), L.ptr(w), L.ptr(act), L.ptr(counts),
                                            1 if counts.dtype == torch.int64 else 0, x.shape[0], x.shape[1], inter, groups, 1,
                                            L.dtype_code(x.dtype), L.ptr(ws), ws.numel(), L.stream_of(x))
        if rc == L.MOJO_EUNSUPPORTED:
            return False
        L.check(rc, "HIPExperts fused gemm + swiglu")
        return True

    def forward(self, sorted_hidden_states: torch.Tensor, tokens_per_expert: torch.Tensor) -> torch.Tensor:
        up_w, down_w = self.up_proj_weight.detach(), self.down_proj_weight.detach()
        L.require_cuda(sorted_hidden_states, up_w, down_w)
        if sorted_hidden_states.dtype != up_w.dtype or up_w.dtype != down_w.dtype:
            raise NotImplementedError("HIPExperts: activations and both weights must share one dtype")
        if tokens_per_expert.dtype not in (torch.int32, torch.int64):
            raise NotImplementedError("HIPExperts: tokens_per_expert must be int32 or int64")
        x = _dense(sorted_hidden_states)
        counts = _dense(tokens_per_expert.to(x.device, non_blocking=True))
        assert counts.numel() == up_w.shape[0]
        inter = down_w.shape[2]
        act = torch.empty(x.shape[0], inter, dtype=x.dtype, device=x.device)
        if self._fused_up_swiglu(x, _dense(up_w), counts, act, inter):                # SwiGLU in the GEMM epilogue
            return self._group_gemm(act, _dense(down_w), counts)
        fc1 = self._group_gemm(x, _dense(up_w), counts)                               # [M, 2I] = [gate | up]
        L.check(L.load().mojo_hip_swiglu_rows(L.ptr(fc1), L.c_void_p(fc1.data_ptr() + inter * fc1.element_size()),
                                              L.ptr(act), x.shape[0], inter, 2 * inter, 2 * inter, inter,
                                              L.dtype_code(x.dtype), 0.0, L.stream_of(x)), "HIPExperts swiglu")
        return self._group_gemm(act, _dense(down_w), counts)


class HIPMoE(MojoMoE):
    """The four HIP stages chained; with ``ep_size == 1`` nothing synchronises with the host."""

    supported_platforms_list = _ROCM

    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        # called through the class so the same body also binds to the reference's MojoMoE (plugin.rebase_hip_backend),
        # which carries the same attributes but not this helper
        return MojoMoE.compose_forward(self, hidden_states)


__all__ = ["HIPMoEGating", "HIPMoEDispatch", "HIPMoECombine", "HIPExperts", "HIPMoE"]
