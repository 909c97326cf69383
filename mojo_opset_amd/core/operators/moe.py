"""API classes of the MoE routing ops either side of `MojoGroupGemm`.

Follows `mojo_opset/core/operators/moe.py`: `MojoMoEGating` (:277-321), `MojoMoEDispatch` (:327-400),
`MojoExperts` (:402-449), `MojoMoECombine` (:670-716).  Constructor signatures, parameter names
and shapes, dtype asserts and return order are the reference's; the forward is abstract here.
"""
import torch

from ..operator import MojoOperator


class MojoMoEGating(MojoOperator):
    """forward(hidden_states [T, hidden]) -> (top_k_indices int32 [T, k], top_k_gates fp32 [T, k]).

    ``softmax(hidden.float() @ gate_weight)`` over all experts, top-k, gates renormalised to sum 1.
    ``gate_weight [hidden, num_experts]`` is an fp32 parameter (marked ``force_dtype`` like the reference).
    """

    def __init__(self, hidden_size: int, num_experts: int, top_k: int, **kwargs):
        super().__init__(**kwargs)
        self.gate_weight = torch.nn.Parameter(torch.empty(hidden_size, num_experts, **self.tensor_factory_kwargs))
        self.top_k = top_k
        setattr(self.gate_weight, "force_dtype", torch.float32)

    def check_call_contract(self, hidden_states):
        assert self.gate_weight.dtype == torch.float32
        assert hidden_states.dim() == 2 and hidden_states.shape[1] == self.gate_weight.shape[0]
        assert 0 < self.top_k <= self.gate_weight.shape[1]

    def extra_repr(self) -> str:
        hidden_size = self.gate_weight.size(0)
        num_experts = self.gate_weight.size(1)
        return f"{hidden_size=}, {num_experts=}, {self.top_k=}".replace("self.", "")


class MojoMoEDispatch(MojoOperator):
    """forward(hidden_states [T, H], top_k_gates fp32 [T, k], top_k_indices int32 [T, k]) ->
    (sorted_hidden_states [T*k, H], tokens_per_expert int32 [E], sorted_gates fp32 [T*k, 1], token_indices int32 [T*k]).

    Rows are bucketed by expert id (ascending).  The order INSIDE a bucket is explicitly not part of the contract
    (reference :367-372): consumers treat a bucket as an unordered set.
    """

    def __init__(self, num_experts: int, **kwargs):
        super().__init__(**kwargs)
        self.num_experts = num_experts

    def check_call_contract(self, hidden_states, top_k_gates, top_k_indices):
        assert top_k_gates.dtype == torch.float32, (
            f"MojoMoEDispatch: top_k_gates must be float32, got {top_k_gates.dtype}."
        )
        assert top_k_indices.dtype == torch.int32, (
            f"MojoMoEDispatch: top_k_indices must be int32, got {top_k_indices.dtype}."
        )


class MojoExperts(MojoOperator):
    """forward(sorted_hidden_states [M, hidden], tokens_per_expert [E]) -> [M, hidden]:
    per expert ``down(silu(gate) * up)`` with ``[gate | up] = x @ up_proj_weight[e].T``.

    ``up_proj_weight [E, 2*inter, hidden]``, ``down_proj_weight [E, hidden, inter]``.
    """

    def __init__(self, num_experts: int, hidden_size: int, intermediate_size: int, activation: str = "swiglu", **kwargs):
        super().__init__(**kwargs)
        if activation != "swiglu":
            raise NotImplementedError(f"MojoExperts: Activation {activation} is not supported.")
        self.activation = activation
        self.up_proj_weight = torch.nn.Parameter(
            torch.empty(num_experts, intermediate_size * 2, hidden_size, **self.tensor_factory_kwargs))
        self.down_proj_weight = torch.nn.Parameter(
            torch.empty(num_experts, hidden_size, intermediate_size, **self.tensor_factory_kwargs))


class MojoMoECombine(MojoOperator):
    """forward(output_buffer [T, H], expert_outputs [N, H], sorted_gates [N, 1], token_indices [N]) -> [T, H]:
    ``out[t] = sum_{j: token_indices[j] == t} expert_outputs[j] (* sorted_gates[j])`` accumulated in fp32 from zero
    (the buffer only lends its shape), returned in ``expert_outputs.dtype``.
    """

    def __init__(self, multiply_by_gates: bool = True, **kwargs):
        super().__init__(**kwargs)
        self.multiply_by_gates = multiply_by_gates


def count_expert_tokens(top_k_indices: torch.Tensor, num_experts: int) -> torch.Tensor:
    """int32 histogram of the routed expert ids (reference `_count_expert_tokens` :323-325)."""
    flat = top_k_indices.reshape(-1).to(dtype=torch.int64, device=top_k_indices.device)
    return torch.bincount(flat, minlength=num_experts).to(dtype=torch.int32, device=top_k_indices.device)


class MojoMoE(MojoOperator):
    """forward(hidden_states [T, hidden]) -> [T, hidden]: ``gating -> dispatch -> experts -> combine`` built from the four
    operators above on this instance's backend (reference `MojoMoE`, `core/operators/moe.py:12-130`), with the reference's
    expert-parallel wiring: ``ep_size`` ranks each hold a contiguous slice of the experts, every rank routes all tokens,
    keeps the rows of its own experts and the partial outputs are summed over ``ep_group`` (all-reduce, or reduce-scatter
    when ``dp_input`` gathered the tokens first)."""

    def __init__(self, num_experts, top_k, hidden_size, intermediate_size=None, activation: str = "swiglu", ep_size: int = 1,
                 ep_rank: int = 0, ep_group=None, dp_input: bool = False, **kwargs):
        super().__init__()
        if activation != "swiglu":
            raise NotImplementedError(f"MojoMoe: Activation {activation} is not supported.")
        if intermediate_size is None:
            raise ValueError("MojoMoE: intermediate_size must be provided.")
        self.num_experts, self.top_k, self.hidden_size, self.intermediate_size = num_experts, top_k, hidden_size, intermediate_size
        self.ep_size, self.ep_rank, self.ep_group, self.dp_input = ep_size, ep_rank, ep_group, dp_input
        base, rem = divmod(num_experts, ep_size)
        self.num_experts_local = base + 1 if ep_rank < rem else base
        self.ep_start = base * ep_rank + min(ep_rank, rem)
        self.ep_end = self.ep_start + self.num_experts_local
        pick = lambda core: core.get_registry().get(self._backend)      # noqa: E731  sub-operators of the same backend
        self.gating = pick(MojoMoEGating)(hidden_size=hidden_size, num_experts=num_experts, top_k=top_k, **kwargs)
        self.dispatch = pick(MojoMoEDispatch)(num_experts=num_experts, **kwargs)
        self.experts = pick(MojoExperts)(num_experts=self.num_experts_local, hidden_size=hidden_size,
                                         intermediate_size=intermediate_size, activation=activation, **kwargs)
        self.combine = pick(MojoMoECombine)(multiply_by_gates=True, **kwargs)

    def compose_forward(self, hidden_states):
        """The orchestration shared by every backend (no arithmetic of its own)."""
        import torch.distributed as dist

        if self.dp_input and self.ep_size > 1:
            full = torch.empty(hidden_states.shape[0] * self.ep_size, *hidden_states.shape[1:], dtype=hidden_states.dtype,
                               device=hidden_states.device)
            dist.all_gather_into_tensor(full, hidden_states.contiguous(), group=self.ep_group)
            hidden_states = full
        top_k_indices, top_k_gates = self.gating(hidden_states)
        rows, per_expert, gates, token_indices = self.dispatch(hidden_states, top_k_gates, top_k_indices)
        if self.ep_size > 1:                      # keep the rows routed to this rank's experts (host sync, as the reference)
            ends = per_expert.cumsum(0)
            lo = 0 if self.ep_start == 0 else int(ends[self.ep_start - 1].item())
            hi = int(ends[self.ep_end - 1].item())
            rows, gates, token_indices = rows[lo:hi], gates[lo:hi], token_indices[lo:hi]
            per_expert = per_expert[self.ep_start:self.ep_end]
        expert_out = self.experts(rows, per_expert)
        out = self.combine(torch.zeros_like(hidden_states, memory_format=torch.contiguous_format), expert_out, gates, token_indices)
        if self.ep_size > 1:
            if self.dp_input:
                local = torch.empty(out.shape[0] // self.ep_size, *out.shape[1:], dtype=out.dtype, device=out.device)
                dist.reduce_scatter_tensor(local, out.contiguous(), op=dist.ReduceOp.SUM, group=self.ep_group)
                out = local
            else:
                dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.ep_group)
        return out


__all__ = ["MojoMoEGating", "MojoMoEDispatch", "MojoExperts", "MojoMoECombine", "MojoMoE", "count_expert_tokens"]
