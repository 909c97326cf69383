"""API classes `MojoRotaryEmbedding` / `MojoApplyRoPE`.

Follows `mojo_opset/core/operators/position_embedding.py:9-175`.  ``inv_freq`` and the optional
``cos``/``sin`` cache are non-persistent buffers with the reference's names, computed in fp32.
The tables are computed once on the host and then moved to the factory device, so the cached
gather path is bit-identical to the CPU golden regardless of the device's libm.
"""
from typing import Optional

import torch

from ..operator import MojoOperator


def rope_inv_freq(rope_theta, rope_dim: int) -> torch.Tensor:
    """``1 / theta ** (arange(0, d, 2) / d)`` in fp32 (reference :13-15)."""
    return 1.0 / (rope_theta ** (torch.arange(0, rope_dim, 2, dtype=torch.float32) / rope_dim))


def rope_table(inv_freq: torch.Tensor, max_length: int, attention_scaling: float):
    """fp32 ``cos, sin [max_length, rope_dim]`` = f(cat(freqs, freqs)) * scaling (reference :33-41)."""
    pos = torch.arange(max_length)
    freqs = pos[..., None] * inv_freq[None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos() * attention_scaling, emb.sin() * attention_scaling


class MojoRotaryEmbedding(MojoOperator):
    """forward(x, cu_q_lens=None, total_seq_lens=None, position_ids=None) -> (cos, sin) fp32.

    1. var-len prefill: ``x [T, H]`` + ``cu_q_lens [B+1]`` (+ ``total_seq_lens [B]``) -> ``[T, d]``
    2. padded prefill:  ``x [B, S, H]`` and nothing else                              -> ``[S, d]``
    3. decode / explicit: ``position_ids`` shaped like ``x.shape[:-1]``                -> ``[..., d]``
    """

    def __init__(self, rope_theta, rope_dim, attention_scaling: float = 1.0,
                 init_max_length: Optional[int] = None, **kwargs):
        super().__init__(**kwargs)
        device = self.tensor_factory_kwargs.get("device")
        self.rope_theta = rope_theta
        self.rope_dim = rope_dim
        self.attention_scaling = attention_scaling
        self.init_max_length = None
        self.register_buffer("inv_freq", rope_inv_freq(rope_theta, rope_dim).to(device), persistent=False)
        if init_max_length is not None:
            self._rope_init(init_max_length)

        def _ignore_rope_buffers(module, incompatible_keys) -> None:
            missing = incompatible_keys.missing_keys
            missing[:] = [k for k in missing if k.split(".")[-1] not in ("inv_freq", "cos", "sin")]

        self.register_load_state_dict_post_hook(_ignore_rope_buffers)

    def _rope_init(self, max_length: int) -> None:
        device = self.tensor_factory_kwargs.get("device")
        self.init_max_length = max_length
        cos, sin = rope_table(self.inv_freq.detach().cpu(), max_length, self.attention_scaling)
        self.register_buffer("cos", cos.to(device), persistent=False)
        self.register_buffer("sin", sin.to(device), persistent=False)

    @staticmethod
    def check_index_contract(x, cu_q_lens, total_seq_lens, position_ids) -> None:
        """int32 index tensors; at most one of ``cu_q_lens`` / ``position_ids`` (reference :59-65,:82)."""
        for t in (cu_q_lens, total_seq_lens, position_ids):
            if t is not None:
                assert t.dtype == torch.int32
        assert position_ids is None or cu_q_lens is None, "At most one of cu_q_lens or position_ids should be provided"
        if cu_q_lens is not None:
            assert x.dim() == 2, "x must be 2D: [T, D]"
        elif position_ids is not None:
            assert position_ids.shape == x.shape[:-1], (
                "position_ids must have the same shape as x except the hidden dimension"
            )


class MojoApplyRoPE(MojoOperator):
    """forward(q, k, cos, sin, head_first=True) -> (q_rot, k_rot), rotate-half on the last
    ``cos.shape[-1]`` features of every head; the leading features pass through."""

    def __init__(self, interleaved: bool = False):
        super().__init__()
        assert not interleaved, "interleaved impl is not supported yet."
        self.interleaved = interleaved

    @staticmethod
    def check_shape_contract(q, k, cos, sin) -> None:
        assert q.ndim == k.ndim, "q and k must have the same dimension"
        assert q.ndim == 3 or q.ndim == 4, "q and k must be 3D or 4D"
        assert cos.shape == sin.shape, "cos and sin must have the same shape"
        if q.ndim == 3:
            assert cos.ndim == 2, (
                "rotary position embedding (cos/sin) must be of shape [num_tokens, rope_dim] "
                "for varlen prefill or decode"
            )

    def extra_repr(self) -> str:
        return f"interleaved={self.interleaved!r}"
