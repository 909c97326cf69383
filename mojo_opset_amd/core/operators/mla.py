"""API classes of the paged MLA pair.

Follows `mojo_opset/experimental/operators/attention.py` (`MojoPagedDecodeMLA` :131-227,
`MojoPagedPrefillMLA` :325-447).  Faithful quirk: ``kv_b_proj`` is created by a bare
`torch.empty` that ignores the factory kwargs (:153-155), so it is fp32 on the default device
until the module is cast/moved; ``attn_sink`` is always fp32 (:14-17).
"""
import torch

from ..operator import MojoOperator


class _MLABase:
    def _init_mla(self, num_heads, qk_nope_head_dim, qk_rope_head_dim, v_head_dim, kv_lora_rank, use_attn_sink):
        self.num_heads = num_heads
        self.qk_nope_head_dim = qk_nope_head_dim
        self.qk_rope_head_dim = qk_rope_head_dim
        self.v_head_dim = v_head_dim
        self.kv_lora_rank = kv_lora_rank
        self.qk_head_dim = qk_nope_head_dim + qk_rope_head_dim
        self.use_attn_sink = use_attn_sink
        self.kv_b_proj = torch.nn.Parameter(torch.empty(num_heads * (qk_nope_head_dim + v_head_dim), kv_lora_rank))
        if use_attn_sink:
            sink_kwargs = dict(self.tensor_factory_kwargs)
            sink_kwargs["dtype"] = torch.float32
            self.attn_sink = torch.nn.Parameter(torch.empty(num_heads, **sink_kwargs))

    def _mla_repr(self) -> str:
        return (
            f"num_heads={self.num_heads}, qk_nope_head_dim={self.qk_nope_head_dim}, "
            f"qk_rope_head_dim={self.qk_rope_head_dim}, v_head_dim={self.v_head_dim}, "
            f"kv_lora_rank={self.kv_lora_rank}"
        )


class MojoPagedDecodeMLA(_MLABase, MojoOperator):
    """forward(query [B,H,nope+rope], compressed_kv_cache [N,1,page,r], k_pe_cache [N,1,page,rope],
    total_seq_lens [B] i32, block_tables [B,nb] i32, softmax_scale=None) -> [B,H,v]"""

    def __init__(self, num_heads, qk_nope_head_dim, qk_rope_head_dim, v_head_dim, kv_lora_rank,
                 use_attn_sink: bool = False, **kwargs):
        super().__init__(**kwargs)
        self._init_mla(num_heads, qk_nope_head_dim, qk_rope_head_dim, v_head_dim, kv_lora_rank, use_attn_sink)

    def extra_repr(self) -> str:
        return f"{self._mla_repr()}, use_attn_sink={self.use_attn_sink}"


class MojoPagedPrefillMLA(_MLABase, MojoOperator):
    """forward(query [T,H,nope+rope], compressed_kv_cache, k_pe_cache, cu_q_lens [B+1] i32,
    block_tables [B,nb] i32, softmax_scale=None, cu_total_seq_lens=None) -> [T,H,v]"""

    def __init__(self, num_heads, qk_nope_head_dim, qk_rope_head_dim, v_head_dim, kv_lora_rank,
                 is_causal: bool = True, use_attn_sink: bool = False, **kwargs):
        super().__init__(**kwargs)
        self._init_mla(num_heads, qk_nope_head_dim, qk_rope_head_dim, v_head_dim, kv_lora_rank, use_attn_sink)
        self.is_causal = is_causal

    def extra_repr(self) -> str:
        return f"{self._mla_repr()}, is_causal={self.is_causal}, use_attn_sink={self.use_attn_sink}"
