"""API classes of the paged GQA attention pair.

Constructor arguments, contracts and `forward` signatures follow
`mojo_opset/core/operators/attention.py` (`MojoPagedDecodeGQA` :113-232,
`MojoPagedPrefillGQA` :315-451, contracts :12-37).  The classes are API-only; see
`core/operator.py` for why the golden `forward` is not here.
"""

import torch

from ..operator import MojoOperator

_GQA_LAYOUTS = ("ABAB", "AABB")


def assert_paged_decode_contract(block_tables, total_seq_lens) -> None:
    """int32 `[B]` lengths and int32 `[B, max_blocks]` table (reference :31-37)."""
    assert isinstance(block_tables, torch.Tensor) and isinstance(total_seq_lens, torch.Tensor)
    assert total_seq_lens.dtype == torch.int32
    assert block_tables.dtype == torch.int32
    assert block_tables.dim() == 2
    assert block_tables.shape[0] == total_seq_lens.shape[0]


def assert_paged_prefill_contract(cu_q_lens, block_tables, cu_total_seq_lens) -> None:
    """int32 `[B+1]` cumulative lengths and int32 `[B, max_blocks]` table (reference :12-28)."""
    assert isinstance(cu_q_lens, torch.Tensor) and isinstance(block_tables, torch.Tensor)
    assert cu_q_lens.dtype == torch.int32
    assert block_tables.dtype == torch.int32
    assert block_tables.dim() == 2
    batch = cu_q_lens.shape[0] - 1
    if cu_total_seq_lens is not None:
        assert isinstance(cu_total_seq_lens, torch.Tensor)
        assert cu_total_seq_lens.dtype == torch.int32
        assert cu_total_seq_lens.dim() == 1
        assert cu_total_seq_lens.shape[0] == batch + 1
    assert block_tables.shape[0] == batch


class _PagedGQABase:
    def _init_gqa(self, is_causal: bool, gqa_layout: str) -> None:
        if gqa_layout not in _GQA_LAYOUTS:
            raise ValueError(f"gqa_layout must be one of ['ABAB', 'AABB'], got {gqa_layout}")
        self.is_causal = is_causal
        self.gqa_layout = gqa_layout

    def extra_repr(self) -> str:
        return f"is_causal={self.is_causal!r}, gqa_layout={self.gqa_layout!r}"


class MojoPagedDecodeGQA(_PagedGQABase, MojoOperator):
    """One query token per sequence against a paged KV cache.

    forward(query [B,Hq,D], key_cache/value_cache [N_blocks,Hkv,page,D], total_seq_lens [B] i32,
            block_tables [B,max_blocks] i32 (unused = -1), softmax_scale=None, mask=None, *,
            max_total_seq_len=None) -> [B,Hq,D]; rows with seq_len <= 0 are zeros.
    """

    def __init__(self, is_causal: bool = True, gqa_layout: str = "AABB"):
        super().__init__()
        self._init_gqa(is_causal, gqa_layout)


class MojoPagedPrefillGQA(_PagedGQABase, MojoOperator):
    """Packed var-len queries against a paged KV cache, causal offset ``kv_len - q_len``.

    forward(query [T,Hq,D], key_cache, value_cache, cu_q_lens [B+1] i32, block_tables [B,nb] i32,
            softmax_scale=None, cu_total_seq_lens=None, mask=None, max_q_len=None,
            max_total_seq_len=None) -> [T,Hq,D]
    """

    def __init__(self, is_causal: bool = True, gqa_layout: str = "AABB"):
        super().__init__()
        self._init_gqa(is_causal, gqa_layout)
