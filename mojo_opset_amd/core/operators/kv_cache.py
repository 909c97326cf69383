"""API class `MojoStorePagedKVCache` and the int32 store-plan builder.

Follows `mojo_opset/core/operators/kv_cache.py` (contracts :9-30, builder :33-101, op :104-171).
A plan row is ``(src_token_start, dst_block_id, dst_block_offset, chunk_len)``.
"""

import torch

from ..operator import MojoOperator


def assert_paged_kv_store_contract(chunk_metadata: torch.Tensor) -> None:
    assert chunk_metadata.dtype == torch.int32
    assert chunk_metadata.dim() == 2
    assert chunk_metadata.shape[1] == 4


def assert_paged_kv_layout_contract(block_table, cu_q_lens, context_kv_lens) -> None:
    assert block_table.dtype == torch.int32
    assert block_table.dim() == 2
    if cu_q_lens is not None:
        assert cu_q_lens.dtype == torch.int32
        assert cu_q_lens.dim() == 1
    if context_kv_lens is not None:
        assert context_kv_lens.dtype == torch.int32
        assert context_kv_lens.dim() == 1
        assert block_table.shape[0] == context_kv_lens.shape[0]


def build_paged_kv_chunk_metadata(block_table, cu_q_lens, context_kv_lens, block_size: int) -> torch.Tensor:
    """Plan the copy of new tokens into pages: one int32 row per (sequence, touched page).

    Decode mode (``cu_q_lens is None``): one token per sequence, landing at logical page
    ``ctx // block_size``, slot ``ctx % block_size``; rows with ``ctx < 0``, a page index beyond
    the table, or a negative physical id are dropped (reference :56-74).
    Prefill mode: the interval ``[ctx, ctx + q_len)`` is intersected with every logical page;
    empty intersections, ``q_len == 0``, ``ctx < 0`` and negative ids are dropped (:78-101).
    Row order is sequence-major, page-minor, exactly as the reference's boolean mask produces.
    """
    assert_paged_kv_layout_contract(block_table, cu_q_lens, context_kv_lens)
    batch = context_kv_lens.shape[0]
    if cu_q_lens is not None:
        assert cu_q_lens.shape[0] == batch + 1
    dev = block_table.device
    max_pages = block_table.shape[1]
    if batch == 0 or max_pages == 0:
        return torch.empty((0, 4), dtype=torch.int32, device=dev)

    ctx = context_kv_lens.to(torch.int32)
    if cu_q_lens is None:
        seq = torch.arange(batch, dtype=torch.int32, device=dev)
        ctx0 = ctx.clamp_min(0)
        page = torch.div(ctx0, block_size, rounding_mode="floor")
        phys = block_table[seq.long(), page.clamp(0, max_pages - 1).long()]
        keep = (ctx >= 0) & (page < max_pages) & (phys >= 0)
        plan = torch.stack((seq, phys, torch.remainder(ctx0, block_size), torch.ones_like(seq)), dim=-1)
        return plan[keep]

    q_len = (cu_q_lens[1:] - cu_q_lens[:-1]).to(torch.int32)
    first_tok = cu_q_lens[:-1].to(torch.int32).unsqueeze(1)
    lo = ctx.unsqueeze(1)                                             # [B,1] first new position
    hi = (ctx + q_len).unsqueeze(1)                                   # [B,1] one past the last
    page_lo = torch.arange(max_pages, dtype=torch.int32, device=dev).unsqueeze(0) * block_size
    a = torch.maximum(lo, page_lo)
    b = torch.minimum(hi, page_lo + block_size)
    n = (b - a).clamp_min(0)
    keep = (q_len > 0).unsqueeze(1) & (ctx >= 0).unsqueeze(1) & (n > 0) & (block_table >= 0)
    plan = torch.stack((first_tok + (a - lo), block_table, a - page_lo, n), dim=-1)
    return plan[keep]


class MojoStorePagedKVCache(MojoOperator):
    """forward(key_states, value_states [T,Hkv,D], key_cache, value_cache [N,Hkv,page,D],
    block_table=None, cu_q_lens=None, context_kv_lens=None, *, chunk_metadata=None)
    -> (key_cache, value_cache), written **in place**.

    Either a prebuilt ``chunk_metadata`` plan or the legacy triple, never both (reference :139-154).
    """

    def __init__(self):
        super().__init__()

    @staticmethod
    def check_call_contract(key_states, value_states, block_table, cu_q_lens, context_kv_lens, chunk_metadata):
        assert key_states.dim() == 3 and value_states.dim() == 3 and key_states.shape == value_states.shape, (
            "key/value states must be (token_num, kv_head_num, head_dim), please check."
        )
        if chunk_metadata is None:
            assert block_table is not None, "block_table is required when chunk_metadata is not provided."
            assert context_kv_lens is not None, "context_kv_lens is required when chunk_metadata is not provided."
        else:
            assert block_table is None and cu_q_lens is None and context_kv_lens is None, (
                "chunk_metadata path should not be mixed with block_table/cu_q_lens/context_kv_lens."
            )
            assert_paged_kv_store_contract(chunk_metadata)


class MojoStorePagedMLAKVCache(MojoOperator):
    """forward(compressed_kv_states [T, r], k_pe_states [T, rope], compressed_kv_cache [N,1,page,r],
    k_pe_cache [N,1,page,rope], block_table [B, max_blocks], cu_q_lens [B+1] | None, context_kv_lens [B])
    -> (compressed_kv_cache, k_pe_cache), written **in place** (SURVEY §8 f3).

    Follows `mojo_opset/experimental/operators/kv_cache.py:13-106`: per sequence the new tokens land at positions
    ``context_kv_lens[b] ..``; ``cu_q_lens is None`` is decode mode (one token per sequence); a negative context length
    or a negative first table entry skips the sequence, and the first negative page id met ends its store.
    """

    def __init__(self):
        super().__init__()
