"""Device-side paged KV cache with a block allocator — the role of the reference's `PagedDummyCache`
(`mojo_opset/modeling/qwen3/mojo_qwen3_dense.py:41-135`), SURVEY §8 (f4).

Same state and the same methods (`update`, `get_kv_for_prefill`, `get_kv_for_decode`, `get_seq_length`); what differs
is where the bookkeeping runs.  The reference reads every sequence's length with `.item()` (:99-109) and the longest
one again in `get_kv_for_decode` (:129), i.e. B + 1 host syncs per layer per step, which also rules out graph
capture.  Here `seq_lens`, `block_tables` and the free-list cursor live on the device and are updated by
`mojo_hip_page_pool_extend` / `mojo_hip_page_pool_advance`; the store is `MojoStorePagedKVCache` (hip backend) in its
legacy-argument mode, which evaluates the chunk plan per token on the device.  Nothing in `update` synchronises or
allocates outside torch's caching allocator, so `update` + paged decode attention can be captured in one HIP graph.

Given the same initial free list the block tables come out identical to the reference's (blocks are popped from the end
of the free list, sequence by sequence — see csrc/page_pool.hip).  Exhaustion cannot raise from a kernel: the launch then
changes nothing and latches an error word, and `check()` (one device->host copy; call it outside captured regions)
raises the reference's `ValueError("PagedDummyCache: Out of memory!")`.

Capture note: `max_total_seq_len_hint()` is host state and does not advance under graph replay — captured steps pass a
static upper bound to the paged ops (see the method's docstring).
"""
from typing import Optional

import torch

from .backends.hip import lib as L
from .core.operators.kv_cache import MojoStorePagedKVCache


class PagedDummyCache:
    def __init__(self, config, batch_size: int, device: str, block_size: int = 16, dtype: torch.dtype = torch.bfloat16,
                 total_blocks: Optional[int] = None):
        """``config`` supplies ``num_hidden_layers, num_key_value_heads, head_dim, max_position_embeddings`` (the four
        fields the reference reads, :43-50).  ``total_blocks`` defaults to the reference's worst case
        ``batch * ceil(max_position_embeddings / block_size) * layers`` (:52)."""
        self.num_layers = config.num_hidden_layers
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("PagedDummyCache (hip backend) needs a ROCm device; there is no CPU path")
        self.block_size = block_size
        self.num_kv_heads = config.num_key_value_heads
        self.head_dim = config.head_dim
        self.batch_size = batch_size
        self.max_blocks_per_seq = (config.max_position_embeddings + block_size - 1) // block_size
        self.total_blocks = (batch_size * self.max_blocks_per_seq * self.num_layers) if total_blocks is None else int(total_blocks)
        shape = (self.total_blocks, self.num_kv_heads, block_size, self.head_dim)
        self.k_cache = torch.zeros(shape, dtype=dtype, device=self.device)
        self.v_cache = torch.zeros(shape, dtype=dtype, device=self.device)
        self.block_tables = torch.full((self.num_layers, batch_size, self.max_blocks_per_seq), -1, dtype=torch.int32,
                                       device=self.device)
        self.seq_lens = torch.zeros((self.num_layers, batch_size), dtype=torch.int32, device=self.device)
        self.free_blocks = torch.arange(self.total_blocks, device=self.device, dtype=torch.int32)
        # {num_free, error, high-water of used blocks, 0} — the allocator's cursor lives on the device
        self.pool_state = torch.tensor([self.total_blocks, 0, 0, 0], dtype=torch.int32, device=self.device)
        self._store_ctx = torch.zeros(batch_size, dtype=torch.int32, device=self.device)   # written by the extend kernel
        self._cu_cache = {}
        self._len_bound = [0] * self.num_layers        # host-side UPPER bound of max(seq_lens[layer]): the kernels' grid hint
        self._pool_init = self.pool_state.clone()      # device-resident initial state: `reset` copies it without a host sync
        self.store_paged_kv = MojoStorePagedKVCache.get_backend_impl("hip", strict=True)()

    # ---- allocator state --------------------------------------------------------------------------------------
    @property
    def num_free_blocks(self) -> int:
        """Blocks left in the pool (reads the device cursor: one host sync; not for captured regions)."""
        return int(self.pool_state[0].item())

    def check(self) -> None:
        """Raise what the reference raises at the point of failure (:78-79) if an `update` ran out of blocks."""
        err = int(self.pool_state[1].item())
        if err == 1:
            raise ValueError("PagedDummyCache: Out of memory!")
        if err == 2:
            raise ValueError("PagedDummyCache: a sequence outgrew max_position_embeddings (block table too narrow)")

    def reset(self) -> None:
        """Return every block to the pool and forget all sequences (no sync: the initial cursor is a device tensor)."""
        self.block_tables.fill_(-1)
        self.seq_lens.zero_()
        self.pool_state.copy_(self._pool_init)
        self._len_bound = [0] * self.num_layers

    def _cu_q(self, new_len: int) -> torch.Tensor:
        cu = self._cu_cache.get(new_len)
        if cu is None:                                   # constant per step shape: built once, reused (and graph-safe)
            cu = torch.arange(0, (self.batch_size + 1) * new_len, new_len, device=self.device, dtype=torch.int32)
            self._cu_cache[new_len] = cu
        return cu

    # ---- the reference's interface ----------------------------------------------------------------------------
    def update(self, key_states: torch.Tensor, value_states: torch.Tensor, layer_idx: int,
               new_lens: Optional[torch.Tensor] = None) -> None:
        """``key_states / value_states [B, Hkv, S, D]``: append S tokens to every sequence of layer ``layer_idx``
        (:84-123).  ``new_lens`` (int32 ``[B]`` on the device, extension) appends only ``new_lens[i] <= S`` tokens of
        row i — padded rows of a bucketed batch pass 0 and keep their length, blocks and cache untouched."""
        batch, heads, new_len, dim = key_states.shape
        assert batch == self.batch_size and heads == self.num_kv_heads and dim == self.head_dim
        k = key_states.permute(0, 2, 1, 3).reshape(-1, heads, dim)
        v = value_states.permute(0, 2, 1, 3).reshape(-1, heads, dim)
        lens = self.seq_lens[layer_idx]
        table = self.block_tables[layer_idx]
        lib = L.load()
        stream = L.stream_of(self.k_cache)
        nl = None
        if new_lens is not None:
            assert new_lens.dtype == torch.int32 and new_lens.shape == (batch,) and new_lens.is_cuda
            if new_len != 1:                             # (checked BEFORE anything is enqueued: a raise behind the extend launch
                # would leave blocks popped and written into the table while the lengths never advance)
                raise NotImplementedError("PagedDummyCache.update: per-row new_lens is supported for decode steps (S = 1)")
            nl = new_lens.contiguous()
        L.check(lib.mojo_hip_page_pool_extend(L.ptr(table), table.stride(0), table.shape[1], L.ptr(lens), L.ptr(nl),
                                              new_len, L.ptr(self.free_blocks), L.ptr(self.pool_state),
                                              L.ptr(self._store_ctx), batch, self.block_size, self.total_blocks, stream),
                "PagedDummyCache.update")
        # the store evaluates the reference's chunk plan (kv_cache.py:33-101) per token on the device from
        # (table, cu_q_lens | None, context lengths); rows that append nothing carry context -1 = skipped
        if new_len == 1:
            self.store_paged_kv(k, v, self.k_cache, self.v_cache, table, None, self._store_ctx)
        else:
            self.store_paged_kv(k, v, self.k_cache, self.v_cache, table, self._cu_q(new_len), self._store_ctx)
        L.check(lib.mojo_hip_page_pool_advance(L.ptr(lens), L.ptr(nl), new_len, L.ptr(self.pool_state), batch, stream),
                "PagedDummyCache.update")
        self._len_bound[layer_idx] += new_len

    def get_kv_for_prefill(self, layer_idx: int):
        return None, None

    def get_kv_for_decode(self, layer_idx: int):
        """``(k_cache, v_cache, block_tables[layer])``.  The reference trims the table to the longest sequence with a
        `.item()` (:129-132); the hip kernels take the full-width table (-1 padded) plus a host-side bound
        (`max_total_seq_len_hint`) instead, so nothing is read back."""
        return self.k_cache, self.v_cache, self.block_tables[layer_idx]

    def max_total_seq_len_hint(self, layer_idx: int = 0) -> int:
        """Host-known upper bound of ``max(seq_lens[layer])`` — the `max_total_seq_len=` argument of the paged ops.

        It advances with every EAGER ``update`` call only: a captured step replays the kernels, not this Python counter, and
        the paged ops clamp lengths to the hint they were captured with.  A captured decode loop must therefore pass a
        static bound (``max_position_embeddings``, or the length the loop will reach) instead of this value."""
        return min(self._len_bound[layer_idx], self.max_blocks_per_seq * self.block_size)

    def get_seq_length(self, layer_idx: int = 0):
        return self.seq_lens[layer_idx].clone()
