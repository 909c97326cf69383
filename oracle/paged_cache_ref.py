"""CPU restatement of the reference's `PagedDummyCache` (mojo_opset/modeling/qwen3/mojo_qwen3_dense.py:41-135) — TEST
INFRASTRUCTURE: only tests/ may import it.  It is the checker for `mojo_opset_amd.PagedDummyCache`, whose allocator
runs on the device; block tables, lengths and cache contents must agree entry for entry / bit for bit."""
import torch

import mojo_opset_amd as mo

from . import torch_golden  # noqa: F401  (registers the Torch* classes)


class PagedDummyCacheRef:
    def __init__(self, num_layers, num_kv_heads, head_dim, max_position_embeddings, batch_size, block_size=16,
                 dtype=torch.bfloat16, total_blocks=None):
        self.num_layers, self.block_size, self.batch_size = num_layers, block_size, batch_size
        max_blocks_per_seq = (max_position_embeddings + block_size - 1) // block_size                      # :50
        total = batch_size * max_blocks_per_seq * num_layers if total_blocks is None else total_blocks    # :52
        self.k_cache = torch.zeros((total, num_kv_heads, block_size, head_dim), dtype=dtype)               # :55-64
        self.v_cache = torch.zeros_like(self.k_cache)
        self.block_tables = torch.full((num_layers, batch_size, max_blocks_per_seq), -1, dtype=torch.int32)  # :67-69
        self.seq_lens = torch.zeros((num_layers, batch_size), dtype=torch.int32)                           # :71
        self.free_blocks = torch.arange(total, dtype=torch.int32)                                          # :73
        self.num_free_blocks = total
        self.store = mo.MojoStorePagedKVCache.get_backend_impl("torch", strict=True)()

    def _allocate_blocks(self, n):                                                                         # :77-82
        if n > self.num_free_blocks:
            raise ValueError("PagedDummyCache: Out of memory!")
        got = self.free_blocks[self.num_free_blocks - n: self.num_free_blocks]
        self.num_free_blocks -= n
        return got

    def update(self, key_states, value_states, layer_idx):                                                 # :84-123
        batch, heads, new_len, dim = key_states.shape
        k = key_states.permute(0, 2, 1, 3).reshape(-1, heads, dim).contiguous()
        v = value_states.permute(0, 2, 1, 3).reshape(-1, heads, dim).contiguous()
        cu = torch.arange(0, (batch + 1) * new_len, step=new_len, dtype=torch.int32)
        cur = self.seq_lens[layer_idx]
        for i in range(batch):
            ctx = int(cur[i])
            old_nb = (ctx + self.block_size - 1) // self.block_size
            new_nb = (ctx + new_len + self.block_size - 1) // self.block_size
            if new_nb > old_nb:
                self.block_tables[layer_idx, i, old_nb:new_nb] = self._allocate_blocks(new_nb - old_nb)
        plan = mo.build_paged_kv_chunk_metadata(self.block_tables[layer_idx], cu, cur, self.block_size)
        self.store(k, v, self.k_cache, self.v_cache, chunk_metadata=plan)
        self.seq_lens[layer_idx] += new_len

    def get_kv_for_decode(self, layer_idx):                                                                # :128-132
        max_slen = int(self.seq_lens[layer_idx].max())
        max_blocks = (max_slen + self.block_size - 1) // self.block_size
        return self.k_cache, self.v_cache, self.block_tables[layer_idx, :, :max_blocks]
