"""GPU tests of the device-side paged KV cache / block allocator: `mojo_opset_amd.PagedDummyCache`
against (1) states captured from the reference's `PagedDummyCache` (tests/golden/paged_cache.pt,
modeling/qwen3/mojo_qwen3_dense.py:41-135) and (2) the CPU restatement on random step sequences; exhaustion; and a
captured decode step (append + paged attention) replayed on static buffers with no host sync."""
import math
from types import SimpleNamespace

import pytest
import torch

import mojo_opset_amd as mo
from conftest import load_golden
from hip_utils import DEV, hip_cls, torch_cls

pytestmark = pytest.mark.gpu


def _cfg(layers, heads, dim, max_pos):
    return SimpleNamespace(num_hidden_layers=layers, num_key_value_heads=heads, head_dim=dim, max_position_embeddings=max_pos)


def _same_state(cache, tables, lens, num_free):
    assert torch.equal(cache.block_tables.cpu(), tables)
    assert torch.equal(cache.seq_lens.cpu(), lens)
    assert cache.num_free_blocks == num_free


def test_paged_cache_reproduces_the_reference_trace():
    for case in load_golden("paged_cache"):
        if case["op"] != "PagedDummyCache":
            continue
        cache = mo.PagedDummyCache(SimpleNamespace(**case["config"]), case["batch"], DEV, block_size=case["block_size"])
        for step in case["trace"]:
            cache.update(step["k"].to(DEV), step["v"].to(DEV), step["layer"])
            _same_state(cache, step["block_tables"], step["seq_lens"], step["num_free"])
            # the reference trims the decode table to the longest row; ours is the same table, full width, -1 padded
            full = cache.get_kv_for_decode(step["layer"])[2].cpu()
            w = step["decode_table"].shape[1]
            assert torch.equal(full[:, :w], step["decode_table"]) and bool((full[:, w:] == -1).all())
            assert cache.max_total_seq_len_hint(step["layer"]) >= int(step["seq_lens"][step["layer"]].max())
        cache.check()
        assert torch.equal(cache.k_cache.cpu(), case["k_cache"]) and torch.equal(cache.v_cache.cpu(), case["v_cache"])


def test_paged_cache_random_steps_against_the_restatement():
    from oracle.paged_cache_ref import PagedDummyCacheRef

    g = torch.Generator().manual_seed(9)
    layers, heads, dim, max_pos, batch, page = 2, 8, 128, 700, 300, 16      # batch > 256: the allocator's scan crosses blocks of sequences
    ref = PagedDummyCacheRef(layers, heads, dim, max_pos, batch, block_size=page)
    cache = mo.PagedDummyCache(_cfg(layers, heads, dim, max_pos), batch, DEV, block_size=page)
    for new_len in (37, 1, 1, 16, 1, 250, 1, 1):
        for layer in range(layers):
            k = torch.randn(batch, heads, new_len, dim, generator=g).to(torch.bfloat16)
            v = torch.randn(batch, heads, new_len, dim, generator=g).to(torch.bfloat16)
            ref.update(k, v, layer)
            cache.update(k.to(DEV), v.to(DEV), layer)
        _same_state(cache, ref.block_tables, ref.seq_lens, ref.num_free_blocks)
    assert torch.equal(cache.k_cache.cpu(), ref.k_cache) and torch.equal(cache.v_cache.cpu(), ref.v_cache)
    assert torch.equal(cache.get_seq_length(1).cpu(), ref.seq_lens[1])


def test_paged_cache_exhaustion_changes_nothing_and_reports_it():
    want = [c for c in load_golden("paged_cache") if c["op"] == "PagedDummyCache.oom"][0]["message"]
    cache = mo.PagedDummyCache(_cfg(1, 1, 8, 8), 2, DEV, block_size=4)
    z = torch.zeros(2, 1, 8, 8, dtype=torch.bfloat16, device=DEV)
    cache.update(z, z, 0)
    cache.check()
    tables, lens = cache.block_tables.clone(), cache.seq_lens.clone()
    kc = cache.k_cache.clone()
    one = torch.ones(2, 1, 1, 8, dtype=torch.bfloat16, device=DEV)
    cache.update(one, one, 0)                                     # needs a third block per row: none left
    with pytest.raises(ValueError, match="Out of memory") as e:
        cache.check()
    assert str(e.value) == want
    assert torch.equal(cache.block_tables, tables) and torch.equal(cache.seq_lens, lens) and torch.equal(cache.k_cache, kc)
    cache.reset()
    cache.check()
    assert cache.num_free_blocks == cache.total_blocks and int(cache.seq_lens.sum()) == 0
    # blocks left but the table too narrow (a sequence past max_position_embeddings): also refused, also reported
    wide = mo.PagedDummyCache(_cfg(1, 1, 8, 8), 2, DEV, block_size=4, total_blocks=64)
    wide.update(z, z, 0)
    wide.update(one, one, 0)
    with pytest.raises(ValueError, match="max_position_embeddings"):
        wide.check()


def test_paged_cache_padded_rows_append_nothing():
    batch, heads, dim, page = 6, 2, 64, 16
    cache = mo.PagedDummyCache(_cfg(1, heads, dim, 256), batch, DEV, block_size=page)
    g = torch.Generator().manual_seed(2)
    k0 = torch.randn(batch, heads, 20, dim, generator=g).to(torch.bfloat16).to(DEV)
    cache.update(k0, k0, 0)
    before_t, before_k = cache.block_tables.clone(), cache.k_cache.clone()
    live = torch.tensor([1, 0, 1, 1, 0, 1], dtype=torch.int32, device=DEV)
    k1 = torch.randn(batch, heads, 1, dim, generator=g).to(torch.bfloat16).to(DEV)
    cache.update(k1, k1, 0, new_lens=live)
    assert cache.seq_lens[0].tolist() == [21, 20, 21, 21, 20, 21]
    assert torch.equal(cache.block_tables, before_t)               # 20 -> 21 stays inside the second block
    for b in range(batch):
        blk = int(cache.block_tables[0, b, 1])
        row = cache.k_cache[blk, :, 20 - page]
        assert torch.equal(row, k1[b, :, 0]) if int(live[b]) else torch.equal(row, before_k[blk, :, 20 - page])


def test_decode_step_with_the_cache_replays_in_a_graph():
    """One decode step of one layer — append the new K/V through the allocator, then paged decode attention over the
    pool — captured once and replayed: lengths, tables and the free-list cursor advance on the device; every replay
    must equal the oracle on the state the restatement reaches with the same inputs."""
    from oracle.paged_cache_ref import PagedDummyCacheRef

    batch, hq, hkv, d, page, max_pos = 4, 8, 2, 128, 16, 256
    g = torch.Generator().manual_seed(4)
    ref = PagedDummyCacheRef(1, hkv, d, max_pos, batch, block_size=page)
    cache = mo.PagedDummyCache(_cfg(1, hkv, d, max_pos), batch, DEV, block_size=page)
    k0 = torch.randn(batch, hkv, 30, d, generator=g).to(torch.bfloat16)
    v0 = torch.randn(batch, hkv, 30, d, generator=g).to(torch.bfloat16)
    ref.update(k0, v0, 0)
    cache.update(k0.to(DEV), v0.to(DEV), 0)
    attn = hip_cls("MojoPagedDecodeGQA")()
    gold = torch_cls("MojoPagedDecodeGQA")()
    sq = torch.zeros(batch, hq, d, dtype=torch.bfloat16, device=DEV)
    sk = torch.zeros(batch, hkv, 1, d, dtype=torch.bfloat16, device=DEV)
    sv = torch.zeros_like(sk)

    def step():
        cache.update(sk, sv, 0)
        kc, vc, table = cache.get_kv_for_decode(0)
        return attn(sq, kc, vc, cache.seq_lens[0], table, softmax_scale=1 / math.sqrt(d), max_total_seq_len=max_pos)

    def feed():
        q = torch.randn(batch, hq, d, generator=g).to(torch.bfloat16)
        k = torch.randn(batch, hkv, 1, d, generator=g).to(torch.bfloat16)
        v = torch.randn(batch, hkv, 1, d, generator=g).to(torch.bfloat16)
        sq.copy_(q), sk.copy_(k), sv.copy_(v)
        ref.update(k, v, 0)
        kc, vc, table = ref.get_kv_for_decode(0)
        return gold(q, kc, vc, ref.seq_lens[0], table, softmax_scale=1 / math.sqrt(d))

    want = feed()
    torch.testing.assert_close(step().cpu().float(), want.float(), atol=2e-2, rtol=2e-2)      # eager warm-up step
    want = feed()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = step()
    torch.cuda.synchronize()
    # capture does not execute: the cache has not advanced yet
    assert cache.seq_lens[0].tolist() == [31] * batch
    for i in range(5):                                            # crosses a block boundary at length 32 -> 33
        graph.replay()
        torch.cuda.synchronize()
        torch.testing.assert_close(out.cpu().float(), want.float(), atol=2e-2, rtol=2e-2)
        assert torch.equal(cache.block_tables.cpu(), ref.block_tables) and torch.equal(cache.seq_lens.cpu(), ref.seq_lens)
        assert torch.equal(cache.k_cache.cpu(), ref.k_cache)
        want = feed() if i < 4 else None
    cache.check()


def test_update_rejects_per_row_lengths_on_multi_token_steps_before_touching_the_pool():
    """the unsupported combination (per-row new_lens with S > 1) must raise BEFORE the extend kernel is
    enqueued — no block popped, no table entry written, lengths unchanged; `reset()` restores the cursor from a device
    tensor (no host copy)."""
    cache = mo.PagedDummyCache(_cfg(1, 2, 64, 128), 3, DEV, block_size=16)
    k = torch.randn(3, 2, 4, 64, dtype=torch.bfloat16, device=DEV)
    before = (cache.pool_state.clone(), cache.block_tables.clone(), cache.seq_lens.clone())
    with pytest.raises(NotImplementedError):
        cache.update(k, k, 0, new_lens=torch.tensor([4, 0, 2], dtype=torch.int32, device=DEV))
    torch.cuda.synchronize()
    assert torch.equal(cache.pool_state, before[0]) and torch.equal(cache.block_tables, before[1]) and torch.equal(cache.seq_lens, before[2])
    cache.update(k, k, 0)
    assert cache.num_free_blocks == cache.total_blocks - 3
    cache.reset()
    assert cache.num_free_blocks == cache.total_blocks and int(cache.seq_lens.sum()) == 0
