"""HIP-graph capturability of the hot-path ops: no op may synchronise with the host or allocate outside
torch's caching allocator, so a decode step can be captured once and replayed on static buffers — the contract of the
reference's `*_with_graph` tests (mojo_opset/tests/accuracy/operators/test_attention.py:218-353)."""
import pytest
import torch

from hip_utils import DEV, hip_cls

pytestmark = pytest.mark.gpu


def _capture(fn, warmup=2):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warmup):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = fn()
    return graph, out


def test_decode_attention_step_replays_on_static_buffers():
    torch.manual_seed(0)
    b, hq, hkv, d, page, max_len = 8, 32, 8, 128, 16, 512
    pages = max_len // page
    n_blocks = b * pages + 2
    k_cache = torch.zeros(n_blocks, hkv, page, d, dtype=torch.bfloat16, device=DEV)
    v_cache = torch.zeros_like(k_cache)
    table = torch.randperm(n_blocks, dtype=torch.int32)[: b * pages].view(b, pages).to(DEV)
    # static inputs of one decode step
    q = torch.zeros(b, hq, d, dtype=torch.bfloat16, device=DEV)
    k_new = torch.zeros(b, hkv, d, dtype=torch.bfloat16, device=DEV)
    v_new = torch.zeros_like(k_new)
    cos, sin = torch.zeros(b, d, device=DEV), torch.zeros(b, d, device=DEV)
    ctx = torch.zeros(b, dtype=torch.int32, device=DEV)          # tokens already in the cache
    total = torch.zeros(b, dtype=torch.int32, device=DEV)        # ctx + 1
    resid = torch.zeros(b, hq * d, dtype=torch.bfloat16, device=DEV)
    rope, store, attn = hip_cls("MojoApplyRoPE")(), hip_cls("MojoStorePagedKVCache")(), hip_cls("MojoPagedDecodeGQA")()
    norm = hip_cls("MojoResidualAddRMSNorm")(hq * d, 1e-5, "pre", dtype=torch.bfloat16, device=DEV)
    with torch.no_grad():
        norm.weight.copy_(torch.randn(hq * d))

    def step():
        q_r, k_r = rope(q.unsqueeze(0), k_new.unsqueeze(0), cos, sin, head_first=False)
        store(k_r.squeeze(0), v_new, k_cache, v_cache, table, None, ctx)
        o = attn(q_r.squeeze(0), k_cache, v_cache, total, table)
        return norm(o.reshape(b, hq * d), resid)

    def load(i):
        g = torch.Generator().manual_seed(100 + i)
        q.copy_(torch.randn(b, hq, d, generator=g))
        k_new.copy_(torch.randn(b, hkv, d, generator=g))
        v_new.copy_(torch.randn(b, hkv, d, generator=g))
        cos.copy_(torch.randn(b, d, generator=g))
        sin.copy_(torch.randn(b, d, generator=g))
        resid.copy_(torch.randn(b, hq * d, generator=g))
        lens = torch.randint(0, max_len - 1, (b,), generator=g).to(torch.int32)
        lens[i % b] = -1 if i % 2 else 0                 # a padded row (-1: not stored, attends over nothing) or an empty one
        ctx.copy_(lens)
        total.copy_(torch.clamp(lens + 1, min=0))

    load(0)
    graph, static_out = _capture(step)
    for i in range(1, 4):
        load(i)
        snap_k, snap_v = k_cache.clone(), v_cache.clone()
        graph.replay()
        torch.cuda.synchronize()
        got = [t.clone() for t in static_out]
        got_k, got_v = k_cache.clone(), v_cache.clone()
        k_cache.copy_(snap_k)                         # run the same step eagerly from the same cache state
        v_cache.copy_(snap_v)
        want = step()
        torch.cuda.synchronize()
        # Rows with total == 0 are padding: the captured attention leaves its output row as it was (the reference's
        # replay contract, tests/accuracy/operators/test_attention.py:340-353) while the eager call writes zeros, so
        # the layers behind it are compared on the live rows only.
        live = (total > 0).nonzero().flatten()
        assert live.numel() == (b - 1 if i % 2 else b)
        assert all(torch.equal(a[live], b_[live]) for a, b_ in zip(got, want))
        assert torch.equal(got_k, k_cache) and torch.equal(got_v, v_cache)


def test_moe_layer_and_quant_gemm_replay():
    torch.manual_seed(1)
    tokens, hidden, inter, experts, k = 64, 512, 1024, 8, 2
    x = torch.zeros(tokens, hidden, dtype=torch.bfloat16, device=DEV)
    gating = hip_cls("MojoMoEGating")(hidden_size=hidden, num_experts=experts, top_k=k).to(DEV)
    ffn = hip_cls("MojoExperts")(num_experts=experts, hidden_size=hidden, intermediate_size=inter).to(torch.bfloat16).to(DEV)
    with torch.no_grad():
        gating.gate_weight.normal_(std=0.05)
        ffn.up_proj_weight.normal_(std=0.05)
        ffn.down_proj_weight.normal_(std=0.05)
    dispatch, combine = hip_cls("MojoMoEDispatch")(num_experts=experts), hip_cls("MojoMoECombine")()
    quant = hip_cls("MojoDynamicQuant")()
    qgemm = hip_cls("MojoQuantGemm")(hidden, 256).to(DEV)
    with torch.no_grad():
        qgemm.weight.copy_(torch.randint(-127, 128, (hidden, 256), dtype=torch.int8))
        qgemm.weight_scale.copy_(torch.rand(256) * 0.01)

    def step():
        idx, gates = gating(x)
        rows, counts, sg, tok = dispatch(x, gates, idx)
        y = combine(x, ffn(rows, counts), sg, tok)
        y_q, s = quant(y)
        return y, qgemm(y_q, s.reshape(-1))

    x.copy_(torch.rand(tokens, hidden))
    graph, static_out = _capture(step)
    for i in range(3):
        x.copy_(torch.rand(tokens, hidden, generator=torch.Generator().manual_seed(7 + i)))
        graph.replay()
        torch.cuda.synchronize()
        got = [t.clone() for t in static_out]
        want = step()
        torch.cuda.synchronize()
        assert all(torch.equal(a, b_) for a, b_ in zip(got, want))


def test_mla_decode_step_replays():
    torch.manual_seed(2)
    b, h, nope, rope, vd, r, page, max_len = 4, 128, 128, 64, 128, 512, 16, 256
    pages = max_len // page
    n_blocks = b * pages + 1
    ckv = torch.zeros(n_blocks, 1, page, r, dtype=torch.bfloat16, device=DEV)
    kpe = torch.zeros(n_blocks, 1, page, rope, dtype=torch.bfloat16, device=DEV)
    table = torch.randperm(n_blocks, dtype=torch.int32)[: b * pages].view(b, pages).to(DEV)
    new_c = torch.zeros(b, r, dtype=torch.bfloat16, device=DEV)
    new_p = torch.zeros(b, rope, dtype=torch.bfloat16, device=DEV)
    q = torch.zeros(b, h, nope + rope, dtype=torch.bfloat16, device=DEV)
    ctx = torch.zeros(b, dtype=torch.int32, device=DEV)
    total = torch.zeros(b, dtype=torch.int32, device=DEV)
    store = hip_cls("MojoStorePagedMLAKVCache")()
    attn = hip_cls("MojoPagedDecodeMLA")(h, nope, rope, vd, r).to(torch.bfloat16).to(DEV)
    with torch.no_grad():
        attn.kv_b_proj.normal_(std=0.05)

    def step():
        store(new_c, new_p, ckv, kpe, table, None, ctx)
        return attn(q, ckv, kpe, total, table)

    def load(i):
        g = torch.Generator().manual_seed(50 + i)
        new_c.copy_(torch.randn(b, r, generator=g))
        new_p.copy_(torch.randn(b, rope, generator=g))
        q.copy_(torch.randn(b, h, nope + rope, generator=g))
        lens = torch.randint(0, max_len - 1, (b,), generator=g).to(torch.int32)
        ctx.copy_(lens)
        total.copy_(lens + 1)

    ckv.normal_()
    kpe.normal_()
    load(0)
    graph, static_out = _capture(step)
    for i in range(1, 4):
        load(i)
        snap = (ckv.clone(), kpe.clone())
        graph.replay()
        torch.cuda.synchronize()
        got = static_out.clone()
        ckv.copy_(snap[0])
        kpe.copy_(snap[1])
        want = step()
        torch.cuda.synchronize()
        assert torch.equal(got, want)
