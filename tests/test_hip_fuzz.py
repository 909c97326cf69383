"""Seeded random configurations of the paged attention operators against the oracle: shapes, page sizes, head groupings,
ragged / zero lengths, holes (-1) in the block tables, layouts and dtypes drawn at random.  The fixed cases elsewhere pin
the reference's own parameter sets; this sweep looks for a combination nobody wrote down."""
import math
import random

import pytest
import torch

from hip_utils import DEV, assert_close_tree, hip_cls, to_cpu, torch_cls
from test_hip_decode_gqa import make_decode_inputs
from test_hip_mla import build, check_mla, cu, exact_mla, make_mla, prefill_route
from test_hip_prefill_gqa import make_prefill_inputs

pytestmark = pytest.mark.gpu
# MOJO_FUZZ_OFFSET=<n> shifts every seed: a soak run walks offsets 1, 2, ... (the suite itself runs offset 0)
import os
OFFSET = 1000 * int(os.environ.get("MOJO_FUZZ_OFFSET", "0"))


@pytest.mark.parametrize("seed", range(24))
def test_fuzz_decode_gqa(seed):
    rnd = random.Random(1000 + seed + OFFSET)
    hkv = rnd.choice([1, 2, 4, 8])
    g = rnd.choice([1, 2, 4, 8])
    d = rnd.choice([64, 96, 128])
    page = rnd.choice([16, 32, 64, 128])
    batch = rnd.choice([1, 2, 3, 5, 8, 17, 64])
    max_len = rnd.choice([40, 300, 1100, 2500])
    lens = [rnd.choice([0, 1, rnd.randint(1, max_len), max_len]) for _ in range(batch)]
    dtype = rnd.choice([torch.bfloat16, torch.bfloat16, torch.float16])
    layout = rnd.choice(["ABAB", "AABB"])
    q, k, v, lens_t, table = make_decode_inputs(batch, hkv * g, hkv, d, max_len, page, dtype=dtype, seed=seed + OFFSET, lens=lens)
    if rnd.random() < 0.4 and table.shape[1] > 2:                       # a hole: the golden stops at the first negative id
        b = rnd.randrange(batch)
        table[b, rnd.randrange(1, table.shape[1])] = -1
    op = hip_cls("MojoPagedDecodeGQA")(is_causal=True, gqa_layout=layout)
    ref = torch_cls("MojoPagedDecodeGQA")(is_causal=True, gqa_layout=layout)
    scale = 1.0 / math.sqrt(d)
    want = ref(q, k, v, lens_t, table, softmax_scale=scale)
    got = op(*[t.to(DEV) for t in (q, k, v, lens_t, table)], softmax_scale=scale)
    assert_close_tree(to_cpu(got), want, 2e-2, 2e-2)
    hinted = op(*[t.to(DEV) for t in (q, k, v, lens_t, table)], softmax_scale=scale, max_total_seq_len=max(max(lens), 1))
    assert_close_tree(to_cpu(hinted), want, 2e-2, 2e-2)


@pytest.mark.parametrize("seed", range(24))
def test_fuzz_prefill_gqa(seed):
    rnd = random.Random(2000 + seed + OFFSET)
    hkv = rnd.choice([1, 2, 4, 8])
    g = rnd.choice([1, 2, 4, 8])
    d = rnd.choice([64, 96, 128])
    page = rnd.choice([16, 32, 64, 128])
    batch = rnd.choice([1, 2, 3, 4, 7])
    q_lens = [rnd.choice([0, 1, rnd.randint(1, 300), rnd.randint(100, 700)]) for _ in range(batch)]
    cached = [rnd.choice([0, 0, rnd.randint(1, 500)]) for _ in range(batch)]
    pad = rnd.choice([0, 0, 3, 40])
    dtype = rnd.choice([torch.bfloat16, torch.bfloat16, torch.float16])
    layout = rnd.choice(["ABAB", "AABB"])
    q, k, v, cu_q, table, cu_kv, kv_lens = make_prefill_inputs(q_lens, cached, hkv * g, hkv, d, page, dtype=dtype, seed=seed + OFFSET, pad_tokens=pad)
    op = hip_cls("MojoPagedPrefillGQA")(is_causal=True, gqa_layout=layout)
    ref = torch_cls("MojoPagedPrefillGQA")(is_causal=True, gqa_layout=layout)
    kw = {} if cu_kv is None else {"cu_total_seq_lens": cu_kv}
    want = ref(q, k, v, cu_q, table, **kw)
    dkw = {k_: v_.to(DEV) for k_, v_ in kw.items()}
    got = to_cpu(op(q.to(DEV), k.to(DEV), v.to(DEV), cu_q.to(DEV), table.to(DEV), **dkw))
    assert_close_tree(got, want, 2e-2, 2e-2)
    if pad:
        assert torch.count_nonzero(got[sum(q_lens):]) == 0
    if rnd.random() < 0.5:                                              # with the host hints: same bits unless the hints change the key split
        got2 = to_cpu(op(q.to(DEV), k.to(DEV), v.to(DEV), cu_q.to(DEV), table.to(DEV), max_q_len=max(q_lens + [0]),
                         max_total_seq_len=max(kv_lens + [0]), **dkw))
        from mojo_opset_amd.backends.hip import lib as L
        ws = lambda hq_, hk_: L.load().mojo_hip_paged_prefill_gqa_workspace_bytes(q.shape[0], batch, hkv * g, hkv, d, page, table.shape[1], hq_, hk_)  # noqa: E731
        if ws(0, 0) == 0 and ws(max(q_lens + [0]), max(kv_lens + [0])) == 0:
            assert torch.equal(got, got2)
        else:
            torch.testing.assert_close(got.float(), got2.float(), atol=8e-3, rtol=8e-3)


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_mla(seed):
    rnd = random.Random(3000 + seed + OFFSET)
    nope, rope, vd, r = rnd.choice([(128, 64, 128, 512), (64, 32, 64, 32), (96, 32, 128, 64)])
    h = rnd.choice([8, 16, 40, 128] if r == 512 else [8, 16])
    page = rnd.choice([16, 32, 64])
    sink = rnd.random() < 0.5
    batch = rnd.choice([1, 2, 3, 6])
    wscale = 0.05 if r == 512 else 0.2
    if rnd.random() < 0.5:                                              # decode
        lens = [rnd.choice([0, 1, rnd.randint(1, 700), rnd.randint(200, 1500)]) for _ in range(batch)]
        ckv, kpe, table, w, sk = make_mla(lens, h, nope, rope, vd, r, page, sink, seed=seed, wscale=wscale)
        g = torch.Generator().manual_seed(seed)
        q = torch.randn(batch, h, nope + rope, generator=g).to(torch.bfloat16)
        lens_t = torch.tensor(lens, dtype=torch.int32)
        ref = build("MojoPagedDecodeMLA", h, nope, rope, vd, r, sink, w, sk, "cpu")
        op = build("MojoPagedDecodeMLA", h, nope, rope, vd, r, sink, w, sk, DEV)
        got = to_cpu(op(q.to(DEV), ckv.to(DEV), kpe.to(DEV), lens_t.to(DEV), table.to(DEV)))
        check_mla(got, ref(q, ckv, kpe, lens_t, table), exact_mla(q, ckv, kpe, table, w, sk, h, nope, rope, vd, r, lens))
    else:                                                               # prefill
        kv_lens = [rnd.choice([0, rnd.randint(1, 200), rnd.randint(100, 600)]) for _ in range(batch)]
        q_lens = [min(n, rnd.choice([1, 7, 40, 130])) for n in kv_lens]
        ckv, kpe, table, w, sk = make_mla(kv_lens, h, nope, rope, vd, r, page, sink, seed=seed, wscale=wscale)
        g = torch.Generator().manual_seed(seed)
        q = torch.randn(sum(q_lens), h, nope + rope, generator=g).to(torch.bfloat16)
        ref = build("MojoPagedPrefillMLA", h, nope, rope, vd, r, sink, w, sk, "cpu", is_causal=True)
        op = build("MojoPagedPrefillMLA", h, nope, rope, vd, r, sink, w, sk, DEV, is_causal=True)
        want = ref(q, ckv, kpe, cu(q_lens), table, cu_total_seq_lens=cu(kv_lens))
        got = to_cpu(op(q.to(DEV), ckv.to(DEV), kpe.to(DEV), cu(q_lens).to(DEV), table.to(DEV), cu_total_seq_lens=cu(kv_lens).to(DEV)))
        exact = exact_mla(q, ckv, kpe, table, w, sk, h, nope, rope, vd, r, kv_lens, q_off=cu(q_lens).tolist())
        check_mla(got, want, exact, prefill_route(h, nope, rope, vd, q.shape[0]))


@pytest.mark.parametrize("seed", range(20))
def test_fuzz_group_gemm_exact(seed):
    """Small-integer operands (every partial sum exact in fp32): random group splits with empty groups, K and N at the
    kernels' alignment edges, both weight layouts, both 16-bit types — equality, element for element."""
    rnd = random.Random(4000 + seed + OFFSET)
    groups = rnd.choice([1, 2, 3, 8, 16])
    k = rnd.choice([32, 64, 96, 128, 192, 320, 448, 512, 1024])
    n = rnd.choice([8, 64, 72, 128, 264, 512, 1000])
    counts = [rnd.choice([0, 1, 15, 16, 17, 64, 100, 255, 256, 257, 300, 700]) for _ in range(groups)]
    if sum(counts) == 0:
        counts[0] = 33
    trans = rnd.random() < 0.5
    dtype = rnd.choice([torch.bfloat16, torch.float16])
    g = torch.Generator().manual_seed(seed)
    x = torch.randint(-3, 4, (sum(counts), k), generator=g).to(dtype)
    w = (torch.randint(-3, 4, (groups, n, k), generator=g) if trans else torch.randint(-3, 4, (groups, k, n), generator=g)).to(dtype)
    gl = torch.tensor(counts, dtype=torch.int32)
    want = torch_cls("MojoGroupGemm")(w.float(), trans)(x.float(), gl)
    got = hip_cls("MojoGroupGemm")(w.to(DEV), trans)(x.to(DEV), gl.to(DEV))
    assert torch.equal(to_cpu(got).float()[: sum(counts)], want.to(dtype).float())


@pytest.mark.parametrize("seed", range(24))
def test_fuzz_dense_gemm_exact(seed):
    """`mojo_hip_gemm` at every row count — the weight-streaming kernels, the 128-row tiles (both tile widths forced or the launcher's / the time model's own
    choice), the 256 x 256 kernel with and without its K split: small-integer operands, ragged M and N, K-tile counts from one
    up, both weight layouts and 16-bit types, bias with the golden's rounding (`F.linear`: one rounding; `x @ w + b`: two)."""
    import torch.nn.functional as F
    from hip_utils import last_launch, switch_env
    from mojo_opset_amd.backends.hip.operators.gemm import dense_gemm
    rnd = random.Random(4500 + seed + OFFSET)
    m = rnd.choice([1, 7, 33, 64, 65, 100, 128, 129, 130, 255, 256, 257, 300, 511, 640, 1000, 1024, 1100, 2047, 2500])   # (<= 128 with [K,N] weights: the 256 x 256 kernel's K split)
    k = 64 * rnd.choice([1, 2, 3, 4, 5, 8, 9, 16, 31, 64])
    n = 8 * rnd.choice([1, 2, 15, 16, 17, 32, 33, 64, 100, 128, 130, 512, 515])
    trans = rnd.random() < 0.5
    dtype = rnd.choice([torch.bfloat16, torch.float16])
    force = rnd.choice([None, None, "1", "128", "256", "0"])
    g = torch.Generator().manual_seed(seed + OFFSET)
    x = torch.randint(-3, 4, (m, k), generator=g).to(dtype).to(DEV)
    w = torch.randint(-3, 4, (n, k), generator=g).to(dtype).to(DEV)
    b = torch.randint(-8, 9, (n,), generator=g).to(dtype).to(DEV) if rnd.random() < 0.5 else None
    if trans:
        w = w.t().contiguous()
        want = (x.float() @ w.float()).to(dtype)
        want = want if b is None else (want.float() + b.float()).to(dtype)
    else:
        want = F.linear(x.float(), w.float(), None if b is None else b.float()).to(dtype)
    with switch_env(MOJO_HIP_GEMM_TILE128=force):
        got = dense_gemm(x, w, b, trans)
        form = last_launch()
    assert torch.equal(got, want), (form, m, k, n, trans, force)
    if force in ("1", "128", "256") and m > 64:
        assert form.startswith("gemm128:"), form
    if force == "0":
        assert not form.startswith("gemm128:"), form


@pytest.mark.parametrize("seed", range(20))
def test_fuzz_quant_gemm_int8_exact(seed):
    from test_hip_quant_gemm import _quantize, quant_gemm_formula
    rnd = random.Random(5000 + seed + OFFSET)
    m = rnd.choice([1, 2, 4, 5, 16, 31, 32, 33, 64, 100, 128, 129, 300, 513, 700, 1500, 2100])     # (above 128 rows with [N,K] weights: the 128-row tiles too)
    k = rnd.choice([48, 64, 128, 256, 272, 512, 1024, 1536, 4096])
    n = rnd.choice([3, 10, 64, 128, 192, 256, 1000, 4096])
    trans = rnd.random() < 0.5
    odt = rnd.choice([torch.bfloat16, torch.float16, torch.float32])
    torch.manual_seed(seed)
    xq, xs = _quantize(torch.randn(m, k))
    wq, ws = _quantize(torch.randn(n, k))
    op = hip_cls("MojoQuantGemm")(in_features=k, out_features=n, output_dtype=odt, trans_weight=trans, device=DEV)
    op.weight.copy_(wq if trans else wq.t())
    op.weight_scale.copy_(ws.to(torch.bfloat16))
    out = to_cpu(op(xq.to(DEV), xs.to(DEV)))
    torch.testing.assert_close(out, quant_gemm_formula(xq, wq.t(), xs, ws.to(torch.bfloat16), odt), atol=0, rtol=0)


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_store_paged_kv_bit_exact(seed):
    from conftest import bit_equal
    from test_hip_streaming import _store_case
    rnd = random.Random(6000 + seed + OFFSET)
    batch = rnd.choice([1, 2, 5, 9])
    page = rnd.choice([8, 16, 64, 128, 1024])
    seqs = [(rnd.choice([-1, 0, 0, rnd.randint(1, 3 * page)]), rnd.choice([0, 1, rnd.randint(1, 2 * page + 5)])) for _ in range(batch)]
    hkv, d = rnd.choice([1, 2, 8, 24]), rnd.choice([64, 96, 128, 256])
    dtype = rnd.choice([torch.bfloat16, torch.float16, torch.float32, torch.int8])
    ks, vs, kc, vc, table, cu_t, ctx = _store_case(seqs, hkv, d, page, dtype, seed=seed)
    ref = torch_cls("MojoStorePagedKVCache")()
    op = hip_cls("MojoStorePagedKVCache")()
    want = ref(ks, vs, kc.clone(), vc.clone(), table, cu_t, ctx)
    got = op(*[t.to(DEV) for t in (ks, vs, kc, vc, table, cu_t, ctx)])
    assert bit_equal(to_cpu(got), want)


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_norm_and_swiglu(seed):
    from hip_utils import max_ulp_bf16ish
    rnd = random.Random(7000 + seed + OFFSET)
    rows = rnd.choice([1, 2, 7, 57, 64, 300, 2048])
    d = rnd.choice([8, 64, 256, 1000, 1024, 4096, 7338, 8192])
    dtype = rnd.choice([torch.bfloat16, torch.float16, torch.float32])
    pos = rnd.choice(["pre", "post"])
    g = torch.Generator().manual_seed(seed)
    x, r, w = (torch.randn(rows, d, generator=g).to(dtype), torch.randn(rows, d, generator=g).to(dtype), torch.randn(d, generator=g).to(dtype))
    ref = torch_cls("MojoResidualAddRMSNorm")(d, 1e-5, pos, dtype=dtype)
    op = hip_cls("MojoResidualAddRMSNorm")(d, 1e-5, pos, dtype=dtype, device=DEV)
    with torch.no_grad():
        ref.weight.copy_(w)
        op.weight.copy_(w)
    atol, rtol = (5e-2, 1e-2) if dtype != torch.float32 else (2e-5, 2e-5)
    assert_close_tree(to_cpu(op(x.to(DEV), r.to(DEV))), ref(x, r), atol, rtol)
    limit = rnd.choice([0.0, 0.0, 1.5])
    gate, up = torch.randn(rows, d, generator=g).to(dtype), torch.randn(rows, d, generator=g).to(dtype)
    sref, sop = torch_cls("MojoSwiGLU")(swiglu_limit=limit), hip_cls("MojoSwiGLU")(swiglu_limit=limit)
    got, want = to_cpu(sop(gate.to(DEV), up.to(DEV))), sref(gate, up)
    if dtype == torch.float32:
        torch.testing.assert_close(got, want, atol=1e-5, rtol=1e-5)
    else:
        # the golden rounds silu(gate) to the storage type before the product; a silu that lands on the other side of a
        # rounding boundary (fp32 evaluation differing in the last bits) moves the product by one unit, two after its own
        # rounding — seen on fp16 with its 10-bit mantissa, never on bf16
        assert max_ulp_bf16ish(got, want) <= (1 if dtype == torch.bfloat16 else 2)


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_moe_dispatch_combine(seed):
    from test_hip_moe import _check_buckets
    rnd = random.Random(8000 + seed + OFFSET)
    experts = rnd.choice([2, 3, 8, 16, 60, 256])
    k = rnd.choice([1, 2, 4, 8])
    k = min(k, experts)
    hidden = rnd.choice([8, 100, 512, 1024, 4096])
    tokens = rnd.choice([1, 7, 64, 300, 2048])
    dtype = rnd.choice([torch.bfloat16, torch.float16])
    g = torch.Generator().manual_seed(seed + OFFSET)
    x = torch.rand(tokens, hidden, generator=g).to(dtype)
    probs = torch.softmax(torch.randn(tokens, experts, generator=g), dim=-1)
    gates, ids = torch.topk(probs, k, dim=-1)
    gates = (gates / gates.sum(-1, keepdim=True)).contiguous()
    ids = ids.to(torch.int32).contiguous()
    op = hip_cls("MojoMoEDispatch")(num_experts=experts)
    out = to_cpu(op(x.to(DEV), gates.to(DEV), ids.to(DEV)))
    want = torch_cls("MojoMoEDispatch")(num_experts=experts)(x, gates, ids)
    assert torch.equal(out[1], want[1])                                   # rows per expert
    _check_buckets(*out, x, gates, ids)
    # combine: bit-exact against the oracle on the oracle's own dispatch output
    by_gates = rnd.random() < 0.5
    buf = torch.zeros(tokens, hidden, dtype=dtype)
    rows, per_expert, sgates, tok = want
    cw = torch_cls("MojoMoECombine")(multiply_by_gates=by_gates)(buf.clone(), rows, sgates, tok)
    cg = hip_cls("MojoMoECombine")(multiply_by_gates=by_gates)(buf.to(DEV), rows.to(DEV), sgates.to(DEV), tok.to(DEV))
    assert torch.equal(to_cpu(cg), cw)


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_dynamic_quant_bit_exact(seed):
    from conftest import bit_equal
    rnd = random.Random(9000 + seed + OFFSET)
    rows = rnd.choice([1, 3, 64, 257, 4096])
    d = rnd.choice([8, 128, 1000, 4096, 7168])
    dtype = rnd.choice([torch.bfloat16, torch.float16])
    smooth = rnd.random() < 0.5
    g = torch.Generator().manual_seed(seed + OFFSET)
    x = (torch.randn(rows, d, generator=g) * rnd.choice([0.01, 1.0, 30.0])).to(dtype)
    op = hip_cls("MojoDynamicQuant")(input_size=d if smooth else None)
    ref = torch_cls("MojoDynamicQuant")(input_size=d if smooth else None)
    if smooth:
        w = torch.rand(d, generator=g) + 0.5
        with torch.no_grad():
            ref.inv_smooth_scale.copy_(w)
            op.inv_smooth_scale.copy_(w)
        op = op.to(DEV)
    assert bit_equal(to_cpu(op(x.to(DEV))), ref(x))
