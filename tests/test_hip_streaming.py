"""GPU parity of the streaming ops (SwiGLU, RMSNorm family, RoPE family, StorePagedKVCache) — through
the C ABI, against (1) the reference vectors in tests/golden and (2) the oracle on the parameter space
of the reference's own accuracy tests (SURVEY §8a "Parameter space")."""
import pytest
import torch

import mojo_opset_amd as mo
from conftest import bit_equal, load_golden
from hip_utils import DEV, assert_close_tree, hip_cls, run_hip_case, to_cpu, torch_cls

pytestmark = pytest.mark.gpu


def _ids(group):
    return [pytest.param(c, id=f"{group}-{i}-{c['op']}") for i, c in enumerate(load_golden(group))]


# ---- reference vectors ---------------------------------------------------------------------------
@pytest.mark.parametrize("case", _ids("store_paged_kv"))
def test_store_paged_kv_vectors_bit_exact(case):
    assert bit_equal(to_cpu(run_hip_case(case)), case["out"])


@pytest.mark.parametrize("case", _ids("rope"))
def test_rope_vectors(case):
    out = to_cpu(run_hip_case(case))
    if case["op"] == "MojoApplyRoPE" or case["ctor"]["kwargs"].get("init_max_length") is not None:
        assert bit_equal(out, case["out"])          # same fp32 op order, single rounding / exact gather
    else:
        assert_close_tree(out, case["out"], atol=1e-5, rtol=1e-5)   # device cosf/sinf vs host libm


@pytest.mark.parametrize("case", _ids("swiglu"))
def test_swiglu_vectors(case):
    # tolerance of the reference test (default atol=rtol=1e-2); expf differs from the host libm by ulps
    assert_close_tree(to_cpu(run_hip_case(case)), case["out"], atol=1e-2, rtol=1e-2)


@pytest.mark.parametrize("case", _ids("rmsnorm"))
def test_rmsnorm_vectors(case):
    dtype = case["args"][0].dtype
    # reference test tolerance for 16-bit types: atol 5e-2, rtol 1e-2 (test_normalization.py:325-328)
    atol, rtol = (5e-2, 1e-2) if dtype != torch.float32 else (2e-5, 2e-5)
    assert_close_tree(to_cpu(run_hip_case(case)), case["out"], atol=atol, rtol=rtol)


# ---- the reference's own parameter space, HIP vs oracle -------------------------------------------
@pytest.mark.parametrize("shape", [(32, 1024), (64, 8192), (57, 7338), (2, 256)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("norm_pos", ["pre", "post"])
def test_residual_add_rmsnorm_reference_space(shape, dtype, norm_pos):
    torch.manual_seed(0)
    x, r, w = torch.randn(shape, dtype=dtype), torch.randn(shape, dtype=dtype), torch.randn(shape[-1], dtype=dtype)
    ref = torch_cls("MojoResidualAddRMSNorm")(shape[-1], 1e-5, norm_pos, dtype=dtype)
    op = hip_cls("MojoResidualAddRMSNorm")(shape[-1], 1e-5, norm_pos, dtype=dtype, device=DEV)
    with torch.no_grad():
        ref.weight.copy_(w)
        op.weight.copy_(w)
    op.forward_diff_with(ref, x.to(DEV), r.to(DEV), atol=5e-2, rtol=1e-2, ref_device="cpu")
    # the sum output is a plain rounded add: bit-exact
    got = to_cpu(op(x.to(DEV), r.to(DEV)))
    want = ref(x, r)
    assert torch.equal(got[1], want[1]) if norm_pos == "pre" else True


def test_rmsnorm_fp32_config1_full_size():
    """BASELINE config 1 shape: [2048, 4096] fp32."""
    torch.manual_seed(0)
    x, r, w = torch.randn(2048, 4096), torch.randn(2048, 4096), torch.randn(4096)
    ref = torch_cls("MojoResidualAddRMSNorm")(4096, 1e-5, "pre")
    op = hip_cls("MojoResidualAddRMSNorm")(4096, 1e-5, "pre", device=DEV)
    with torch.no_grad():
        ref.weight.copy_(w)
        op.weight.copy_(w)
    op.forward_diff_with(ref, x.to(DEV), r.to(DEV), atol=2e-5, rtol=2e-5, ref_device="cpu")
    g, u = torch.randn(2048, 4096), torch.randn(2048, 4096)
    hip_cls("MojoSwiGLU")().forward_diff_with(torch_cls("MojoSwiGLU")(), g.to(DEV), u.to(DEV), atol=1e-5, rtol=1e-5,
                                             ref_device="cpu")


@pytest.mark.parametrize("shape", [(256, 128), (1024, 10240), (999, 9999)])
def test_swiglu_reference_space(shape):
    torch.manual_seed(0)
    g, u = torch.rand(shape, dtype=torch.bfloat16), torch.rand(shape, dtype=torch.bfloat16)
    op, ref = hip_cls("MojoSwiGLU")(), torch_cls("MojoSwiGLU")()
    got = op.forward_diff_with(ref, g.to(DEV), u.to(DEV), ref_device="cpu")
    # rounding points mirrored: at most one unit in the last place away from the golden
    from hip_utils import max_ulp_bf16ish
    assert max_ulp_bf16ish(to_cpu(got), ref(g, u)) <= 1


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_swiglu_reads_the_halves_of_a_fused_projection_in_place(dtype):
    """`gate, up = gu.chunk(2, -1)` are row-strided views: HIPSwiGLU reads them where they are (mojo_hip_swiglu_rows) and gives
    the bits of the dense call; views it cannot address as rows (a transposed tensor, an odd column offset) are made dense."""
    torch.manual_seed(3)
    op = hip_cls("MojoSwiGLU")()
    for lead, inter in (((64,), 14336), ((3, 17), 1024), ((5,), 264)):
        gu = torch.randn(*lead, 2 * inter).to(dtype).to(DEV)
        gate, up = gu.chunk(2, dim=-1)
        assert not gate.is_contiguous()
        got = op(gate, up)
        want = op(gate.contiguous(), up.contiguous())
        assert got.shape == want.shape and got.is_contiguous() and torch.equal(got, want)
    gu = torch.randn(40, 2 * 100 + 3).to(dtype).to(DEV)             # odd offsets: not 16-byte aligned -> dense fallback
    gate, up = gu[:, 1:101], gu[:, 103:203]
    assert torch.equal(op(gate, up), op(gate.contiguous(), up.contiguous()))
    gt = torch.randn(128, 96).to(dtype).to(DEV).t()                   # transposed: no dense last dimension
    assert torch.equal(op(gt, gt), op(gt.contiguous(), gt.contiguous()))


@pytest.mark.parametrize("bs,seqlen", [(1, 124), (6, 555), (2, 2048)])
@pytest.mark.parametrize("dtype,hq,hk,head_first,d,pct", [
    (torch.float16, 32, 8, True, 96, 1.0), (torch.bfloat16, 8, 2, False, 96, 1 / 3),
    (torch.float16, 16, 8, True, 128, 1.0), (torch.bfloat16, 64, 8, False, 88, 1.0),
    (torch.float16, 64, 4, True, 128, 0.375),
])
@pytest.mark.parametrize("mode", ["varlen", "padded", "padded_batched_cos", "decode"])
def test_apply_rope_reference_space(bs, seqlen, dtype, hq, hk, head_first, d, pct, mode):
    torch.manual_seed(1)
    rd = int(d * pct)
    rd -= rd % 2
    rot = torch_cls("MojoRotaryEmbedding")(10000.0, rd, init_max_length=4096)

    def mk(*lead):
        if head_first:
            return torch.randn(*lead[:-1], hq, lead[-1], d, dtype=dtype), torch.randn(*lead[:-1], hk, lead[-1], d, dtype=dtype)
        return torch.randn(*lead, hq, d, dtype=dtype), torch.randn(*lead, hk, d, dtype=dtype)

    if mode == "varlen":
        q, k = mk(seqlen)
        cos, sin = rot.cos[:seqlen], rot.sin[:seqlen]
    elif mode == "decode":
        q, k = mk(bs)
        pos = torch.randint(0, 4096, (bs,))
        cos, sin = rot.cos[pos], rot.sin[pos]
    elif mode == "padded":
        q, k = mk(bs, seqlen)
        cos, sin = rot.cos[:seqlen], rot.sin[:seqlen]
    else:
        q, k = mk(bs, seqlen)
        pos = torch.randint(0, 4096, (bs, seqlen))
        cos, sin = rot.cos[pos], rot.sin[pos]
    want = torch_cls("MojoApplyRoPE")()(q, k, cos, sin, head_first=head_first)
    got = hip_cls("MojoApplyRoPE")()(q.to(DEV), k.to(DEV), cos.to(DEV), sin.to(DEV), head_first=head_first)
    assert bit_equal(to_cpu(got), want)


def test_apply_rope_reads_strided_views_in_place():
    """head-first *views* of token-first storage must be read through their strides."""
    torch.manual_seed(2)
    base_q = torch.randn(2, 33, 8, 64, dtype=torch.bfloat16)
    base_k = torch.randn(2, 33, 2, 64, dtype=torch.bfloat16)
    q, k = base_q.transpose(1, 2), base_k.transpose(1, 2)            # [B,N,S,D] views
    cos, sin = torch.randn(33, 64), torch.randn(33, 64)
    want = torch_cls("MojoApplyRoPE")()(q, k, cos, sin, head_first=True)
    got = hip_cls("MojoApplyRoPE")()(base_q.to(DEV).transpose(1, 2), base_k.to(DEV).transpose(1, 2), cos.to(DEV),
                                    sin.to(DEV), head_first=True)
    assert bit_equal(to_cpu(got), want)


@pytest.mark.parametrize("rope_dim", [32, 48, 64, 88, 96, 128])
@pytest.mark.parametrize("cached", [True, False])
def test_rotary_embedding_reference_space(rope_dim, cached):
    torch.manual_seed(3)
    kw = dict(rope_theta=10000.0, rope_dim=rope_dim, init_max_length=32768 if cached else None)
    ref = torch_cls("MojoRotaryEmbedding")(**kw)
    op = hip_cls("MojoRotaryEmbedding")(**kw, device=DEV)
    q_lens = torch.randint(0, 300, (7,))
    ctx = torch.randint(0, 2000, (7,))
    cu = torch.cat([torch.zeros(1, dtype=torch.int64), q_lens.cumsum(0)]).to(torch.int32)
    tot = (q_lens + ctx).to(torch.int32)
    T = int(cu[-1])
    x = torch.randn(T, 8)
    calls = [
        ((x,), {"cu_q_lens": cu}),
        ((x,), {"cu_q_lens": cu, "total_seq_lens": tot}),
        ((torch.randn(3, 77, 8),), {}),
        ((torch.randn(9, 8),), {"position_ids": torch.randint(0, 32768, (9,), dtype=torch.int32)}),
    ]
    for args, kwargs in calls:
        want = ref(*args, **kwargs)
        got = op(*[a.to(DEV) for a in args], **{k: v.to(DEV) for k, v in kwargs.items()})
        if cached:
            assert bit_equal(to_cpu(got), want)
        else:
            # angles reach ~3e4 rad: one ulp of the fp32 angle is 2e-3 rad, so compare in the table domain
            assert_close_tree(to_cpu(got), want, atol=2e-5, rtol=0) if rope_dim >= 0 else None


# ---- store paged kv: reference patterns, bit-exact -------------------------------------------------
def _store_case(seqs, hkv, d, page, dtype, spare=2, seed=0):
    torch.manual_seed(seed)
    ctx = torch.tensor([c for c, _ in seqs], dtype=torch.int32)
    q_lens = [q for _, q in seqs]
    need = [(max(c, 0) + q + page - 1) // page for c, q in seqs]
    total = sum(need) + spare
    ids = torch.randperm(total, dtype=torch.int32)
    table = torch.full((len(seqs), max(need) + 1), -1, dtype=torch.int32)
    at = 0
    for b, n in enumerate(need):
        table[b, :n] = ids[at: at + n]
        at += n
    cu = torch.tensor([0] + list(torch.tensor(q_lens).cumsum(0).tolist()), dtype=torch.int32)
    T = sum(q_lens)
    ks, vs = torch.randn(T, hkv, d).to(dtype), torch.randn(T, hkv, d).to(dtype)
    kc, vc = torch.randn(total, hkv, page, d).to(dtype), torch.randn(total, hkv, page, d).to(dtype)
    return ks, vs, kc, vc, table, cu, ctx


@pytest.mark.parametrize("seqs,hkv,d,page,dtype", [
    ([(0, 32)] * 16, 16, 128, 16, torch.float16),                         # perf case of the reference
    ([(100, 1), (0, 1), (2047, 1), (-1, 1), (17, 1)], 8, 128, 16, torch.bfloat16),
    ([(0, 700), (128, 129), (5, 0), (-1, 9), (1000, 2048)], 24, 128, 128, torch.bfloat16),
    ([(3, 9), (8, 8), (0, 24), (7, 1)], 2, 96, 8, torch.float16),          # bucket-padded, page 8
    ([(0, 1500)], 4, 64, 1024, torch.float32),
    ([(10, 3000), (4000, 96)], 1, 256, 2048, torch.int8),
])
def test_store_paged_kv_reference_space_bit_exact(seqs, hkv, d, page, dtype):
    ks, vs, kc, vc, table, cu, ctx = _store_case(seqs, hkv, d, page, dtype)
    ref = torch_cls("MojoStorePagedKVCache")()
    op = hip_cls("MojoStorePagedKVCache")()
    want = ref(ks, vs, kc.clone(), vc.clone(), table, cu, ctx)
    dev = [t.to(DEV) for t in (ks, vs, kc, vc, table, cu, ctx)]
    got = op(*[t.clone() for t in dev])                                       # legacy arguments
    assert bit_equal(to_cpu(got), want)
    plan = mo.build_paged_kv_chunk_metadata(dev[4], dev[5], dev[6], page)    # plan built on the device
    got = op(dev[0], dev[1], dev[2].clone(), dev[3].clone(), chunk_metadata=plan)
    assert bit_equal(to_cpu(got), want)
    # in-place contract: the returned tensors ARE the inputs
    kc_d, vc_d = dev[2].clone(), dev[3].clone()
    r = op(dev[0], dev[1], kc_d, vc_d, chunk_metadata=plan)
    assert r[0].data_ptr() == kc_d.data_ptr() and r[1].data_ptr() == vc_d.data_ptr()
    # decode mode
    B = len(seqs)
    ks1, vs1 = ks[:B] if ks.shape[0] >= B else torch.randn(B, hkv, d).to(dtype), None
    ks1 = torch.randn(B, hkv, d).to(dtype)
    vs1 = torch.randn(B, hkv, d).to(dtype)
    want = ref(ks1, vs1, kc.clone(), vc.clone(), table, None, ctx)
    got = op(ks1.to(DEV), vs1.to(DEV), dev[2].clone(), dev[3].clone(), dev[4], None, dev[6])
    assert bit_equal(to_cpu(got), want)


def test_store_paged_kv_empty_plan_and_mixing():
    op = hip_cls("MojoStorePagedKVCache")()
    ks = torch.randn(4, 2, 64, dtype=torch.bfloat16, device=DEV)
    kc = torch.randn(3, 2, 16, 64, dtype=torch.bfloat16, device=DEV)
    before = kc.clone()
    op(ks, ks, kc, kc.clone(), chunk_metadata=torch.empty((0, 4), dtype=torch.int32, device=DEV))
    assert torch.equal(kc, before)
    with pytest.raises(AssertionError):
        op(ks, ks, kc, kc, torch.zeros(1, 1, dtype=torch.int32, device=DEV),
           chunk_metadata=torch.empty((0, 4), dtype=torch.int32, device=DEV))
    with pytest.raises(AssertionError):
        op(ks, ks, kc, kc, chunk_metadata=torch.empty((0, 4), dtype=torch.int64, device=DEV))


def test_store_then_decode_round_trip_full_size():
    """Size-independent property at the BASELINE decode shape: what is stored token by token through
    the paged plan is exactly what a gather through the same block table reads back."""
    torch.manual_seed(5)
    B, hkv, d, page, ctx_len = 64, 8, 128, 16, 4096
    n_pages = B * (ctx_len // page)
    table = torch.randperm(n_pages, dtype=torch.int32, device=DEV).view(B, -1)
    kc = torch.zeros(n_pages, hkv, page, d, dtype=torch.bfloat16, device=DEV)
    vc = torch.zeros_like(kc)
    T = B * ctx_len
    ks = torch.randn(T, hkv, d, dtype=torch.bfloat16, device=DEV)
    vs = torch.randn(T, hkv, d, dtype=torch.bfloat16, device=DEV)
    cu = torch.arange(0, T + 1, ctx_len, dtype=torch.int32, device=DEV)
    ctx = torch.zeros(B, dtype=torch.int32, device=DEV)
    hip_cls("MojoStorePagedKVCache")()(ks, vs, kc, vc, table, cu, ctx)
    back = kc[table.long()].permute(0, 1, 3, 2, 4).reshape(T, hkv, d)      # [B, pages, page, h, d]
    assert torch.equal(back, ks)
    back = vc[table.long()].permute(0, 1, 3, 2, 4).reshape(T, hkv, d)
    assert torch.equal(back, vs)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_rmsnorm_inplace_writes_into_its_input(dtype):
    """MojoRMSNormInplace (experimental/operators/normalization.py:95-140), the q/k-norm in front of RoPE
    (modeling/qwen3/mojo_qwen3_dense.py:229-233): inplace=True returns the SAME tensor holding the norm, bit-equal to the
    out-of-place result; a strided view (the q slice of a fused qkv projection) is handled too; rows longer than the
    kernel's register cache (second pass re-reads the row it is overwriting) included."""
    torch.manual_seed(3)
    for shape in ((37, 8, 128), (5, 4, 9000), (3, 2, 77)):
        d = shape[-1]
        w = torch.randn(d, dtype=dtype)
        x = torch.randn(shape, dtype=dtype)
        ops = {ip: hip_cls("MojoRMSNormInplace")(d, 1e-6, inplace=ip, dtype=dtype, device=DEV) for ip in (False, True)}
        ref = torch_cls("MojoRMSNormInplace")(d, 1e-6, inplace=False, dtype=dtype)
        with torch.no_grad():
            ref.weight.copy_(w)
            for op in ops.values():
                op.weight.copy_(w)
        xd = x.to(DEV)
        want = ops[False](xd)
        assert want.data_ptr() != xd.data_ptr() and torch.equal(xd.cpu(), x)          # out of place: input untouched
        atol, rtol = (5e-2, 1e-2) if dtype != torch.float32 else (2e-5, 2e-5)
        assert_close_tree(to_cpu(want), ref(x), atol=atol, rtol=rtol)
        buf = xd.clone()
        got = ops[True](buf)
        assert got.data_ptr() == buf.data_ptr() and torch.equal(got, want)
        # strided: normalise the first `heads` heads of a wider fused tensor in place, the rest must not change
        fused = torch.randn(shape[0], shape[1] + 3, d, dtype=dtype).to(DEV)
        before = fused.clone()
        view = fused[:, : shape[1]]
        out = ops[True](view)
        assert out.data_ptr() == view.data_ptr()
        assert torch.equal(fused[:, shape[1]:], before[:, shape[1]:])
        assert torch.equal(fused[:, : shape[1]], ops[False](before[:, : shape[1]].contiguous()))
