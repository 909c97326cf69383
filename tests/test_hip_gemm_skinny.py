"""GPU parity of the decode-sized grouped / dense GEMM path (`csrc/gemm_skinny.hip`): equal-sized groups of <= 128 rows,
K-major weights, optional row maps — reached through `mojo_hip_group_gemm_strided` / `mojo_hip_gemm`."""
import pytest
import torch

from hip_utils import DEV, last_launch, max_ulp_bf16ish, to_cpu
from mojo_opset_amd.backends.hip import lib as L
from mojo_opset_amd.backends.hip.operators.gemm import dense_gemm

pytestmark = pytest.mark.gpu


def _grouped(x, w, groups, rows, k, n, lda, a_map=None, c_map=None, out=None, ldc=None):
    lib = L.load()
    out = torch.empty(groups * rows, n, dtype=x.dtype, device=x.device) if out is None else out
    ws = torch.empty(lib.mojo_hip_group_gemm_workspace_bytes(groups), dtype=torch.uint8, device=x.device)
    L.check(lib.mojo_hip_group_gemm_strided(L.ptr(x), L.ptr(w), L.ptr(out), None, 0, groups * rows, k, n, groups, lda,
                                            n if ldc is None else ldc, n * k, 1, k,
                                            None if a_map is None else L.ints4(*a_map), None if c_map is None else L.ints4(*c_map),
                                            L.dtype_code(x.dtype), L.ptr(ws), ws.numel(), L.stream_of(x)), "test gemm")
    return out


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("groups,rows,k,n", [(1, 1, 128, 64), (3, 16, 256, 128), (128, 64, 512, 128), (128, 64, 128, 512),
                                             (2, 17, 1024, 192), (5, 128, 2048, 64), (1, 100, 4096, 256), (4, 33, 896, 320)])
def test_skinny_grouped_gemm_integer_data_is_exact(dtype, groups, rows, k, n):
    g = torch.Generator().manual_seed(groups * 1000 + rows)
    x = torch.randint(-3, 4, (groups * rows, k), generator=g).to(dtype)
    w = torch.randint(-3, 4, (groups, n, k), generator=g).to(dtype)                     # [G, N, K]
    got = to_cpu(_grouped(x.to(DEV), w.to(DEV), groups, rows, k, n, k))
    want = torch.einsum("grk,gnk->grn", x.view(groups, rows, k).float(), w.float()).reshape(groups * rows, n).to(dtype)
    assert torch.equal(got.float(), want.float())


def test_skinny_grouped_gemm_row_maps_token_major():
    """The MLA projections' view: logical row h*T + t lives at storage row t*H + h on both sides."""
    torch.manual_seed(0)
    heads, tokens, k, n = 16, 24, 256, 128
    x = torch.randn(tokens, heads, k, dtype=torch.bfloat16)
    w = torch.randn(heads, n, k, dtype=torch.bfloat16) * 0.1
    out = torch.zeros(tokens, heads, n, dtype=torch.bfloat16, device=DEV)
    m = (tokens, 1, 0, heads)
    _grouped(x.to(DEV), w.to(DEV), heads, tokens, k, n, k, a_map=m, c_map=m, out=out)
    want = torch.einsum("thk,hnk->thn", x.float(), w.float())
    assert max_ulp_bf16ish(to_cpu(out), want.to(torch.bfloat16), atol=1e-2) <= 1


@pytest.mark.parametrize("m,k,n,bias", [(1, 4096, 4096, False), (64, 8192, 1024, True), (128, 1024, 28672, False), (37, 384, 64, True),
                                        # long K, few column tiles: split-K slabs + finalize
                                        (64, 14336, 4096, False), (5, 28672, 64, True), (128, 8192, 512, True)])
def test_skinny_dense_gemm_matches_fp32_reference(m, k, n, bias):
    torch.manual_seed(1)
    x = torch.randn(m, k, dtype=torch.bfloat16)
    w = torch.randn(n, k, dtype=torch.bfloat16) * 0.05                                   # F.linear layout = [N, K]
    b = torch.randn(n, dtype=torch.bfloat16) if bias else None
    got = to_cpu(dense_gemm(x.to(DEV), w.to(DEV), None if b is None else b.to(DEV), False))
    # [N, K] weights = the golden's F.linear(input, weight, bias): the bias joins the fp32 accumulator, ONE rounding
    # (core/operators/compute_with_comm.py:12-24, gemm.py:45-46; torch's CPU addmm rounds once — round-5 probe)
    want = torch.nn.functional.linear(x, w, b)
    assert max_ulp_bf16ish(got, want, atol=2e-2) <= 1


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("counts,k,n", [
    ([2, 0, 5, 1, 0, 0, 3, 7], 256, 128),                       # MoE decode: a few rows per expert, empty experts
    ([1] * 64, 1024, 192), ([0] * 31 + [9], 128, 64), ([40, 0, 33, 1], 512, 64),      # 40 and 33 rows: two 32-row chunks
    ([3, 100, 0, 2, 1, 1, 0, 0], 384, 256),                     # one hot expert inside an otherwise sparse routing
])
def test_ragged_decode_groups_integer_data_is_exact(dtype, counts, k, n):
    """MoE-decode-shaped `HIPGroupGemm` calls: a few rows per expert, empty experts, one hot expert."""
    from hip_utils import hip_cls, torch_cls
    g = torch.Generator().manual_seed(len(counts) * 7 + k)
    groups = len(counts)
    x = torch.randint(-3, 4, (sum(counts), k), generator=g).to(dtype)
    w = torch.randint(-3, 4, (groups, n, k), generator=g).to(dtype)
    gl = torch.tensor(counts, dtype=torch.int32)
    want = torch_cls("MojoGroupGemm")(w.float(), True)(x.float(), gl).to(dtype)
    got = hip_cls("MojoGroupGemm")(w.to(DEV), True)(x.to(DEV), gl.to(DEV))
    assert torch.equal(to_cpu(got).float(), want.float())


@pytest.mark.parametrize("m,k,n,bias", [(129, 4096, 4096, False), (256, 4096, 4096, True), (300, 1024, 520, True), (1000, 8192, 4096, False),
                                        (513, 512, 260, False), (200, 4160, 4100, True), (1024, 8192, 8192, False)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_dense_gemm_split_over_k_for_few_output_tiles_is_exact(m, k, n, bias, dtype, monkeypatch):
    """More than 128 rows but few 256x256 output tiles: K is cut into slices that go to fp32 slabs and a second launch sums
    them in slice order (gemm_api.hip, gemm_dense_splitk256).  Small-integer data makes every product and partial sum exact,
    so the split, the unsplit (MOJO_HIP_GEMM_SPLITK=1) and the fp32 reference must agree to the bit — ragged edges in M
    and N, the bias joining the accumulator before the one rounding (F.linear semantics of the [N, K] layout)."""
    import torch.nn.functional as F
    from mojo_opset_amd.backends.hip.operators.gemm import dense_gemm
    torch.manual_seed(m + n)
    x = torch.randint(-4, 5, (m, k)).to(dtype).to(DEV)
    w = torch.randint(-4, 5, (n, k)).to(dtype).to(DEV)
    b = torch.randint(-8, 9, (n,)).to(dtype).to(DEV) if bias else None
    want = F.linear(x.float(), w.float(), None if b is None else b.float()).to(dtype)     # F.linear: ONE rounding, bias included
    got = dense_gemm(x, w, b, False)
    assert torch.equal(got, want)
    monkeypatch.setenv("MOJO_HIP_GEMM_SPLITK", "1")
    assert torch.equal(dense_gemm(x, w, b, False), want)
    unsplit = last_launch()
    monkeypatch.setenv("MOJO_HIP_GEMM_SPLITK", "3")
    assert torch.equal(dense_gemm(x, w, b, False), want)
    if k >= 3 * 8 * 64:                                            # (a slice keeps at least 8 K-tiles: shorter K clamps the forced split)
        assert ":splitk" in last_launch() and ":splitk" not in unsplit, (unsplit, last_launch())
    monkeypatch.delenv("MOJO_HIP_GEMM_SPLITK")
    # random data: the split only reorders fp32 partial sums
    xr, wr = torch.randn(m, k, device=DEV, dtype=dtype), torch.randn(n, k, device=DEV, dtype=dtype)
    ref = F.linear(xr.float(), wr.float())
    out = dense_gemm(xr, wr, None, False).float()
    assert (out - ref).abs().max() <= 1e-2 * ref.abs().max()
    assert torch.equal(out, dense_gemm(xr, wr, None, False).float())          # same bits on a second launch


def test_split_k_combined_inside_the_launch_gives_the_finalize_kernels_bits(monkeypatch):
    """csrc/experiments/splitk_combine.h (experiments build only): the last K slice to arrive sums all slices in index order.  Same bits as the two-launch form
    (the default; MOJO_HIP_SPLITK_INLAUNCH=1 selects the combine), the same bits launch after launch while OTHER products keep the chip unevenly busy (the
    hand-off must not depend on placement or timing), across more calls than there are ticket slots (64), for bf16, int8
    and fp8."""
    from hip_utils import hip_cls, skip_unless_experiments_build
    skip_unless_experiments_build()
    torch.manual_seed(3)
    cases = []
    for m, k, n in ((64, 8192, 1024), (17, 14336, 512), (128, 8192, 512), (1, 28672, 64)):
        x = torch.randn(m, k, dtype=torch.bfloat16).to(DEV)
        w = (torch.randn(n, k, dtype=torch.bfloat16) * 0.05).to(DEV)
        cases.append(lambda x=x, w=w: dense_gemm(x, w, None, False))
    for qd in (torch.int8, torch.float8_e4m3fn):
        for m, k, n in ((128, 7168, 4096), (32, 7168, 4096), (48, 18432, 1024)):
            op = hip_cls("MojoQuantGemm")(k, n, trans_weight=True, quant_dtype=qd, weight_dtype=qd, device=DEV)
            if qd == torch.int8:
                op.weight.copy_(torch.randint(-127, 128, (n, k), dtype=torch.int8).to(DEV))
                xq = torch.randint(-127, 128, (m, k), dtype=torch.int8).to(DEV)
            else:
                op.weight.copy_(torch.randn(n, k).to(DEV).to(qd))
                xq = torch.randn(m, k).to(DEV).to(qd)
            op.weight_scale.fill_(0.01)
            sc = torch.rand(m).to(DEV)
            cases.append(lambda op=op, xq=xq, sc=sc: op(xq, sc))
    monkeypatch.setenv("MOJO_HIP_SPLITK_INLAUNCH", "0")
    two_launch = [c().clone() for c in cases]
    monkeypatch.setenv("MOJO_HIP_SPLITK_INLAUNCH", "1")
    noise_x = torch.randn(4096, 4096, dtype=torch.bfloat16, device=DEV)
    side = torch.cuda.Stream()
    for rnd in range(12):                                              # 12 x 10 calls: every ticket slot is used twice
        with torch.cuda.stream(side):                                  # uneven load next to the products
            for _ in range(1 + rnd % 3):
                noise_x @ noise_x
        for c, want in zip(cases, two_launch):
            assert torch.equal(c(), want)
    torch.cuda.synchronize()


# ---- decode-sized fusions: GEMM + SwiGLU in one launch, split-K slabs summed by the residual RMSNorm ----------------------

def _swiglu_chain(x, w_gu):
    from hip_utils import hip_cls
    gu = dense_gemm(x, w_gu, None, False)
    inter = w_gu.shape[0] // 2
    return hip_cls("MojoSwiGLU")()(gu[:, :inter], gu[:, inter:])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,k,inter", [(64, 4096, 14336), (1, 128, 8), (17, 256, 56), (33, 384, 1024), (64, 1024, 2304),
                                       (5, 512, 40), (48, 128, 4104),
                                       (100, 256, 128), (64, 192, 64)])      # > 64 rows / K % 128: the two-launch route
def test_gemm_swiglu_fused_equals_the_separate_calls_and_the_oracle(dtype, m, k, inter):
    """`mojo_hip_gemm_swiglu`: same bits as linear -> MojoSwiGLU wherever the separate linear runs unsplit (every workgroup size the balancing rule picks: 56 columns
    -> 7 units, 1024 -> 128, 2304 -> 288, 40 -> 5, 4104 -> 513 units), and the oracle's chain within its tolerance."""
    from hip_utils import torch_cls
    from mojo_opset_amd.backends.hip.operators.gemm import dense_gemm_swiglu
    torch.manual_seed(m * 31 + inter)
    x = torch.randn(m, k, dtype=dtype)
    w = (torch.randn(2 * inter, k) * (2.0 / k ** 0.5)).to(dtype)
    got = dense_gemm_swiglu(x.to(DEV), w.to(DEV))
    chain = _swiglu_chain(x.to(DEV), w.to(DEV))
    assert got.shape == (m, inter)
    if L.load().mojo_hip_gemm_workspace_bytes(m, k, 2 * inter) <= 64:       # the separate projection does not cut K: same bits
        assert torch.equal(got, chain)
    else:                                                                    # it sums two or more K slices: another fp32 order
        assert max_ulp_bf16ish(to_cpu(got), to_cpu(chain), atol=2e-3) <= 2
    gu = (x.float() @ w.float().t()).to(dtype)
    want = torch_cls("MojoSwiGLU")()(gu[:, :inter], gu[:, inter:])
    torch.testing.assert_close(to_cpu(got).float(), want.float(), atol=3e-2, rtol=3e-2)


def test_gemm_swiglu_every_workgroup_size_gives_the_same_bits(monkeypatch):
    from mojo_opset_amd.backends.hip.operators.gemm import dense_gemm_swiglu
    torch.manual_seed(5)
    m, k, inter = 64, 512, 1016                                              # 127 wave units: every size leaves a partial workgroup
    x = torch.randn(m, k, dtype=torch.bfloat16, device=DEV)
    w = (torch.randn(2 * inter, k, device=DEV) * 0.1).to(torch.bfloat16)
    ref = _swiglu_chain(x, w)
    for nw in (4, 5, 6, 7, 8):
        monkeypatch.setenv("MOJO_HIP_GEMM_WAVES", str(nw))
        assert torch.equal(dense_gemm_swiglu(x, w), ref), nw
        assert last_launch() == f"gemm_skinny:glu:waves{nw}", last_launch()
    monkeypatch.delenv("MOJO_HIP_GEMM_WAVES")
    monkeypatch.setenv("MOJO_HIP_GEMM_SKINNY", str(31 & ~4))               # without the fused SwiGLU epilogue: GEMM, then the activation
    assert torch.equal(dense_gemm_swiglu(x, w), ref)
    assert not last_launch().startswith("gemm_skinny:glu"), last_launch()


def test_gemm_swiglu_integer_data_is_exact():
    from mojo_opset_amd.backends.hip.operators.gemm import dense_gemm_swiglu
    g = torch.Generator().manual_seed(9)
    m, k, inter = 37, 256, 72
    x = torch.randint(-2, 3, (m, k), generator=g).to(torch.bfloat16)
    w = torch.randint(-2, 3, (2 * inter, k), generator=g).to(torch.bfloat16)
    gu = (x.float() @ w.float().t()).to(torch.bfloat16)                       # exact
    want = (torch.nn.functional.silu(gu[:, :inter]) * gu[:, inter:])
    got = to_cpu(dense_gemm_swiglu(x.to(DEV), w.to(DEV)))
    assert max_ulp_bf16ish(got, want, atol=1e-3) <= 1                        # (device exp2 / rcp vs torch's silu: one step)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,k,n,bias,resid", [(64, 4096, 4096, False, True), (64, 14336, 4096, False, True), (1, 2048, 512, True, True),
                                              (33, 8192, 8192, True, True), (64, 4096, 2048, False, False), (16, 1024, 16384, False, True),
                                              (7, 256, 192, True, True),          # no K split: the two-launch route
                                              (200, 512, 1024, False, True),      # 128-row tiles, unsplit, then the norm
                                              (100, 4096, 4096, True, True),      # 128-row tiles with their own K split: slabs into the norm
                                              (256, 8192, 1024, False, True), (512, 4096, 4096, True, False)])
def test_gemm_residual_rmsnorm_fused_equals_the_separate_calls_and_the_oracle(dtype, m, k, n, bias, resid):
    from hip_utils import hip_cls, torch_cls
    from mojo_opset_amd.backends.hip.operators.gemm import dense_gemm_residual_rmsnorm
    torch.manual_seed(m + n)
    x = torch.randn(m, k, dtype=dtype)
    w = (torch.randn(n, k) / k ** 0.5).to(dtype)
    b = torch.randn(n, dtype=dtype) if bias else None
    r = torch.randn(m, n, dtype=dtype) if resid else None
    nw = (1.0 + 0.1 * torch.randn(n)).to(dtype)
    eps = 1e-5
    dev = lambda t: None if t is None else t.to(DEV)
    normed, summed = dense_gemm_residual_rmsnorm(dev(x), dev(w), dev(b), dev(r), dev(nw), eps)
    if k >= 4096 and m >= 100:
        assert L.last_launch() == "gemm_skinny:splitk->resnorm"                # (the norm that sums the slabs; the product before it:)
    a = dense_gemm(dev(x), dev(w), dev(b), False)
    if k >= 4096 and m >= 100:
        assert L.last_launch() == "gemm128:128x128:NK:splitk", L.last_launch()
    if resid:
        op = hip_cls("MojoResidualAddRMSNorm")(n, eps, "pre", dtype=dtype, device=DEV)
        op.weight.data.copy_(nw)
        want_n, want_s = op(a, dev(r))
        assert torch.equal(summed, want_s)
    else:
        op = hip_cls("MojoRMSNorm")(n, eps, dtype=dtype, device=DEV)
        op.weight.data.copy_(nw)
        want_n = op(a)
        assert summed is None
    assert torch.equal(normed, want_n)
    # the oracle's chain (fp32 product rounded once, then the golden norm)
    a_ref = torch.nn.functional.linear(x.float(), w.float(), b.float() if bias else None).to(dtype)
    if resid:
        ref = torch_cls("MojoResidualAddRMSNorm")(n, eps, "pre", dtype=dtype)
        ref.weight.data.copy_(nw)
        ref_n, ref_s = ref(a_ref, r)
        torch.testing.assert_close(to_cpu(summed).float(), ref_s.float(), atol=3e-2, rtol=2e-2)
    else:
        ref = torch_cls("MojoRMSNorm")(n, eps, dtype=dtype)
        ref.weight.data.copy_(nw)
        ref_n = ref(a_ref)
    torch.testing.assert_close(to_cpu(normed).float(), ref_n.float(), atol=3e-2, rtol=2e-2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("b,k,hq,hkv,d,page,bias", [(64, 4096, 32, 8, 128, 16, False), (5, 1024, 8, 2, 64, 8, True), (1, 512, 4, 4, 128, 16, True),
                                                   (33, 256, 16, 1, 32, 4, False),
                                                   (64, 192, 4, 2, 64, 16, False)])          # K % 128: the projection takes the tile kernel
def test_qkv_rope_store_fused_equals_the_separate_calls_and_the_oracle(dtype, b, k, hq, hkv, d, page, bias):
    """`mojo_hip_qkv_rope_store` against dense_gemm -> HIPApplyRoPE -> HIPStorePagedKVCache (same bits, caches included: rows
    with a negative context length or a missing page stay untouched) and against the oracle's chain."""
    from hip_utils import hip_cls, torch_cls
    from mojo_opset_amd.backends.hip.operators.gemm import qkv_rope_store
    g = torch.Generator().manual_seed(b * 7 + d)
    n = (hq + 2 * hkv) * d
    x = torch.randn(b, k, generator=g).to(dtype)
    w = (torch.randn(n, k, generator=g) / k ** 0.5).to(dtype)
    bs = torch.randn(n, generator=g).to(dtype) if bias else None
    cos, sin = torch.randn(b, d, generator=g), torch.randn(b, d, generator=g)
    pages_per_seq = 6
    n_blocks = b * pages_per_seq + 3
    table = torch.randperm(n_blocks, generator=g)[: b * pages_per_seq].view(b, pages_per_seq).to(torch.int32)
    ctx = torch.randint(0, pages_per_seq * page, (b,), generator=g).to(torch.int32)
    if b >= 5:
        ctx[1] = -1                                           # a padded row: nothing stored
        table[2, int(ctx[2]) // page] = -1                    # a hole in the table: nothing stored
        ctx[3] = pages_per_seq * page + 2                     # past the table: nothing stored
    kc0 = torch.randn(n_blocks, hkv, page, d, generator=g).to(dtype)
    vc0 = torch.randn(n_blocks, hkv, page, d, generator=g).to(dtype)
    dev = lambda t: None if t is None else t.to(DEV)
    kc, vc = dev(kc0).clone(), dev(vc0).clone()
    q_got = qkv_rope_store(dev(x), dev(w), dev(bs), dev(cos), dev(sin), kc, vc, dev(table), dev(ctx), hq, hkv)
    # the separate calls
    qkv = dense_gemm(dev(x), dev(w), dev(bs), False)
    q = qkv[:, : hq * d].reshape(b, hq, d)
    kk = qkv[:, hq * d: (hq + hkv) * d].reshape(b, hkv, d)
    v = qkv[:, (hq + hkv) * d:].reshape(b, hkv, d).contiguous()
    q_r, k_r = hip_cls("MojoApplyRoPE")()(q.unsqueeze(0), kk.unsqueeze(0), dev(cos), dev(sin), head_first=False)
    kc2, vc2 = dev(kc0).clone(), dev(vc0).clone()
    hip_cls("MojoStorePagedKVCache")()(k_r.squeeze(0).contiguous(), v, kc2, vc2, dev(table), None, dev(ctx))
    assert torch.equal(q_got, q_r.squeeze(0))
    assert torch.equal(kc, kc2) and torch.equal(vc, vc2)
    assert not torch.equal(kc, dev(kc0))                      # (something was stored)
    # the oracle's chain
    qkv_ref = torch.nn.functional.linear(x.float(), w.float(), bs.float() if bias else None).to(dtype)
    q_ref, k_ref = torch_cls("MojoApplyRoPE")()(qkv_ref[:, : hq * d].reshape(1, b, hq, d), qkv_ref[:, hq * d: (hq + hkv) * d].reshape(1, b, hkv, d),
                                                cos, sin, head_first=False)
    torch.testing.assert_close(to_cpu(q_got).float(), q_ref.squeeze(0).float(), atol=3e-2, rtol=2e-2)
    live = [i for i in range(b) if 0 <= int(ctx[i]) < pages_per_seq * page and int(table[i, int(ctx[i]) // page]) >= 0]
    for i in live[:8]:
        blk, slot = int(table[i, int(ctx[i]) // page]), int(ctx[i]) % page
        torch.testing.assert_close(to_cpu(kc[blk, :, slot]).float(), k_ref[0, i].float(), atol=3e-2, rtol=2e-2)
        torch.testing.assert_close(to_cpu(vc[blk, :, slot]).float(), qkv_ref[i, (hq + hkv) * d:].reshape(hkv, d).float(), atol=3e-2, rtol=2e-2)


@pytest.mark.parametrize("seed", range(12))
def test_decode_fusions_random_shapes_equal_the_separate_calls(seed):
    """Seeded random shapes (further seeds with MOJO_FUZZ_OFFSET): every fused form against its row of separate calls."""
    import os
    import random

    from hip_utils import hip_cls
    from mojo_opset_amd.backends.hip.operators.gemm import dense_gemm_residual_rmsnorm, dense_gemm_swiglu, qkv_rope_store
    rnd = random.Random(9000 + seed + 1000 * int(os.environ.get("MOJO_FUZZ_OFFSET", "0")))
    dtype = rnd.choice([torch.bfloat16, torch.float16])
    g = torch.Generator().manual_seed(seed)
    # gemm + swiglu (m <= 64 takes the fused kernel when K % 128 == 0; anything else the two-launch route)
    m, k, inter = rnd.randint(1, 80), rnd.choice([128, 256, 384, 1024, 200]), 8 * rnd.randint(1, 300)
    x = torch.randn(m, k, generator=g).to(dtype).to(DEV)
    w = (torch.randn(2 * inter, k, generator=g) / k ** 0.5).to(dtype).to(DEV)
    got = dense_gemm_swiglu(x, w)
    chain = _swiglu_chain(x, w)
    if L.load().mojo_hip_gemm_workspace_bytes(m, k, 2 * inter) <= 64:
        assert torch.equal(got, chain), (m, k, inter)
    else:
        assert max_ulp_bf16ish(to_cpu(got), to_cpu(chain), atol=2e-3) <= 2, (m, k, inter)
    # gemm + residual rmsnorm
    m, k, n = rnd.randint(1, 140), 128 * rnd.randint(1, 64), 64 * rnd.randint(1, 96)
    x = torch.randn(m, k, generator=g).to(dtype).to(DEV)
    w = (torch.randn(n, k, generator=g) / k ** 0.5).to(dtype).to(DEV)
    r = torch.randn(m, n, generator=g).to(dtype).to(DEV)
    nw = (1 + 0.1 * torch.randn(n, generator=g)).to(dtype).to(DEV)
    normed, summed = dense_gemm_residual_rmsnorm(x, w, None, r, nw, 1e-5)
    op = hip_cls("MojoResidualAddRMSNorm")(n, 1e-5, "pre", dtype=dtype, device=DEV)
    op.weight.data.copy_(nw)
    want_n, want_s = op(dense_gemm(x, w, None, False), r)
    assert torch.equal(summed, want_s) and torch.equal(normed, want_n), (m, k, n)
    # qkv + rope + store
    b, hkv, grp, d, page = rnd.randint(1, 70), rnd.choice([1, 2, 4, 8]), rnd.choice([1, 2, 4, 8]), rnd.choice([32, 64, 128]), rnd.choice([4, 8, 16, 32])
    hq, k = hkv * grp, 128 * rnd.randint(1, 24)
    n = (hq + 2 * hkv) * d
    x = torch.randn(b, k, generator=g).to(dtype).to(DEV)
    w = (torch.randn(n, k, generator=g) / k ** 0.5).to(dtype).to(DEV)
    cos, sin = torch.randn(b, d, generator=g).to(DEV), torch.randn(b, d, generator=g).to(DEV)
    pages = 5
    n_blocks = b * pages + 2
    table = torch.randperm(n_blocks, generator=g)[: b * pages].view(b, pages).to(torch.int32)
    ctx = torch.randint(-1, pages * page + 2, (b,), generator=g).to(torch.int32)
    table[rnd.randrange(b), rnd.randrange(pages)] = -1
    table, ctx = table.to(DEV), ctx.to(DEV)
    kc0 = torch.randn(n_blocks, hkv, page, d, generator=g).to(dtype).to(DEV)
    vc0 = torch.randn(n_blocks, hkv, page, d, generator=g).to(dtype).to(DEV)
    kc, vc = kc0.clone(), vc0.clone()
    q_got = qkv_rope_store(x, w, None, cos, sin, kc, vc, table, ctx, hq, hkv)
    qkv = dense_gemm(x, w, None, False)
    q_r, k_r = hip_cls("MojoApplyRoPE")()(qkv[:, : hq * d].reshape(1, b, hq, d), qkv[:, hq * d: (hq + hkv) * d].reshape(1, b, hkv, d), cos, sin, head_first=False)
    kc2, vc2 = kc0.clone(), vc0.clone()
    hip_cls("MojoStorePagedKVCache")()(k_r.squeeze(0).contiguous(), qkv[:, (hq + hkv) * d:].reshape(b, hkv, d).contiguous(), kc2, vc2, table, None, ctx)
    assert torch.equal(q_got, q_r.squeeze(0)) and torch.equal(kc, kc2) and torch.equal(vc, vc2), (b, hq, hkv, d, page, k)
