"""GPU parity of MojoQuantGemm through the C ABI.

int8: the accelerated result must equal the exact integer formula — atol = rtol = 0 — exactly as the
reference demands of its golden (mojo_opset/tests/accuracy/operators/test_gemm.py:95-111).
fp8-e4m3 (extension, parity unpinned): compared with the oracle restatement at fp32-accumulate tolerance."""
import pytest
import torch

from conftest import bit_equal, load_golden
from hip_utils import DEV, hip_cls, last_launch, run_hip_case, to_cpu, torch_cls
from oracle import quant_gemm_formula

pytestmark = pytest.mark.gpu


def _quantize(x):
    scale = x.abs().amax(dim=-1).clamp_min(1e-8) / 127.0
    return torch.clamp(torch.round(x / scale.unsqueeze(-1)), -128, 127).to(torch.int8), scale


@pytest.mark.parametrize("case", [pytest.param(c, id=f"qg-{i}") for i, c in enumerate(load_golden("quant_gemm"))])
def test_quant_gemm_vectors_bit_exact(case):
    assert bit_equal(to_cpu(run_hip_case(case)), case["out"])


@pytest.mark.parametrize("m,k,n", [(1, 4096, 4096), (32, 4096, 11008), (128, 2048, 4096), (64, 4096, 4096),
                                   (300, 1024, 1000), (257, 128, 264), (5, 96, 40),
                                   # decode-sized M on the weight-streaming kernel ([N,K] weights): every row-tile count,
                                   # ragged M, one K block per slice, no split, DeepSeek-V3 shapes
                                   (16, 512, 64), (17, 7168, 4096), (33, 1536, 7168), (48, 256, 128), (64, 18432, 7168),
                                   (3, 2048, 64), (65, 1024, 128), (128, 7168, 4096), (100, 512, 64),
                                   # M <= 4: the GEMV kernel (one wave per two weight rows), ragged K and N
                                   (1, 7168, 4096), (2, 1536, 7168), (4, 18432, 64), (3, 272, 10), (1, 48, 3)])
@pytest.mark.parametrize("trans_weight", [False, True])
@pytest.mark.parametrize("odt", [torch.bfloat16, torch.float16, torch.float32])
def test_quant_gemm_int8_equals_integer_formula(m, k, n, trans_weight, odt):
    torch.manual_seed(0)
    xq, xs = _quantize(torch.randn(m, k))
    wq, ws = _quantize(torch.randn(n, k))
    op = hip_cls("MojoQuantGemm")(in_features=k, out_features=n, output_dtype=odt, trans_weight=trans_weight, device=DEV)
    op.weight.copy_(wq if trans_weight else wq.t())
    op.weight_scale.copy_(ws.to(torch.bfloat16))
    scale = xs if (m + k) % 2 else xs.unsqueeze(-1)
    out = to_cpu(op(xq.to(DEV), scale.to(DEV)))
    expect = quant_gemm_formula(xq, wq.t(), xs, ws.to(torch.bfloat16), odt)
    torch.testing.assert_close(out, expect, atol=0, rtol=0)
    assert set(op.state_dict()) == {"weight", "weight_scale"}


def test_quant_gemm_int8_extreme_values_accumulate_in_int32():
    """+-127/128 everywhere: |sum| = K*16384 > 2^24, where an fp32 running sum would round — the int32
    accumulator must stay exact (the oracle's float64 formula is the arbiter here)."""
    m, k, n = 64, 8192, 256
    xq = torch.full((m, k), -128, dtype=torch.int8)
    wq = torch.full((n, k), 127, dtype=torch.int8)
    wq[::2] = -128
    op = hip_cls("MojoQuantGemm")(k, n, output_dtype=torch.float32, trans_weight=True, device=DEV)
    op.weight.copy_(wq)
    op.weight_scale.fill_(1.0)
    out = to_cpu(op(xq.to(DEV), torch.ones(m, device=DEV)))
    expect = (xq.double() @ wq.double().t()).float()
    assert torch.equal(out, expect)


def test_quant_gemm_error_conventions():
    op = hip_cls("MojoQuantGemm")(64, 32, device=DEV)
    with pytest.raises(ValueError):
        op(torch.zeros(2, 3, 64, dtype=torch.int8, device=DEV), torch.ones(2, device=DEV))
    with pytest.raises(ValueError):
        op(torch.zeros(2, 48, dtype=torch.int8, device=DEV), torch.ones(2, device=DEV))


@pytest.mark.parametrize("m,k,n", [(128, 7168, 1536), (32, 2048, 7168), (1, 512, 256), (300, 1024, 1000), (7, 96, 40),
                                   (17, 7168, 4096), (64, 1536, 7168), (1, 7168, 4096), (4, 2048, 130)])
@pytest.mark.parametrize("trans_weight", [False, True])
def test_quant_gemm_fp8_matches_oracle(m, k, n, trans_weight):
    """Extension dtype — parity unpinned: there is no reference implementation (gemm.py:171-173 asserts int8)."""
    torch.manual_seed(1)
    f8 = torch.float8_e4m3fn
    x = torch.randn(m, k).to(f8)
    w_nk = torch.randn(n, k).to(f8)
    s_in, s_w = torch.rand(m) + 0.5, (torch.rand(n) + 0.5).to(torch.bfloat16)
    kw = dict(in_features=k, out_features=n, output_dtype=torch.bfloat16, trans_weight=trans_weight, quant_dtype=f8, weight_dtype=f8)
    ref = torch_cls("MojoQuantGemm")(**kw)
    op = hip_cls("MojoQuantGemm")(**kw, device=DEV)
    for o in (ref, op):
        o.weight.copy_((w_nk if trans_weight else w_nk.t().contiguous()).to(o.weight.device))
        o.weight_scale.copy_(s_w)
    want = ref(x, s_in)
    got = to_cpu(op(x.to(DEV), s_in.to(DEV)))
    exact = quant_gemm_formula(x, w_nk.t(), s_in, s_w, torch.bfloat16)
    torch.testing.assert_close(got.float(), exact.float(), atol=2e-2 * k ** 0.5, rtol=2 ** -7)
    torch.testing.assert_close(got.float(), want.float(), atol=2e-2 * k ** 0.5, rtol=2 ** -6)


def test_quant_gemm_fp8_small_integers_exact():
    """fp8 values that are small integers: products and sums are exact, so the operand mapping of the
    fp8 MFMA and both weight layouts are checked element for element."""
    f8 = torch.float8_e4m3fn
    g = torch.Generator().manual_seed(3)
    m, k, n = 300, 512, 520
    x = torch.randint(-3, 4, (m, k), generator=g).float()
    w = torch.randint(-3, 4, (n, k), generator=g).float()
    for trans in (False, True):
        op = hip_cls("MojoQuantGemm")(k, n, output_dtype=torch.float32, trans_weight=trans, quant_dtype=f8, weight_dtype=f8, device=DEV)
        op.weight.copy_((w if trans else w.t().contiguous()).to(f8))
        op.weight_scale.fill_(1.0)
        out = to_cpu(op(x.to(f8).to(DEV), torch.ones(m, device=DEV)))
        assert torch.equal(out, x @ w.t())


@pytest.mark.parametrize("quant_dtype", [torch.int8, torch.float8_e4m3fn])
@pytest.mark.parametrize("odt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,k,n", [(300, 512, 768), (1024, 1024, 512), (257, 2048, 1280)])
def test_quant_gemm_row_staged_epilogue_is_bit_identical_to_direct_stores(m, k, n, odt, quant_dtype, monkeypatch):
    """Full N tiles of the 256x256 kernel leave through the row-staged epilogue (scales applied on the way into the
    wave-private LDS transpose); MOJO_HIP_GEMM_STAGE_ROWS=0 stores directly.  Same bits."""
    torch.manual_seed(13)
    op = hip_cls("MojoQuantGemm")(k, n, output_dtype=odt, trans_weight=True, quant_dtype=quant_dtype, weight_dtype=quant_dtype, device=DEV)
    if quant_dtype == torch.int8:
        op.weight.copy_(torch.randint(-127, 128, (n, k), dtype=torch.int8, device=DEV))
        x = torch.randint(-127, 128, (m, k), dtype=torch.int8, device=DEV)
    else:
        op.weight.copy_(torch.randn(n, k, device=DEV).to(quant_dtype))
        x = torch.randn(m, k, device=DEV).to(quant_dtype)
    op.weight_scale.copy_(torch.rand(n, device=DEV) * 0.02)
    s_in = torch.rand(m, device=DEV)
    # (few output tiles would otherwise take the split-K route, whose slabs go through a finalize kernel — NOT this epilogue:
    # round 4's version of this test compared that route with itself)
    monkeypatch.setenv("MOJO_HIP_GEMM_SPLITK", "1")
    monkeypatch.setenv("MOJO_HIP_GEMM_TILE128", "0")             # (and not the 128-row-tile kernel, which has its own epilogue)
    staged = op(x, s_in)
    assert last_launch().startswith("gemm256:staged"), last_launch()
    monkeypatch.setenv("MOJO_HIP_GEMM_STAGE_ROWS", "0")          # (the fixture makes the library re-read its switches)
    direct = op(x, s_in)
    assert last_launch().startswith("gemm256:direct"), last_launch()         # the OTHER epilogue really ran
    assert torch.equal(staged, direct)
    if quant_dtype == torch.int8:                                 # the direct-store epilogue on its own: the integer formula, exactly
        exact = quant_gemm_formula(x.cpu(), op.weight.cpu().t(), s_in.cpu(), op.weight_scale.cpu(), odt)
        torch.testing.assert_close(to_cpu(direct), exact, atol=0, rtol=0)


# ---- BASELINE config 5 at full size (DeepSeek-V3 dense projections, M = 4096) -------------------------------------------
FULL_SHAPES = [(4096, 7168, 36864), (4096, 18432, 7168)]


def _sample_rows(m, g):
    return torch.cat([torch.arange(0, 16), torch.arange(m - 16, m), torch.randint(16, m - 16, (96,), generator=g)]).unique()


def _exact_rows(x_rows, w_nk, s_in_rows, s_w, odt):
    """The oracle's formula (`oracle.quant_gemm_formula`: float64 accumulation — exact for int8 — then the golden's fp32
    scaling and one rounding), with the float64 product evaluated by torch ON THE DEVICE: 128 x K x N in float64 takes the
    box's 16-core CPU share minutes, and a float64 sum of products below 2^53 is the same number wherever it is added up."""
    acc = (x_rows.to(DEV).double() @ w_nk.to(DEV).double().t()).cpu()
    return (acc.float() * s_in_rows.float().reshape(-1, 1).cpu() * s_w.float().reshape(1, -1).cpu()).to(odt)


@pytest.mark.parametrize("m,k,n", FULL_SHAPES)
def test_quant_gemm_int8_full_size_exact(m, k, n):
    """int8 at the benchmarked shapes.  The exact integer formula on a sample of whole rows (first / last 16 and 96 random
    ones) must match at atol = rtol = 0; every other element is covered by a checksum of checksums: with unit scales and
    fp32 output each element is the exact integer sum, so the column sums of the whole [M, N] result must equal
    (sum_m x[m, :]) @ W exactly."""
    g = torch.Generator().manual_seed(k + n)
    gd = torch.Generator(device=DEV).manual_seed(k + n)
    xq, xs = _quantize(torch.randn(m, k, generator=gd, device=DEV))
    wq, ws = _quantize(torch.randn(n, k, generator=gd, device=DEV))
    rows = _sample_rows(m, g).to(DEV)
    op = hip_cls("MojoQuantGemm")(in_features=k, out_features=n, output_dtype=torch.bfloat16, trans_weight=True, device=DEV)
    op.weight.copy_(wq)
    op.weight_scale.copy_(ws.to(torch.bfloat16))
    out = op(xq, xs)
    expect = _exact_rows(xq[rows], wq, xs[rows], ws.to(torch.bfloat16), torch.bfloat16)
    # the helper must be the oracle's formula: cross-check on a few rows against the CPU implementation itself
    few = rows[:4]
    assert torch.equal(_exact_rows(xq[few], wq, xs[few], ws.to(torch.bfloat16), torch.bfloat16),
                       quant_gemm_formula(xq[few].cpu(), wq.cpu().t(), xs[few].cpu(), ws.to(torch.bfloat16).cpu(), torch.bfloat16))
    torch.testing.assert_close(to_cpu(out[rows]), expect, atol=0, rtol=0)
    del out
    op32 = hip_cls("MojoQuantGemm")(in_features=k, out_features=n, output_dtype=torch.float32, trans_weight=True, device=DEV)
    op32.weight.copy_(wq)
    op32.weight_scale.fill_(1.0)
    raw = op32(xq, torch.ones(m, device=DEV))
    assert float(raw.abs().max()) < 2 ** 24                     # every element is an exactly represented integer
    col = raw.double().sum(0)
    want = xq.double().sum(0) @ wq.double().t()
    assert torch.equal(col, want)
    assert torch.equal(raw[rows], (xq[rows].double() @ wq.double().t()).float())


@pytest.mark.parametrize("m,k,n", FULL_SHAPES)
def test_quant_gemm_fp8_full_size(m, k, n):
    """fp8-e4m3 (extension, PARITY UNPINNED: no reference implementation) at the benchmarked shapes: sampled whole rows
    against the float64 formula, and the column-sum identity to fp32-accumulation tolerance."""
    f8 = torch.float8_e4m3fn
    g = torch.Generator().manual_seed(k + n + 1)
    gd = torch.Generator(device=DEV).manual_seed(k + n + 1)
    x = torch.randn(m, k, generator=gd, device=DEV).to(f8)
    w = torch.randn(n, k, generator=gd, device=DEV).to(f8)
    s_in = torch.rand(m, generator=gd, device=DEV) + 0.5
    s_w = (torch.rand(n, generator=gd, device=DEV) + 0.5).to(torch.bfloat16)
    rows = _sample_rows(m, g).to(DEV)
    op = hip_cls("MojoQuantGemm")(in_features=k, out_features=n, output_dtype=torch.bfloat16, trans_weight=True,
                                  quant_dtype=f8, weight_dtype=f8, device=DEV)
    op.weight.copy_(w)
    op.weight_scale.copy_(s_w)
    out = op(x, s_in)
    exact = _exact_rows(x[rows], w, s_in[rows], s_w, torch.bfloat16)
    torch.testing.assert_close(to_cpu(out[rows]).float(), exact.float(), atol=2e-2 * k ** 0.5, rtol=2 ** -7)
    op32 = hip_cls("MojoQuantGemm")(in_features=k, out_features=n, output_dtype=torch.float32, trans_weight=True,
                                    quant_dtype=f8, weight_dtype=f8, device=DEV)
    op32.weight.copy_(w)
    op32.weight_scale.fill_(1.0)
    raw = op32(x, torch.ones(m, device=DEV))
    col = raw.double().sum(0)
    want = x.double().sum(0) @ w.double().t()
    torch.testing.assert_close(col, want, atol=1e-3 * (m * k) ** 0.5, rtol=1e-5)


@pytest.mark.parametrize("quant_dtype", [torch.int8, torch.float8_e4m3fn])
def test_quant_gemm_split_k_route_is_deterministic_and_exact(quant_dtype, monkeypatch):
    """M >= 256 with few output tiles: each K slice writes its own fp32 / int32 slab and a finalize kernel sums the slabs in
    a fixed order.  Forced here with 1 (no split), 2 and 8 slices: int8 is bit-identical across all of them and to the
    integer formula; fp8 is bit-stable run to run for a given split and within tolerance of the float64 formula."""
    m, k, n = 512, 4096, 520
    g = torch.Generator().manual_seed(21)
    if quant_dtype == torch.int8:
        x, xs = _quantize(torch.randn(m, k, generator=g))
        w, ws = _quantize(torch.randn(n, k, generator=g))
    else:
        x, w = torch.randn(m, k, generator=g).to(quant_dtype), torch.randn(n, k, generator=g).to(quant_dtype)
        xs, ws = torch.rand(m, generator=g) + 0.5, torch.rand(n, generator=g) + 0.5
    op = hip_cls("MojoQuantGemm")(k, n, output_dtype=torch.bfloat16, trans_weight=True, quant_dtype=quant_dtype,
                                  weight_dtype=quant_dtype, device=DEV)
    op.weight.copy_(w)
    op.weight_scale.copy_(ws.to(torch.bfloat16))
    exact = quant_gemm_formula(x, w.t(), xs, ws.to(torch.bfloat16), torch.bfloat16)
    outs = {}
    for sk in ("1", "2", "8"):
        monkeypatch.setenv("MOJO_HIP_GEMM_SPLITK", sk)
        a = op(x.to(DEV), xs.to(DEV))
        b = op(x.to(DEV), xs.to(DEV))
        assert torch.equal(a, b), f"split {sk}: not deterministic"
        outs[sk] = to_cpu(a)
        if quant_dtype == torch.int8:
            torch.testing.assert_close(outs[sk], exact, atol=0, rtol=0)
        else:
            torch.testing.assert_close(outs[sk].float(), exact.float(), atol=2e-2 * k ** 0.5, rtol=2 ** -7)
    if quant_dtype == torch.int8:
        assert torch.equal(outs["1"], outs["2"]) and torch.equal(outs["1"], outs["8"])


@pytest.mark.parametrize("quant_dtype", [torch.int8, torch.float8_e4m3fn])
@pytest.mark.parametrize("odt", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("m,k,n", [(129, 128, 128), (300, 384, 520), (1024, 7168, 1536), (513, 1024, 4104), (2048, 512, 2304)])
@pytest.mark.parametrize("shape", ["128", "256"])
@pytest.mark.parametrize("trans_weight", [True, False])
def test_quant_gemm_tile128_equals_the_256_tile_kernel_and_the_integer_formula(m, k, n, odt, quant_dtype, shape, trans_weight, monkeypatch):
    """Mid-size M with `[N,K]` weights (`trans_weight=True`): the 128-row-tile kernel (csrc/gemm_tile128_core.h; both tile
    widths forced in turn) against the unsplit 256 x 256 kernel — int32 accumulation is exact, and the fp8 products are summed by
    the same instruction in the same order, so the outputs must be the same bits; int8 also equals the integer formula exactly.
    Ragged M and N, one and odd K-tile counts."""
    torch.manual_seed(m + n)
    if not trans_weight:
        if quant_dtype != torch.int8:
            pytest.skip("(K, N) weights on the 128-row tiles: int8 only (the reference's operator is int8 only; fp8 stays on the 256 x 256 kernel)")
        n = (n + 15) // 16 * 16                                       # [K, N] rows are read in 16-byte pieces
    op = hip_cls("MojoQuantGemm")(k, n, output_dtype=odt, trans_weight=trans_weight, quant_dtype=quant_dtype, weight_dtype=quant_dtype, device=DEV)
    if quant_dtype == torch.int8:
        w_nk = torch.randint(-127, 128, (n, k), dtype=torch.int8, device=DEV)
        x = torch.randint(-127, 128, (m, k), dtype=torch.int8, device=DEV)
    else:
        w_nk = torch.randn(n, k, device=DEV).to(quant_dtype)
        x = torch.randn(m, k, device=DEV).to(quant_dtype)
    op.weight.copy_(w_nk if trans_weight else w_nk.t())               # (N, K) with trans_weight, else the operator's default (K, N)
    op.weight_scale.copy_(torch.rand(n, device=DEV) * 0.02)
    s_in = torch.rand(m, device=DEV)
    monkeypatch.setenv("MOJO_HIP_GEMM_TILE128", shape)
    monkeypatch.setenv("MOJO_HIP_GEMM_SPLITK", "1")
    small = op(x, s_in)
    assert last_launch() == f"gemm128:128x{shape}:" + ("NK" if trans_weight else "KN"), last_launch()
    monkeypatch.setenv("MOJO_HIP_GEMM_TILE128", "0")
    large = op(x, s_in)
    assert last_launch().startswith("gemm256:") and ":splitk" not in last_launch(), last_launch()
    assert torch.equal(small, large)
    if quant_dtype == torch.int8:
        exact = quant_gemm_formula(x.cpu(), w_nk.cpu().t(), s_in.cpu(), op.weight_scale.cpu(), odt)
        torch.testing.assert_close(to_cpu(small), exact, atol=0, rtol=0)


@pytest.mark.parametrize("quant_dtype", [torch.int8, torch.float8_e4m3fn])
@pytest.mark.parametrize("odt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("m,k,n,sk", [(256, 8192, 1024, 0), (200, 4096, 4096, 0),      # the model's own split
                                      (129, 384, 132, 3), (300, 2048, 520, 5), (32, 4096, 1024, 4)])   # forced: one K-tile per slice, uneven slices, decode rows ([K,N] only)
@pytest.mark.parametrize("trans_weight", [True, False])
def test_quant_gemm_tile128_split_k(m, k, n, sk, odt, quant_dtype, trans_weight, monkeypatch):
    """Few 128 x 128 tiles over a long K: the tiles' own K split — int32 / fp32 slabs, summed in slice order and dequantised by
    quant_finalize_kernel.  int8: int32 sums are exact, so the result equals the integer formula and the unsplit kernel to the
    bit; fp8: the fp32 slices are added in another order than the unsplit K loop — within the operator's fp8 bound."""
    torch.manual_seed(m + n)
    if not trans_weight:
        if quant_dtype != torch.int8:
            pytest.skip("(K, N) weights on the 128-row tiles: int8 only")
        n = (n + 15) // 16 * 16
    elif m <= 64:
        pytest.skip("[N,K] weights and at most 64 rows: the weight-streaming kernel")
    op = hip_cls("MojoQuantGemm")(k, n, output_dtype=odt, trans_weight=trans_weight, quant_dtype=quant_dtype, weight_dtype=quant_dtype, device=DEV)
    if quant_dtype == torch.int8:
        w_nk = torch.randint(-127, 128, (n, k), dtype=torch.int8, device=DEV)
        x = torch.randint(-127, 128, (m, k), dtype=torch.int8, device=DEV)
    else:
        w_nk = torch.randn(n, k, device=DEV).to(quant_dtype)
        x = torch.randn(m, k, device=DEV).to(quant_dtype)
    op.weight.copy_(w_nk if trans_weight else w_nk.t())
    op.weight_scale.copy_(torch.rand(n, device=DEV) * 0.02)
    s_in = torch.rand(m, device=DEV)
    if sk:
        monkeypatch.setenv("MOJO_HIP_GEMM_TILE128", "1")
        monkeypatch.setenv("MOJO_HIP_GEMM_SPLITK", str(sk))
    split = op(x, s_in)
    assert last_launch() == "gemm128:128x128:" + ("NK" if trans_weight else "KN") + ":splitk", last_launch()
    assert torch.equal(split, op(x, s_in))
    monkeypatch.setenv("MOJO_HIP_GEMM_TILE128", "1")
    monkeypatch.setenv("MOJO_HIP_GEMM_SPLITK", "1")
    whole = op(x, s_in)
    assert last_launch() == "gemm128:128x128:" + ("NK" if trans_weight else "KN"), last_launch()
    if quant_dtype == torch.int8:
        assert torch.equal(split, whole)
        exact = quant_gemm_formula(x.cpu(), w_nk.cpu().t(), s_in.cpu(), op.weight_scale.cpu(), odt)
        torch.testing.assert_close(to_cpu(split), exact, atol=0, rtol=0)
    else:
        torch.testing.assert_close(to_cpu(split).float(), to_cpu(whole).float(), atol=2e-2 * k ** 0.5 * 0.02, rtol=2 ** -7)


def test_quant_gemm_tile128_default_choice():
    """Taken by default where the time model prefers it (few 256 x 256 tiles, more than 128 rows, either weight layout);
    chip-filling launches stay on the 256 x 256 kernel."""
    def form(m, k, n, trans):
        op = hip_cls("MojoQuantGemm")(k, n, output_dtype=torch.bfloat16, trans_weight=trans, device=DEV)
        op.weight.copy_(torch.randint(-127, 128, (n, k) if trans else (k, n), dtype=torch.int8, device=DEV))
        op.weight_scale.fill_(0.01)
        op(torch.randint(-127, 128, (m, k), dtype=torch.int8, device=DEV), torch.rand(m, device=DEV))
        return last_launch()
    assert form(1024, 4096, 4096, True) == "gemm128:128x128:NK"
    assert form(2048, 4096, 4096, True) == "gemm128:128x256:NK"
    assert form(1024, 4096, 4096, False) == "gemm128:128x128:KN"      # the operator's default (K, N) layout
    assert form(8192, 1024, 8192, True).startswith("gemm256:")
