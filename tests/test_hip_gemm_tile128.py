"""GPU parity of the 128-row-tile dense GEMM (`csrc/gemm_tile128.hip`): the form `mojo_hip_gemm` / `mojo_hip_gemm_rowmap`
take for one 16-bit product — `[N, K]` weights (the golden's `F.linear(input, weight, bias)`, core/operators/gemm.py:45-46) or
`[K, N]` weights (`input @ weight (+ bias)`, compute_with_comm.py:12-24 with trans_weight) — whose 256 x 256 tiles would leave
most of the chip idle: a prefill chunk of a few hundred to two thousand rows.

Exactness: small-integer data makes every product and partial sum exact in fp32, so the result must equal the fp32 reference
to the bit whatever the tile shape; on random data the 128-tile form and the unsplit 256-tile form add the same products in
the same order (K in index order, fp32), so their outputs must be the same bits (an A/B with the launch form asserted on both
legs).
"""
import pytest
import torch
import torch.nn.functional as F

from hip_utils import DEV, last_launch, max_ulp_bf16ish, switch_env, to_cpu
from mojo_opset_amd.backends.hip.operators.compute_with_comm import HipGemmEngine
from mojo_opset_amd.backends.hip.operators.gemm import dense_gemm

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,k,n,bias", [
    (129, 64, 128, False),          # one K-tile, one row past a tile
    (256, 192, 384, True),          # odd K-tile count
    (300, 1024, 520, True),         # ragged in M and N (N % 128 = 8)
    (1000, 4096, 4096, False),      # 8 x 32 tiles = one per CU
    (1024, 8192, 1024, True),       # 64 tiles
    (2048, 512, 4100, False),       # 16 x 33 tiles of 128 x 128 > CUs: 128 x 256 tiles, eight waves, three-stage ring; N % 256 = 4
    (513, 320, 260, True),
])
@pytest.mark.parametrize("shape", ["1", "128", "256"])
@pytest.mark.parametrize("layout", ["NK", "KN"])
def test_tile128_integer_data_is_exact(dtype, m, k, n, bias, shape, layout):
    torch.manual_seed(m + n)
    if layout == "KN":
        n = (n + 7) // 8 * 8                                        # [K, N] rows are read in 16-byte pieces
    x = torch.randint(-4, 5, (m, k)).to(dtype).to(DEV)
    w = torch.randint(-4, 5, (n, k)).to(dtype).to(DEV)
    b = torch.randint(-8, 9, (n,)).to(dtype).to(DEV) if bias else None
    if layout == "NK":
        want = F.linear(x.float(), w.float(), None if b is None else b.float()).to(dtype)  # F.linear: ONE rounding, bias included
    else:
        w = w.t().contiguous()                                      # [K, N]: `x @ w` is rounded, then `+ b` is (two operators)
        want = (x.float() @ w.float()).to(dtype)
        want = want if b is None else (want.float() + b.float()).to(dtype)
    with switch_env(MOJO_HIP_GEMM_TILE128=shape, MOJO_HIP_GEMM_SPLITK="1"):   # 1: the launcher's own choice of shape; 128 / 256: forced
        got = dense_gemm(x, w, b, layout == "KN")
        form = last_launch()
    wide = shape == "256" or (shape == "1" and ((m + 127) // 128) * ((n + 127) // 128) > 256)
    assert form == ("gemm128:128x256:" if wide else "gemm128:128x128:") + layout, form
    assert torch.equal(got, want)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,k,n,bias", [(1024, 4096, 4096, True), (384, 2048, 6144, False), (1500, 1024, 1000, True), (2048, 256, 8192, False)])
@pytest.mark.parametrize("shape", ["128", "256"])
@pytest.mark.parametrize("layout", ["NK", "KN"])
def test_tile128_gives_the_bits_of_the_unsplit_256_tile_kernel(dtype, m, k, n, bias, shape, layout):
    """Random data, both forms forced and asserted: same products, same fp32 order, same rounding -> the same bits; and both
    within one unit in the last place of the fp32 reference."""
    torch.manual_seed(k + n)
    x = torch.randn(m, k, dtype=dtype, device=DEV)
    w = (torch.randn(n, k, device=DEV) * 0.05).to(dtype)
    b = torch.randn(n, device=DEV).to(dtype) if bias else None
    trans = layout == "KN"
    if trans:
        w = w.t().contiguous()
    with switch_env(MOJO_HIP_GEMM_TILE128=shape, MOJO_HIP_GEMM_SPLITK="1"):
        small = dense_gemm(x, w, b, trans)
        assert last_launch() == f"gemm128:128x{shape}:{layout}", last_launch()
        again = dense_gemm(x, w, b, trans)
    with switch_env(MOJO_HIP_GEMM_TILE128="0", MOJO_HIP_GEMM_SPLITK="1"):
        large = dense_gemm(x, w, b, trans)
        assert last_launch().startswith("gemm256:") and ":splitk" not in last_launch(), last_launch()
    assert torch.equal(small, again)
    assert torch.equal(small, large)
    if trans:
        want = (x.float() @ w.float()).to(dtype)
        want = want if b is None else (want.float() + b.float()).to(dtype)
    else:
        want = F.linear(x.float(), w.float(), b if b is None else b.float()).to(dtype)
    # (x @ w + b: a last-place difference of the rounded product can land on a smaller binade after the bias is added)
    assert max_ulp_bf16ish(to_cpu(small), to_cpu(want), atol=2e-2) <= (2 if trans and bias else 1)


def test_tile128_default_choice_and_switch():
    """Default: taken where the time model of gemm_api.hip prefers it (more than 128 rows, at most 512 tiles of 128 x 128, and a
    modelled time below the 256 x 256 kernel's best split); not for a launch that fills the chip with 256 x 256 tiles, not for
    not below 129 rows; MOJO_HIP_GEMM_TILE128=0 turns it off."""
    def form(m, k, n, trans=False):
        x = torch.randn(m, k, dtype=torch.bfloat16, device=DEV)
        w = torch.randn((k, n) if trans else (n, k), dtype=torch.bfloat16, device=DEV)
        dense_gemm(x, w, None, trans)
        return last_launch()
    assert form(1024, 4096, 4096) == "gemm128:128x128:NK"
    assert form(512, 1024, 8192) == "gemm128:128x128:NK"
    assert form(2048, 4096, 4096) == "gemm128:128x256:NK"             # 512 tiles of 128 x 128 -> 256 of 128 x 256
    assert form(8192, 1024, 8192).startswith("gemm256:")              # 1024 tiles of 256 x 256
    assert form(1024, 4096, 4096, trans=True) == "gemm128:128x128:KN"
    assert form(128, 4096, 4096) == "gemm128:128x128:NK:splitk"      # 32 tiles over 64 K-tiles: the tiles' own K split
    assert form(256, 8192, 1024) == "gemm128:128x128:NK:splitk"
    assert form(128, 4096, 1024).startswith("gemm_skinny")            # 8 MB of weights: the weight stream
    assert form(64, 4096, 4096) == "gemm128:128x128:NK:splitk"       # 16..64 rows: the tiles with their split where the one-shot model says so
    assert form(8, 4096, 4096).startswith("gemm_skinny")             # under 16 rows: the weight stream
    with switch_env(MOJO_HIP_GEMM_TILE128="0"):
        assert form(1024, 4096, 4096).startswith("gemm256:")


@pytest.mark.parametrize("shape", ["128", "256"])
@pytest.mark.parametrize("layout", ["NK", "KN"])
def test_tile128_row_maps_and_strided_operands(shape, layout):
    """The chunked GEMM + collective pipelines' view: logical row m reads A row (m / rc) * ml + off + m % rc and writes the C
    row of its own map; A and C with row strides wider than K and N."""
    torch.manual_seed(5)
    dtype = torch.bfloat16
    k, n, rc, blocks = 512, 640, 96, 4                               # 384 logical rows: every block's c-th sub-chunk
    a_full = torch.randint(-3, 4, (blocks * 2 * rc, k + 64)).to(dtype).to(DEV)
    w = torch.randint(-3, 4, (n, k)).to(dtype).to(DEV)
    out_full = torch.zeros(blocks * 3 * rc, n + 8, dtype=dtype, device=DEV)
    a_map, c_map = (rc, 2 * rc, rc), (rc, 3 * rc, 2 * rc)            # second sub-chunk of 2 -> third sub-chunk of 3
    eng = HipGemmEngine()
    wk = w.t().contiguous() if layout == "KN" else w
    with switch_env(MOJO_HIP_GEMM_TILE128=shape, MOJO_HIP_GEMM_SPLITK="1"):
        eng(a_full[:, :k], wk, None, layout == "KN", out=out_full[:, :n], rows=blocks * rc, a_map=a_map, c_map=c_map)
        assert last_launch() == f"gemm128:128x{shape}:{layout}", last_launch()
    a_rows = torch.cat([a_full[b * 2 * rc + rc: b * 2 * rc + 2 * rc, :k] for b in range(blocks)])
    want = F.linear(a_rows.float(), w.float()).to(dtype)
    got = torch.cat([out_full[b * 3 * rc + 2 * rc: b * 3 * rc + 3 * rc, :n] for b in range(blocks)])
    assert torch.equal(got, want)
    mask = torch.ones(out_full.shape[0], dtype=torch.bool)
    for b in range(blocks):
        mask[b * 3 * rc + 2 * rc: b * 3 * rc + 3 * rc] = False
    assert not out_full[mask.to(DEV)].any() and not out_full[:, n:].any()      # nothing written outside the mapped rows / N columns


@pytest.mark.parametrize("m,k,n,want_form", [(96, 4096, 14336, "gemm128:128x128:NK:splitk"),  # one row of 112 tiles (x 2 slices) beats the 128-row weight stream
                                             (128, 4096, 28672, "gemm128:128x128:NK"),     # 224 tiles, unsplit
                                             (128, 1024, 33024, "gemm128:128x256:NK"),     # 258 tiles of 128 x 128 -> 128 x 256 tiles
                                             (100, 8192, 1024, "gemm_skinny"),             # 8 tiles over a long K: the weight stream with its K split
                                             (100, 14336, 4096, "gemm128:128x128:NK:splitk"),  # 32 tiles over 224 K-tiles: cut into slices
                                             (64, 14336, 4096, "gemm128:128x128:NK:splitk"),   # a decode batch's down projection
                                             (64, 4096, 28672, "gemm_skinny"),             # 224 tiles would run unsplit: the weight stream
                                             (8, 4096, 14336, "gemm_skinny")])             # under 16 rows: the weight stream
def test_rows_up_to_128_take_the_128_row_tiles_where_the_model_says_so(m, k, n, want_form):
    """At most 128 rows with `[N,K]` weights: one row of 128-row tiles where it beats the weight-streaming kernel
    (gemm_api.hip, gemm_rows128_prefers_tile128); integer-exact either way."""
    torch.manual_seed(m)
    x = torch.randint(-4, 5, (m, k)).to(torch.bfloat16).to(DEV)
    w = torch.randint(-4, 5, (n, k)).to(torch.bfloat16).to(DEV)
    b = torch.randint(-8, 9, (n,)).to(torch.bfloat16).to(DEV)
    got = dense_gemm(x, w, b, False)
    assert last_launch().startswith(want_form), last_launch()
    assert torch.equal(got, F.linear(x.float(), w.float(), b.float()).to(torch.bfloat16))


@pytest.mark.parametrize("m,k,n", [(1, 4096, 4096), (32, 8192, 1024), (64, 1024, 8192), (128, 14336, 4096), (100, 4096, 520)])
@pytest.mark.parametrize("bias", [False, True])
def test_decode_sized_rows_with_kn_weights_split_k(m, k, n, bias):
    """Decode-sized rows with `[K,N]` weights (`x @ w`: the GEMM + collective operators with trans_weight) have no weight-streaming
    kernel; the 256 x 256 kernel now cuts K for them too (it ran one round of a few workgroups over the whole K: 3-11 x the
    vendor library's time).  Small-integer data: split, unsplit (MOJO_HIP_GEMM_SPLITK=1) and the fp32 reference agree to the bit."""
    torch.manual_seed(m + n)
    x = torch.randint(-4, 5, (m, k)).to(torch.bfloat16).to(DEV)
    w = torch.randint(-4, 5, (k, n)).to(torch.bfloat16).to(DEV)
    b = torch.randint(-8, 9, (n,)).to(torch.bfloat16).to(DEV) if bias else None
    want = (x.float() @ w.float()).to(torch.bfloat16)
    want = want if b is None else (want.float() + b.float()).to(torch.bfloat16)
    with switch_env(MOJO_HIP_GEMM_TILE128="0"):
        got = dense_gemm(x, w, b, True)
        assert ":splitk" in last_launch(), last_launch()
        with switch_env(MOJO_HIP_GEMM_SPLITK="1"):
            unsplit = dense_gemm(x, w, b, True)
            assert last_launch().startswith("gemm256:") and ":splitk" not in last_launch(), last_launch()
    assert torch.equal(got, want) and torch.equal(unsplit, want)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,k,n,bias,sk", [
    (256, 8192, 1024, True, 0),           # the model's own split (16 tiles over 128 K-tiles)
    (200, 4096, 4096, False, 0),
    (512, 7168, 2048, True, 0),
    (129, 192, 132, True, 3),             # one K-tile per slice, ragged M and N (N % 128 = 4)
    (300, 1024, 520, True, 5),            # 16 K-tiles in 5 uneven slices
    (1000, 512, 4096, False, 2),
])
@pytest.mark.parametrize("layout", ["NK", "KN"])
def test_tile128_split_k_integer_data_is_exact(dtype, m, k, n, bias, sk, layout):
    """Few 128 x 128 tiles over a long K: K is cut, the slices' fp32 accumulators go to slabs and the finalize launch sums them in
    slice order, rounds and adds the bias as the unsplit epilogue does.  Small-integer data: exact whatever the split."""
    torch.manual_seed(m + n)
    if layout == "KN":
        n = (n + 7) // 8 * 8
    x = torch.randint(-4, 5, (m, k)).to(dtype).to(DEV)
    w = torch.randint(-4, 5, (n, k)).to(dtype).to(DEV)
    b = torch.randint(-8, 9, (n,)).to(dtype).to(DEV) if bias else None
    if layout == "NK":
        want = F.linear(x.float(), w.float(), None if b is None else b.float()).to(dtype)
    else:
        w = w.t().contiguous()
        want = (x.float() @ w.float()).to(dtype)
        want = want if b is None else (want.float() + b.float()).to(dtype)
    env = dict(MOJO_HIP_GEMM_TILE128="1", MOJO_HIP_GEMM_SPLITK=str(sk)) if sk else {}
    with switch_env(**env):
        got = dense_gemm(x, w, b, layout == "KN")
        assert last_launch() == f"gemm128:128x128:{layout}:splitk", last_launch()
    assert torch.equal(got, want)


def test_tile128_split_k_row_maps_and_random_data():
    """Split + row maps on both sides (the slabs are indexed by logical row; the finalize applies the C map), and random data
    within one unit in the last place of the fp32 reference."""
    torch.manual_seed(9)
    dtype = torch.bfloat16
    k, n, rc, blocks = 2048, 640, 96, 3
    a_full = torch.randn(blocks * 2 * rc, k + 64, device=DEV).to(dtype)
    w = (torch.randn(n, k, device=DEV) * 0.05).to(dtype)
    b = torch.randn(n, device=DEV).to(dtype)
    out_full = torch.zeros(blocks * 3 * rc, n + 8, dtype=dtype, device=DEV)
    a_map, c_map = (rc, 2 * rc, rc), (rc, 3 * rc, 2 * rc)
    eng = HipGemmEngine()
    with switch_env(MOJO_HIP_GEMM_TILE128="1", MOJO_HIP_GEMM_SPLITK="4"):
        eng(a_full[:, :k], w, b, False, out=out_full[:, :n], rows=blocks * rc, a_map=a_map, c_map=c_map)
        assert last_launch() == "gemm128:128x128:NK:splitk", last_launch()
    a_rows = torch.cat([a_full[i * 2 * rc + rc: i * 2 * rc + 2 * rc, :k] for i in range(blocks)])
    want = F.linear(a_rows.float(), w.float(), b.float()).to(dtype)
    got = torch.cat([out_full[i * 3 * rc + 2 * rc: i * 3 * rc + 3 * rc, :n] for i in range(blocks)])
    assert max_ulp_bf16ish(to_cpu(got), to_cpu(want), atol=2e-2) <= 1
    mask = torch.ones(out_full.shape[0], dtype=torch.bool)
    for i in range(blocks):
        mask[i * 3 * rc + 2 * rc: i * 3 * rc + 3 * rc] = False
    assert not out_full[mask.to(DEV)].any() and not out_full[:, n:].any()
