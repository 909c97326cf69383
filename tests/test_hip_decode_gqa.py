"""GPU parity of MojoPagedDecodeGQA through the C ABI.

Tolerance: atol = rtol = 2e-2, the reference's own bound for this op
(mojo_opset/tests/accuracy/operators/test_attention.py:137-138)."""
import math

import pytest
import torch

from conftest import load_golden
from hip_utils import (DEV, assert_close_tree, hip_cls, last_launch, run_hip_case, skip_unless_experiments_build, switch_env, to_cpu,
                       torch_cls)

pytestmark = pytest.mark.gpu
ATOL = RTOL = 2e-2


def make_decode_inputs(batch, hq, hkv, d, max_len, page, dtype=torch.bfloat16, seed=0, lens=None):
    """Same construction as the reference's generator (test_attention.py:33-83): random lengths,
    pools with 10 spare pages, a shuffled table padded with -1."""
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(batch, hq, d, generator=g).to(dtype)
    if lens is None:
        if max_len > 0:
            lens = torch.randint(0, max_len, (batch,), generator=g, dtype=torch.int32).clamp(min=1)
        else:
            lens = torch.randperm(batch, generator=g, dtype=torch.int32)
    else:
        lens = torch.tensor(lens, dtype=torch.int32)
    need = (lens + page - 1) // page
    width = max(int(need.max()), 1)
    total = max(int(need.sum()), 1) + 10
    k = torch.randn(total, hkv, page, d, generator=g).to(dtype)
    v = torch.randn(total, hkv, page, d, generator=g).to(dtype)
    table = torch.full((batch, width), -1, dtype=torch.int32)
    free = torch.randperm(total, generator=g, dtype=torch.int32)
    at = 0
    for b in range(batch):
        n = int(need[b])
        table[b, :n] = free[at: at + n]
        at += n
    return q, k, v, lens, table


@pytest.mark.parametrize("case", [pytest.param(c, id=f"decode-{i}") for i, c in enumerate(load_golden("paged_decode_gqa"))])
def test_decode_vectors(case):
    assert_close_tree(to_cpu(run_hip_case(case)), case["out"], ATOL, RTOL)


@pytest.mark.parametrize("cfg", [
    (8, 16, 4, 128, 1024, 32), (8, 16, 4, 96, 1024, 128), (8, 8, 1, 128, 8192, 1024),
    (8, 8, 1, 128, 2048, 1024), (8, 8, 1, 128, 0, 1024),
    (64, 32, 8, 128, 1024, 16),                     # BASELINE shape at a length the CPU oracle handles
], ids=["M_BF16", "M_BF16_PADDIM", "M_BF16_LONG", "M_BF16_BIGPAGE", "M_BF16_PADSEQ", "LLAMA3_8B_ctx1k"])
@pytest.mark.parametrize("layout", ["ABAB", "AABB"])
def test_decode_reference_space(cfg, layout):
    batch, hq, hkv, d, max_len, page = cfg
    q, k, v, lens, table = make_decode_inputs(batch, hq, hkv, d, max_len, page, seed=hash(cfg) % 1000)
    op = hip_cls("MojoPagedDecodeGQA")(is_causal=True, gqa_layout=layout)
    ref = torch_cls("MojoPagedDecodeGQA")(is_causal=True, gqa_layout=layout)
    dev = [t.to(DEV) for t in (q, k, v, lens, table)]
    op.forward_diff_with(ref, *dev, softmax_scale=1.0 / math.sqrt(d), max_total_seq_len=int(lens.max()),
                         atol=ATOL, rtol=RTOL, ref_device="cpu")
    # without the host hint the split count comes from the table width: same numbers
    a = op(*dev)
    b = op(*dev, max_total_seq_len=int(lens.max()))
    torch.testing.assert_close(a.float(), b.float(), atol=2e-3, rtol=2e-3)


def test_decode_zero_length_rows_are_zero_and_fp16():
    q, k, v, lens, table = make_decode_inputs(6, 8, 2, 128, 0, 16, dtype=torch.float16, lens=[0, 5, 0, 300, 1, 0])
    op = hip_cls("MojoPagedDecodeGQA")()
    out = op(*[t.to(DEV) for t in (q, k, v, lens, table)])
    assert torch.count_nonzero(out[[0, 2, 5]]) == 0
    want = torch_cls("MojoPagedDecodeGQA")()(q, k, v, lens, table)
    assert_close_tree(to_cpu(out), want, ATOL, RTOL)


def test_decode_contract_errors():
    q, k, v, lens, table = [t.to(DEV) for t in make_decode_inputs(2, 8, 2, 128, 64, 16)]
    op = hip_cls("MojoPagedDecodeGQA")()
    with pytest.raises(AssertionError):
        op(q, k, v, lens.long(), table)
    with pytest.raises(AssertionError):
        op(q, k, v, lens, table.long())
    with pytest.raises(NotImplementedError):
        hip_cls("MojoPagedDecodeGQA")(is_causal=False)(q, k, v, lens, table, mask=torch.ones(8, 8, dtype=torch.bool, device=DEV))
    with pytest.raises(ValueError):
        hip_cls("MojoPagedDecodeGQA")(gqa_layout="BBAA")


def test_decode_invalid_first_page_raises_when_validation_is_on(monkeypatch):
    q, k, v, lens, table = [t.to(DEV) for t in make_decode_inputs(2, 8, 2, 128, 64, 16)]
    table[1, 0] = -1
    monkeypatch.setenv("MOJO_HIP_VALIDATE", "1")
    with pytest.raises(ValueError):
        hip_cls("MojoPagedDecodeGQA")()(q, k, v, lens, table)


def test_decode_full_size_properties():
    """BASELINE config 2 at full size (B=64, 32q/8kv, D=128, page=16, ctx=4096): size-independent
    properties + the oracle on a sample of sequences."""
    B, hq, hkv, d, page, ctx = 64, 32, 8, 128, 16, 4096
    g = torch.Generator().manual_seed(20260716)
    lens = torch.randint(ctx // 2, ctx + 1, (B,), generator=g, dtype=torch.int32)
    lens[0] = ctx
    q, k, v, lens, table = make_decode_inputs(B, hq, hkv, d, 0, page, lens=lens.tolist(), seed=7)
    dev = [t.to(DEV) for t in (q, k, v, lens, table)]
    op = hip_cls("MojoPagedDecodeGQA")()
    out = op(*dev, max_total_seq_len=ctx)
    assert torch.isfinite(out.float()).all()
    # (1) page placement must not matter: relabel the physical pages, permute the pools accordingly
    perm = torch.randperm(k.shape[0], generator=g)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(perm.numel())
    k2, v2 = k[perm], v[perm]                                   # new page p holds old page perm[p]
    table2 = torch.where(table >= 0, inv[table.clamp(min=0).long()].to(torch.int32), table)
    out2 = op(q.to(DEV), k2.to(DEV), v2.to(DEV), dev[3], table2.to(DEV), max_total_seq_len=ctx)
    assert torch.equal(out, out2)
    # (2) softmax weights are a convex combination: constant V rows come back unchanged
    vconst = torch.full_like(v, 0.5)
    outc = op(dev[0], dev[1], vconst.to(DEV), dev[3], dev[4], max_total_seq_len=ctx)
    torch.testing.assert_close(outc.float(), torch.full_like(outc, 0.5).float(), atol=4e-3, rtol=0)
    # (3) oracle on three whole sequences (longest, a ragged one, the last)
    ref = torch_cls("MojoPagedDecodeGQA")()
    for b in (0, 17, B - 1):
        want = ref(q[b: b + 1], k, v, lens[b: b + 1], table[b: b + 1])
        assert_close_tree(to_cpu(out[b: b + 1]), want, ATOL, RTOL)


def test_decode_length_above_the_hint_is_truncated_not_out_of_bounds():
    """A `total_seq_lens` entry above the caller's `max_total_seq_len` hint: the launch (grid, LDS image, partials) was
    sized for the hint, so the kernels truncate that row to the launch's capacity instead of indexing past it.  Both
    the in-LDS merge (<= 8 chunks) and the split + merge-kernel route (forced by a small chunk) are exercised; with
    MOJO_HIP_VALIDATE=1 the shim raises instead."""
    lens = [700, 64, 3000, 1]
    q, k, v, lens_t, table = make_decode_inputs(4, 8, 2, 128, 0, 16, lens=lens, seed=3)
    dev = [t.to(DEV) for t in (q, k, v, lens_t, table)]
    op = hip_cls("MojoPagedDecodeGQA")()
    ref = torch_cls("MojoPagedDecodeGQA")()
    forms = []
    for chunk_env in (None, "64"):
        with switch_env(MOJO_HIP_DECODE_CHUNK=chunk_env):
            hint = 1024                                              # row 2 (3000 tokens) is above it
            out = op(*dev, max_total_seq_len=hint)
            torch.cuda.synchronize()
            forms.append(last_launch())
        assert torch.isfinite(out.float()).all()
        # rows inside the hint are untouched by the clamp
        want = ref(q, k, v, lens_t, table)
        for b in (0, 1, 3):
            assert_close_tree(to_cpu(out[b: b + 1]), want[b: b + 1], ATOL, RTOL)
        # the long row equals attention over a prefix of its context: the launch's capacity (>= the hint, whole chunks)
        got = to_cpu(out[2:3]).float()
        errs = {}
        for cap in sorted({hint, *(c for c in range(hint, 3001, 16))}):
            w = ref(q[2:3], k, v, torch.tensor([cap], dtype=torch.int32), table[2:3]).float()
            errs[cap] = float((got - w).abs().max())
            if errs[cap] <= ATOL:
                break
        assert min(errs.values()) <= ATOL, f"row above the hint matches no prefix: {min(errs.values())}"
    assert "split+merge" not in forms[0] and "merge" in forms[1], forms          # both routes really ran


def test_decode_hint_violation_raises_when_validation_is_on(monkeypatch):
    q, k, v, lens, table = [t.to(DEV) for t in make_decode_inputs(2, 8, 2, 128, 0, 16, lens=[100, 900])]
    monkeypatch.setenv("MOJO_HIP_VALIDATE", "1")
    with pytest.raises(ValueError):
        hip_cls("MojoPagedDecodeGQA")()(q, k, v, lens, table, max_total_seq_len=512)


@pytest.mark.parametrize("layout", ["ABAB", "AABB"])
def test_decode_padded_rows_under_graph_replay(layout):
    """The reference's `test_paged_decode_gqa_with_graph` (tests/accuracy/operators/test_attention.py:218-353): static
    buffers are mutated in place between replays, padded rows get seq_len = 0 and block_tables = -1, and their output
    rows must be LEFT UNCHANGED by the replay; eagerly the same rows are zeros (attention.py:184-185)."""
    B, hq, hkv, d, page, max_len = 8, 16, 4, 128, 32, 1024
    q, k, v, lens, table = make_decode_inputs(B, hq, hkv, d, max_len, page, seed=11)
    width = (max_len + page - 1) // page
    tbl = torch.full((B, width), -1, dtype=torch.int32)
    tbl[:, : table.shape[1]] = table
    pool = B * width + 10                                      # static pools hold any batch the generator can make
    k_pool, v_pool = torch.zeros(pool, hkv, page, d, dtype=k.dtype), torch.zeros(pool, hkv, page, d, dtype=v.dtype)
    k_pool[: k.shape[0]], v_pool[: v.shape[0]] = k, v
    sq, sk, sv, sl, st = [t.to(DEV) for t in (q, k_pool, v_pool, lens, tbl)]
    op = hip_cls("MojoPagedDecodeGQA")(is_causal=True, gqa_layout=layout)
    ref = torch_cls("MojoPagedDecodeGQA")(is_causal=True, gqa_layout=layout)
    op(sq, sk, sv, sl, st, max_total_seq_len=max_len)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = op(sq, sk, sv, sl, st, max_total_seq_len=max_len)
    graph.replay()
    torch.cuda.synchronize()
    assert_close_tree(to_cpu(out), ref(q, k, v, lens, tbl), ATOL, RTOL)
    g = torch.Generator().manual_seed(5)
    for step in range(4):
        cur_b = int(torch.randint(1, B, (), generator=g))
        cq, ck, cv, cl, ct = make_decode_inputs(cur_b, hq, hkv, d, max_len, page, seed=100 + step)
        sk[: ck.shape[0]].copy_(ck.to(DEV))
        sv[: cv.shape[0]].copy_(cv.to(DEV))
        sq[:cur_b].copy_(cq.to(DEV))
        sl[:cur_b].copy_(cl.to(DEV))
        sl[cur_b:] = 0
        st.fill_(-1)
        st[:cur_b, : ct.shape[1]].copy_(ct.to(DEV))
        keep = out[cur_b:].clone()
        graph.replay()
        torch.cuda.synchronize()
        assert_close_tree(to_cpu(out[:cur_b]), ref(cq, ck, cv, cl, ct), ATOL, RTOL)
        assert torch.equal(out[cur_b:], keep), "padded rows were modified by the replay"
        # eager call on the same buffers: padded rows are zeros (golden semantics); explicit flag overrides both ways
        eager = op(sq, sk, sv, sl, st, max_total_seq_len=max_len)
        assert torch.count_nonzero(eager[cur_b:]) == 0
        torch.testing.assert_close(eager[:cur_b].float(), out[:cur_b].float(), atol=0, rtol=0)
        forced = op(sq, sk, sv, sl, st, max_total_seq_len=max_len, leave_empty_rows=False)
        assert torch.count_nonzero(forced[cur_b:]) == 0


@pytest.mark.parametrize("batch", [64, 63, 53])
@pytest.mark.parametrize("layout", ["AABB", "ABAB"])
def test_decode_paired_workgroups_on_ragged_batches(batch, layout, monkeypatch):
    """Llama-3-8B head shape with 52..64 sequences: four waves per (sequence, kv-head) fill the chip once, and the kernel
    then pairs the sequences by length rank (longest with shortest, ...) in 8-wave workgroups that deal their waves by
    length.  Ragged lengths incl. empty rows, an odd batch (the middle sequence has no partner), ties; against the oracle,
    and against the unpaired kernel (MOJO_HIP_DECODE_PAIR=0) on the same inputs."""
    g = torch.Generator().manual_seed(batch)
    lens = torch.randint(1, 700, (batch,), generator=g).tolist()
    lens[3], lens[10], lens[11], lens[12] = 0, 699, 699, 1           # an empty row, a tie, a one-token row
    q, k, v, lens_t, table = make_decode_inputs(batch, 32, 8, 128, 0, 16, lens=lens, seed=batch + 1)
    dev = [t.to(DEV) for t in (q, k, v, lens_t, table)]
    op = hip_cls("MojoPagedDecodeGQA")(is_causal=True, gqa_layout=layout)
    ref = torch_cls("MojoPagedDecodeGQA")(is_causal=True, gqa_layout=layout)
    want = ref(q, k, v, lens_t, table)
    got = op(*dev, max_total_seq_len=700)
    assert last_launch().startswith("decode_mfma:paired"), last_launch()
    assert_close_tree(to_cpu(got), want, ATOL, RTOL)
    assert torch.count_nonzero(got[3]) == 0
    monkeypatch.setenv("MOJO_HIP_DECODE_PAIR", "0")                  # (the fixture makes the library re-read its switches)
    plain = op(*dev, max_total_seq_len=700)
    assert last_launch().startswith("decode_mfma:fused"), last_launch()               # the OTHER form really ran
    assert_close_tree(to_cpu(plain), want, ATOL, RTOL)
    torch.testing.assert_close(got.float(), plain.float(), atol=4e-3, rtol=4e-3)     # different chunk boundaries, same sums
    monkeypatch.delenv("MOJO_HIP_DECODE_PAIR")
    # launch-to-launch determinism
    assert torch.equal(op(*dev, max_total_seq_len=700), got)
    assert last_launch().startswith("decode_mfma:paired"), last_launch()


@pytest.mark.parametrize("layout", ["AABB", "ABAB"])
@pytest.mark.parametrize("batch,hq,hkv,d,lens", [(5, 64, 8, 128, [1, 700, 64, 333, 2049]), (3, 8, 1, 128, [4096, 17, 900]),
                                                  (2, 32, 4, 96, [150, 1000]), (9, 16, 2, 64, [33] * 9)])
@pytest.mark.parametrize("halves", ["0", "1"], ids=["g8", "halves"])
def test_decode_groups_of_eight_query_heads(batch, hq, hkv, d, lens, layout, halves, monkeypatch):
    """Hq / Hkv = 8 (Llama-3-70B's 64 / 8, or its per-rank 8 / 1 under TP 8): the 8-head instance of the kernel (heads in
    blocks of four over a two-tile ring, query slices in LDS: K/V read once; the default) and the round-2 form that runs the
    4-head kernel on the two halves of every kv head (MOJO_HIP_DECODE_G8_HALVES=1, `hshift`).  Against the oracle in both
    head layouts, ragged lengths, the fused, paired and split + merge forms (short and long rows)."""
    if halves == "1":
        skip_unless_experiments_build()                              # (the halves form is compiled into experiments builds only)
    monkeypatch.setenv("MOJO_HIP_DECODE_G8_HALVES", halves)
    monkeypatch.setenv("MOJO_HIP_DECODE_MFMA", "0")                  # both forms belong to the vector-unit kernel
    torch.manual_seed(hq + d)
    page = 16
    need = [(n + page - 1) // page for n in lens]
    total = sum(need) + 3
    k = torch.randn(total, hkv, page, d).to(torch.bfloat16)
    v = torch.randn(total, hkv, page, d).to(torch.bfloat16)
    perm = torch.randperm(total, dtype=torch.int32)
    table = torch.full((batch, max(need)), -1, dtype=torch.int32)
    at = 0
    for i, n in enumerate(need):
        table[i, :n] = perm[at: at + n]
        at += n
    q = torch.randn(batch, hq, d).to(torch.bfloat16)
    sl = torch.tensor(lens, dtype=torch.int32)
    want = torch_cls("MojoPagedDecodeGQA")(is_causal=True, gqa_layout=layout)(q, k, v, sl, table)
    op = hip_cls("MojoPagedDecodeGQA")(is_causal=True, gqa_layout=layout)
    dev = (q.to(DEV), k.to(DEV), v.to(DEV), sl.to(DEV), table.to(DEV))
    got = to_cpu(op(*dev))
    assert (got.float() - want.float()).abs().max().item() <= 2e-2
    assert torch.equal(to_cpu(op(*dev)), got)                        # launch-to-launch determinism


# ---- the matrix-core kernel (csrc/paged_decode_mfma.h) ----------------------------------------------------------------------
@pytest.mark.parametrize("cfg", [
    # batch, hq, hkv, d, page, lens, dtype
    (5, 64, 8, 128, 16, [1, 700, 64, 333, 2049], torch.bfloat16),          # G 8: fused form, ragged, a one-token row
    (64, 32, 8, 128, 16, None, torch.bfloat16),                             # the headline geometry: paired form (G 4)
    (6, 16, 1, 128, 32, [0, 5, 0, 300, 4097, 31], torch.float16),          # G 16, empty rows, fp16, page 32, split + merge
    (3, 8, 1, 128, 64, [4096, 17, 900], torch.bfloat16),                    # Llama-3-70B per-rank shape under TP 8
    (7, 12, 4, 64, 16, [33, 64, 65, 1, 15, 16, 17], torch.bfloat16),        # G 3 (no vector-unit instance), head_dim 64
    (9, 10, 2, 128, 128, [500] * 9, torch.bfloat16),                         # G 5, large pages
    (2, 8, 8, 64, 16, [129, 1000], torch.float16),                           # G 1 (MHA), head_dim 64
], ids=["G8_ragged", "G4_paired", "G16_fp16_split", "G8_tp8", "G3_d64", "G5_page128", "G1_d64"])
@pytest.mark.parametrize("layout", ["AABB", "ABAB"])
def test_decode_matrix_core_kernel(cfg, layout, monkeypatch):
    """MOJO_HIP_DECODE_MFMA=1: QK^T and PV on the matrix cores wherever the kernel applies (pages of a multiple of 16 tokens,
    head_dim 64 / 128, groups of <= 16 query heads) — against the oracle, bit-stable from launch to launch, equal to the
    vector-unit kernel within accumulation-order noise where both exist, page relabelling changes no bit."""
    batch, hq, hkv, d, page, lens, dtype = cfg
    monkeypatch.setenv("MOJO_HIP_DECODE_MFMA", "1")
    q, k, v, lens_t, table = make_decode_inputs(batch, hq, hkv, d, 1024, page, dtype=dtype, seed=hq * 7 + d, lens=lens)
    op = hip_cls("MojoPagedDecodeGQA")(is_causal=True, gqa_layout=layout)
    want = torch_cls("MojoPagedDecodeGQA")(is_causal=True, gqa_layout=layout)(q, k, v, lens_t, table)
    dev = [t.to(DEV) for t in (q, k, v, lens_t, table)]
    got = op(*dev)
    assert last_launch().startswith("decode_mfma:"), last_launch()
    assert_close_tree(to_cpu(got), want, ATOL, RTOL)
    assert torch.equal(op(*dev), got)
    if (lens_t <= 0).any():
        assert torch.count_nonzero(got[(lens_t <= 0).to(DEV)]) == 0
    g = torch.Generator().manual_seed(3)
    perm = torch.randperm(k.shape[0], generator=g)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(perm.numel())
    table2 = torch.where(table >= 0, inv[table.clamp(min=0).long()].to(torch.int32), table)
    assert torch.equal(op(q.to(DEV), k[perm].to(DEV), v[perm].to(DEV), dev[3], table2.to(DEV)), got)
    if hq // hkv in (1, 2, 4, 8):
        monkeypatch.setenv("MOJO_HIP_DECODE_MFMA", "0")
        plain = op(*dev)
        assert last_launch().startswith("decode_valu:"), last_launch()                # the vector-unit kernel really ran
        assert_close_tree(to_cpu(plain), want, ATOL, RTOL)
        torch.testing.assert_close(got.float(), plain.float(), atol=8e-3, rtol=8e-3)


def test_decode_matrix_core_kernel_holes_and_replay(monkeypatch):
    """Pages behind the first negative id read as zero K/V (the golden's `break`), padded rows stay untouched under
    `leave_empty_rows`, lengths beyond the caller's bound are truncated — on the matrix-core kernel."""
    monkeypatch.setenv("MOJO_HIP_DECODE_MFMA", "1")
    q, k, v, lens, table = make_decode_inputs(4, 16, 2, 128, 0, 16, lens=[700, 0, 130, 48], seed=11)
    table[0, 9] = -1                                                        # a hole inside row 0: tokens 144.. read as zeros
    ref = torch_cls("MojoPagedDecodeGQA")()
    want = ref(q, k, v, lens, table)
    op = hip_cls("MojoPagedDecodeGQA")()
    dev = [t.to(DEV) for t in (q, k, v, lens, table)]
    got = op(*dev)
    assert_close_tree(to_cpu(got), want, ATOL, RTOL)
    again = op(*dev, leave_empty_rows=True)
    assert torch.equal(again[[0, 2, 3]], got[[0, 2, 3]])
    short = op(*dev, max_total_seq_len=128)
    want_short = ref(q, k, v, lens.clamp(max=128), table)
    assert_close_tree(to_cpu(short), want_short, ATOL, RTOL)


@pytest.mark.parametrize("fuse", ["1", "0"], ids=["grouped", "no_fuse"])
@pytest.mark.parametrize("cfg", [(8, 32, 8, 128, 8192), (8, 8, 1, 128, 8192), (3, 64, 8, 128, 5000)], ids=["8B_B8", "70B_tp8", "70B_B3"])
def test_decode_small_grids_grouped_form_and_without_fusion(cfg, fuse, monkeypatch):
    """Few (sequence, kv-head) rows, many chunks each: the GROUPED form (eight-wave workgroups merge eight chunks in LDS and
    leave one partial; the merge kernel reads n_chunks / 8 partials per row).  ADVICE r4: with MOJO_HIP_DECODE_FUSE=0 the
    launch fell through to one partial per chunk while the merge still divided the chunk count by eight — rows of 2..8
    chunks were never written, longer rows merged an eighth of their partials.  Both settings against the oracle, ragged
    lengths incl. rows of one chunk, of 2..8 chunks and of more; and the two forms agree."""
    batch, hq, hkv, d, ctx = cfg
    g = torch.Generator().manual_seed(ctx + batch)
    lens = torch.randint(ctx // 2, ctx, (batch,), generator=g).tolist()
    lens[0], lens[1] = 100, 700                                        # one chunk; a few chunks
    lens[-1] = ctx
    q, k, v, lens_t, table = make_decode_inputs(batch, hq, hkv, d, 0, 16, lens=lens, seed=7)
    dev = [t.to(DEV) for t in (q, k, v, lens_t, table)]
    op = hip_cls("MojoPagedDecodeGQA")()
    want = torch_cls("MojoPagedDecodeGQA")()(q, k, v, lens_t, table)
    monkeypatch.setenv("MOJO_HIP_DECODE_FUSE", fuse)
    out = torch.full((batch, hq, d), float("nan"), dtype=torch.bfloat16, device=DEV)
    got = op(*dev, max_total_seq_len=ctx)
    form = last_launch()
    assert form.startswith("decode_mfma:grouped+merge" if fuse == "1" else "decode_mfma:split+merge"), form
    assert torch.isfinite(got.float()).all()
    assert_close_tree(to_cpu(got), want, ATOL, RTOL)
    assert torch.equal(op(*dev, max_total_seq_len=ctx), got)
    del out
