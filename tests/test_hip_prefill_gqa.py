"""GPU parity of MojoPagedPrefillGQA through the C ABI.  Tolerance atol = rtol = 2e-2 — the reference's
strict bound for this op (mojo_opset/tests/accuracy/operators/test_attention.py:577-582)."""
import math

import pytest
import torch

from conftest import load_golden
from hip_utils import DEV, assert_close_tree, hip_cls, last_launch, run_hip_case, skip_unless_experiments_build, to_cpu, torch_cls

pytestmark = pytest.mark.gpu
ATOL = RTOL = 2e-2


def cu(lens):
    return torch.tensor([0] + list(torch.tensor(lens).cumsum(0).tolist()), dtype=torch.int32)


def make_prefill_inputs(q_lens, cached, hq, hkv, d, page, dtype=torch.bfloat16, seed=0, pad_tokens=0):
    g = torch.Generator().manual_seed(seed)
    kv_lens = [a + b for a, b in zip(q_lens, cached)]
    need = [(n + page - 1) // page for n in kv_lens]
    total = max(sum(need), 1) + 10
    k = torch.randn(total, hkv, page, d, generator=g).to(dtype)
    v = torch.randn(total, hkv, page, d, generator=g).to(dtype)
    table = torch.full((len(q_lens), max(max(need), 1)), -1, dtype=torch.int32)
    free = torch.randperm(total, generator=g, dtype=torch.int32)
    at = 0
    for b, n in enumerate(need):
        table[b, :n] = free[at: at + n]
        at += n
    q = torch.randn(sum(q_lens) + pad_tokens, hq, d, generator=g).to(dtype)
    return q, k, v, cu(q_lens), table, (cu(kv_lens) if any(cached) else None), kv_lens


@pytest.mark.parametrize("case", [pytest.param(c, id=f"prefill-{i}") for i, c in enumerate(load_golden("paged_prefill_gqa"))])
def test_prefill_vectors(case):
    assert_close_tree(to_cpu(run_hip_case(case)), case["out"], ATOL, RTOL)


@pytest.mark.parametrize("cfg", [
    # (B, Hq, Hkv, D, max_q, max_cached, page) of the reference (test_attention.py:433-461), lengths drawn below
    (2, 16, 4, 128, 1024, 0, 32), (2, 16, 4, 96, 1024, 0, 128), (2, 8, 1, 128, 2048, 4096, 128),
    (2, 8, 1, 128, 1024, 2048, 1024), (2, 8, 1, 128, 0, 0, 1024), (5, 4, 2, 128, 77, 50, 16),
    (3, 32, 8, 128, 512, 300, 16),
], ids=["M_BF16", "PADDIM", "LONG_CACHED", "BIGPAGE", "EMPTY", "BUCKET_PAGE16", "MIXTRAL_SHAPE"])
@pytest.mark.parametrize("layout", ["ABAB", "AABB"])
def test_prefill_reference_space(cfg, layout):
    batch, hq, hkv, d, max_q, max_c, page = cfg
    g = torch.Generator().manual_seed(batch * 1000 + max_q)
    q_lens = [int(x) for x in (torch.randint(max_q // 2, max_q + 1, (batch,), generator=g) if max_q else torch.zeros(batch))]
    cached = [int(x) for x in (torch.randint(0, max_c + 1, (batch,), generator=g) if max_c else torch.zeros(batch))]
    q, k, v, cu_q, table, cu_kv, kv_lens = make_prefill_inputs(q_lens, cached, hq, hkv, d, page, seed=batch)
    op = hip_cls("MojoPagedPrefillGQA")(is_causal=True, gqa_layout=layout)
    ref = torch_cls("MojoPagedPrefillGQA")(is_causal=True, gqa_layout=layout)
    kw = dict(softmax_scale=1.0 / math.sqrt(d), max_q_len=max(q_lens + [0]), max_total_seq_len=max(kv_lens + [0]))
    if cu_kv is not None:
        kw["cu_total_seq_lens"] = cu_kv
    want = ref(q, k, v, cu_q, table, **kw)
    dkw = {k_: (v_.to(DEV) if isinstance(v_, torch.Tensor) else v_) for k_, v_ in kw.items()}
    got = op(q.to(DEV), k.to(DEV), v.to(DEV), cu_q.to(DEV), table.to(DEV), **dkw)
    assert_close_tree(to_cpu(got), want, ATOL, RTOL)
    # without host hints the grid bound falls back to the token count: same numbers — the same BITS unless the hints
    # change whether (or how finely) the launch is cut along the keys, which changes the association of the sums
    from mojo_opset_amd.backends.hip import lib as L
    ws = lambda hq_, hk_: L.load().mojo_hip_paged_prefill_gqa_workspace_bytes(q.shape[0], batch, hq, hkv, d, page, table.shape[1], hq_, hk_)  # noqa: E731
    same_plan = ws(dkw["max_q_len"], dkw["max_total_seq_len"]) == 0 and ws(0, 0) == 0
    dkw.pop("max_q_len"), dkw.pop("max_total_seq_len")
    got2 = op(q.to(DEV), k.to(DEV), v.to(DEV), cu_q.to(DEV), table.to(DEV), **dkw)
    if same_plan:
        assert torch.equal(got, got2)
    else:
        torch.testing.assert_close(got.float(), got2.float(), atol=8e-3, rtol=8e-3)


def test_prefill_padding_tokens_and_empty_sequences_are_zero():
    q, k, v, cu_q, table, cu_kv, _ = make_prefill_inputs([0, 40, 0, 9], [5, 0, 0, 100], 8, 2, 128, 16, pad_tokens=7)
    op = hip_cls("MojoPagedPrefillGQA")()
    got = to_cpu(op(q.to(DEV), k.to(DEV), v.to(DEV), cu_q.to(DEV), table.to(DEV), cu_total_seq_lens=cu_kv.to(DEV)))
    want = torch_cls("MojoPagedPrefillGQA")()(q, k, v, cu_q, table, cu_total_seq_lens=cu_kv)
    assert torch.count_nonzero(got[49:]) == 0
    assert_close_tree(got, want, ATOL, RTOL)


def test_prefill_sequence_without_keys_reads_zero():
    """q_len > 0 with kv_len == 0 (golden: `continue`, rows stay zero) — the workgroups of that sequence write the zeros."""
    q, k, v, cu_q, table, cu_kv, _ = make_prefill_inputs([37, 20], [0, 30], 8, 2, 128, 16)
    cu_kv = torch.tensor([0, 0, 50], dtype=torch.int32)
    op = hip_cls("MojoPagedPrefillGQA")()
    got = to_cpu(op(q.to(DEV), k.to(DEV), v.to(DEV), cu_q.to(DEV), table.to(DEV), cu_total_seq_lens=cu_kv.to(DEV)))
    want = torch_cls("MojoPagedPrefillGQA")()(q, k, v, cu_q, table, cu_total_seq_lens=cu_kv)
    assert torch.count_nonzero(got[:37]) == 0 and torch.count_nonzero(want[:37]) == 0
    assert_close_tree(got, want, ATOL, RTOL)


def test_prefill_fp16_and_contract():
    q, k, v, cu_q, table, cu_kv, _ = make_prefill_inputs([33, 70], [12, 0], 8, 2, 64, 16, dtype=torch.float16)
    op = hip_cls("MojoPagedPrefillGQA")()
    got = op(q.to(DEV), k.to(DEV), v.to(DEV), cu_q.to(DEV), table.to(DEV), cu_total_seq_lens=cu_kv.to(DEV))
    want = torch_cls("MojoPagedPrefillGQA")()(q, k, v, cu_q, table, cu_total_seq_lens=cu_kv)
    assert_close_tree(to_cpu(got), want, ATOL, RTOL)
    with pytest.raises(AssertionError):
        op(q.to(DEV), k.to(DEV), v.to(DEV), cu_q.long().to(DEV), table.to(DEV))
    with pytest.raises(NotImplementedError):
        hip_cls("MojoPagedPrefillGQA")(is_causal=False)(q.to(DEV), k.to(DEV), v.to(DEV), cu_q.to(DEV), table.to(DEV))


def test_prefill_full_size_properties():
    """BASELINE config 3a at full size: 4 x 2048 new tokens over 2048 cached, Hq=32, Hkv=8, D=128, page=16."""
    q_lens, cached = [2048] * 4, [2048] * 4
    q, k, v, cu_q, table, cu_kv, _ = make_prefill_inputs(q_lens, cached, 32, 8, 128, 16, seed=11)
    op = hip_cls("MojoPagedPrefillGQA")()
    dev = [t.to(DEV) for t in (q, k, v, cu_q, table, cu_kv)]
    out = op(*dev[:5], cu_total_seq_lens=dev[5], max_q_len=2048, max_total_seq_len=4096)
    assert torch.isfinite(out.float()).all()
    # (1) softmax rows are convex weights: constant V comes back unchanged
    outc = op(dev[0], dev[1], torch.full_like(dev[2], 0.25), dev[3], dev[4], cu_total_seq_lens=dev[5])
    torch.testing.assert_close(outc.float(), torch.full_like(outc, 0.25).float(), atol=2e-3, rtol=0)
    # (2) causality: corrupting the last kv position changes only the last query row of each sequence
    k2, v2 = dev[1].clone(), dev[2].clone()
    for b in range(4):
        pid = int(table[b, (4096 - 1) // 16])
        k2[pid, :, 15] += 3.0
        v2[pid, :, 15] -= 2.0
    out2 = op(dev[0], k2, v2, dev[3], dev[4], cu_total_seq_lens=dev[5])
    same = (out2 == out).flatten(1).all(dim=1)
    expect_changed = torch.zeros(8192, dtype=torch.bool, device=DEV)
    expect_changed[2047::2048] = True
    assert torch.equal(~same, expect_changed)
    # (3) the oracle on the first 96 and the last 64 query rows of sequence 1
    ref = torch_cls("MojoPagedPrefillGQA")()
    for lo, hi in ((0, 96), (2048 - 64, 2048)):
        n = hi - lo
        qs = q[2048 + lo: 2048 + hi]
        want = ref(qs, k, v, cu([n]), table[1:2], cu_total_seq_lens=cu([2048 + hi]))
        assert_close_tree(to_cpu(out[2048 + lo: 2048 + hi]), want, ATOL, RTOL)


@pytest.mark.parametrize("page,layout_nhd", [(16, False), (64, False), (128, True), (32, True)])
def test_prefill_fast_staging_is_bit_identical_to_general_staging(page, layout_nhd, monkeypatch):
    """Tiles in front of the diagonal are staged from a scalar page id + loop-invariant lane offsets when pages hold >= 16
    keys; the bytes that land in LDS are the same, so the outputs must be bit-identical to the per-key staging path.
    Ragged lengths, a cached prefix, a hole in one table, and a token-major (strided) cache view."""
    q_lens, cached = [700, 1, 333, 1024], [900, 515, 0, 77]
    q, k, v, cu_q, table, cu_kv, _ = make_prefill_inputs(q_lens, cached, 16, 4, 128, page, seed=23)
    table[0, 5] = -1
    k, v = k.to(DEV), v.to(DEV)
    if layout_nhd:                          # [blocks, page, heads, dim] storage seen through a [blocks, heads, page, dim] view
        k = k.permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)
        v = v.permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)
    op = hip_cls("MojoPagedPrefillGQA")()
    args = (q.to(DEV), k, v, cu_q.to(DEV), table.to(DEV))
    monkeypatch.setenv("MOJO_HIP_PREFILL_FAST_STAGE", "0")
    want = op(*args, cu_total_seq_lens=cu_kv.to(DEV))
    assert ":general_stage:" in last_launch(), last_launch()
    monkeypatch.setenv("MOJO_HIP_PREFILL_FAST_STAGE", "1")
    got = op(*args, cu_total_seq_lens=cu_kv.to(DEV))
    assert ":fast_stage:" in last_launch(), last_launch()             # two different staging paths really ran
    assert torch.equal(got, want)


def test_prefill_block_order_does_not_change_the_result(monkeypatch):
    """The (sequence, query block) -> workgroup mapping is rotated per query-block level so that a long sequence's blocks
    walk over the shader engines; every (sequence, block) must still be visited exactly once: same bits either way."""
    q_lens, cached = [700, 33, 0, 512, 129, 1000, 64], [0, 90, 10, 0, 300, 17, 0]
    q, k, v, cu_q, table, cu_kv, kv_lens = make_prefill_inputs(q_lens, cached, 8, 2, 128, 16, seed=11, pad_tokens=5)
    op = hip_cls("MojoPagedPrefillGQA")()
    args = [t.to(DEV) for t in (q, k, v, cu_q, table)]
    skip_unless_experiments_build()                                    # (the placement switch exists in experiments builds only)
    monkeypatch.setenv("MOJO_HIP_PREFILL_SKEW", "0")
    plain = op(*args, cu_total_seq_lens=cu_kv.to(DEV), max_q_len=max(q_lens), max_total_seq_len=max(kv_lens))
    monkeypatch.setenv("MOJO_HIP_PREFILL_SKEW", "1")
    rotated = op(*args, cu_total_seq_lens=cu_kv.to(DEV), max_q_len=max(q_lens), max_total_seq_len=max(kv_lens))
    assert torch.equal(plain, rotated)
    want = torch_cls("MojoPagedPrefillGQA")()(q, k, v, cu_q, table, cu_total_seq_lens=cu_kv)
    assert_close_tree(to_cpu(rotated), want, ATOL, RTOL)


@pytest.mark.parametrize("cfg", [
    # q_lens, cached, hq, hkv, d, page, forced slices (None: the launch's own rule)
    ([96], [1500], 32, 8, 128, 16, None),                    # chunked prefill of one sequence against a cache: split by the rule
    ([64, 33], [700, 0], 16, 4, 128, 32, 3),                 # ragged, one row without cache, odd slice count
    ([200, 1, 77], [0, 513, 129], 8, 2, 64, 16, 4),          # slices that end inside the diagonal, a one-token row, head_dim 64
    ([40], [90], 8, 8, 128, 16, 8),                          # more slices than key tiles: empty slices
], ids=["CHUNKED", "RAGGED_3", "DIAGONAL_4", "EMPTY_SLICES"])
@pytest.mark.parametrize("layout", ["AABB", "ABAB"])
def test_prefill_key_split(cfg, layout, monkeypatch):
    """PrefillArgs::ksplit (csrc/paged_prefill_gqa.hip): blocks cut along the keys into slices with fp32 partials and a
    merge launch — against the oracle, bit-stable from launch to launch, equal to the unsplit launch within accumulation
    noise, padding rows zeroed, holes honoured."""
    q_lens, cached, hq, hkv, d, page, forced = cfg
    q, k, v, cu_q, table, cu_kv, kv_lens = make_prefill_inputs(q_lens, cached, hq, hkv, d, page, seed=len(q_lens) + hq, pad_tokens=5)
    op = hip_cls("MojoPagedPrefillGQA")(is_causal=True, gqa_layout=layout)
    ref = torch_cls("MojoPagedPrefillGQA")(is_causal=True, gqa_layout=layout)
    kw = dict(max_q_len=max(q_lens), max_total_seq_len=max(kv_lens))
    if cu_kv is not None:
        kw["cu_total_seq_lens"] = cu_kv
    want = ref(q, k, v, cu_q, table, **kw)
    dkw = {k_: (v_.to(DEV) if isinstance(v_, torch.Tensor) else v_) for k_, v_ in kw.items()}
    dev = [t.to(DEV) for t in (q, k, v, cu_q, table)]
    if forced is not None:
        monkeypatch.setenv("MOJO_HIP_PREFILL_KSPLIT", str(forced))
    else:
        from mojo_opset_amd.backends.hip import lib as L
        assert L.load().mojo_hip_paged_prefill_gqa_workspace_bytes(q.shape[0], len(q_lens), hq, hkv, d, page, table.shape[1],
                                                                   max(q_lens), max(kv_lens)) > 0, "the rule should split this launch"
    got = op(*dev, **dkw)
    assert not last_launch().endswith(":ksplit1"), last_launch()           # the split form really ran
    assert_close_tree(to_cpu(got), want, ATOL, RTOL)
    assert torch.count_nonzero(got[sum(q_lens):]) == 0                     # padding rows
    assert torch.equal(op(*dev, **dkw), got)
    monkeypatch.setenv("MOJO_HIP_PREFILL_KSPLIT", "1")
    plain = op(*dev, **dkw)
    assert last_launch().endswith(":ksplit1"), last_launch()               # ... and the unsplit one
    torch.testing.assert_close(got.float(), plain.float(), atol=8e-3, rtol=8e-3)
    # a hole in the block table: rows behind the first negative id read as zero K/V, in the split form too
    table2 = table.clone()
    b0 = int(torch.tensor(kv_lens).argmax())
    if table2.shape[1] > 3:
        table2[b0, 2] = -1
        monkeypatch.setenv("MOJO_HIP_PREFILL_KSPLIT", str(forced or 4))
        want2 = ref(q, k, v, cu_q, table2, **kw)
        got2 = op(dev[0], dev[1], dev[2], dev[3], table2.to(DEV), **dkw)
        assert_close_tree(to_cpu(got2), want2, ATOL, RTOL)


@pytest.mark.parametrize("cfg", [
    # (q_lens, cached, Hq, Hkv, D, page): ragged batch with a cached prefix and a hole; groups of 1 / 8 heads; head_dim 64 / 96
    ([700, 1, 333, 1024], [900, 515, 0, 77], 16, 4, 128, 16),
    ([513, 64], [0, 2000], 8, 8, 128, 64),
    ([300, 129], [31, 0], 16, 2, 64, 16),
    ([257], [100], 8, 2, 96, 32),
], ids=["ragged_hole", "g1_cached", "g8_d64", "d96"])
def test_prefill_phase_alternating_kernel(cfg, monkeypatch):
    """prefill_pp_kernel (MOJO_HIP_PREFILL_PP=1; opt-in, see DESIGN Appendix A): 8-wave workgroups whose two wave groups
    alternate matrix and softmax phases over rings of four K / V tiles — against the oracle and against the default kernel."""
    skip_unless_experiments_build()
    q_lens, cached, hq, hkv, d, page = cfg
    q, k, v, cu_q, table, cu_kv, kv_lens = make_prefill_inputs(q_lens, cached, hq, hkv, d, page, seed=41)
    if len(q_lens) == 4:
        table[0, 5] = -1                                        # a hole: rows behind it read as zero K / V
    op = hip_cls("MojoPagedPrefillGQA")()
    ref = torch_cls("MojoPagedPrefillGQA")()
    kw = dict(softmax_scale=1.0 / math.sqrt(d), max_q_len=max(q_lens), max_total_seq_len=max(kv_lens), cu_total_seq_lens=cu_kv)
    want = ref(q, k, v, cu_q, table, **kw)
    dkw = {k_: (v_.to(DEV) if isinstance(v_, torch.Tensor) else v_) for k_, v_ in kw.items()}
    args = (q.to(DEV), k.to(DEV), v.to(DEV), cu_q.to(DEV), table.to(DEV))
    monkeypatch.setenv("MOJO_HIP_PREFILL_PP", "0")
    base = op(*args, **dkw)
    monkeypatch.setenv("MOJO_HIP_PREFILL_PP", "1")
    got = op(*args, **dkw)
    again = op(*args, **dkw)
    assert torch.equal(got, again)                              # same bits from run to run
    assert_close_tree(to_cpu(got), want, ATOL, RTOL)
    torch.testing.assert_close(got.float(), base.float(), atol=8e-3, rtol=8e-3)


def test_prefill_key_split_at_the_bench_size(monkeypatch):
    """The chunked-prefill bench case at full size (512 new tokens against 16 384 cached, 32 q / 8 kv heads, page 16): the
    launch's own rule cuts the keys into slices; the result equals the unsplit launch within accumulation noise, is
    bit-stable, and three sampled query rows match the oracle."""
    q_lens, cached, hq, hkv, d, page = [512], [16384], 32, 8, 128, 16
    q, k, v, cu_q, table, cu_kv, kv_lens = make_prefill_inputs(q_lens, cached, hq, hkv, d, page, seed=77)
    op = hip_cls("MojoPagedPrefillGQA")()
    dkw = dict(max_q_len=512, max_total_seq_len=kv_lens[0], cu_total_seq_lens=cu_kv.to(DEV))
    dev = [t.to(DEV) for t in (q, k, v, cu_q, table)]
    from mojo_opset_amd.backends.hip import lib as L
    assert L.load().mojo_hip_paged_prefill_gqa_workspace_bytes(512, 1, hq, hkv, d, page, table.shape[1], 512, kv_lens[0]) > 0
    got = op(*dev, **dkw)
    assert torch.equal(op(*dev, **dkw), got)
    monkeypatch.setenv("MOJO_HIP_PREFILL_KSPLIT", "1")
    plain = op(*dev, **dkw)
    torch.testing.assert_close(got.float(), plain.float(), atol=8e-3, rtol=8e-3)
    # oracle on three query rows (first, middle, last): row i sees keys 0 .. 16384 + i
    keys = torch.cat([k[int(p)] for p in table[0]], dim=1)[:, : kv_lens[0]].float()        # [hkv, T, d]
    vals = torch.cat([v[int(p)] for p in table[0]], dim=1)[:, : kv_lens[0]].float()
    for i in (0, 255, 511):
        n = 16384 + i + 1
        qi = q[i].float().view(hkv, hq // hkv, d)
        sc = torch.einsum("kgd,ktd->kgt", qi, keys[:, :n]) / math.sqrt(d)
        want = torch.einsum("kgt,ktd->kgd", torch.softmax(sc, -1).to(torch.bfloat16).float(), vals[:, :n]).reshape(hq, d)
        torch.testing.assert_close(got[i].float().cpu(), want, atol=ATOL, rtol=RTOL)


@pytest.mark.parametrize("cfg", [
    ([1024], [0], 32, 8, 128, 16),                      # pipelined iterations, 2x-unrolled loop taken several times
    ([513, 300, 64], [0, 2000, 100], 32, 8, 128, 16),   # ragged, cached, blocks of every length
    ([384], [0], 8, 8, 128, 64),                        # G = 1: 256 positions per workgroup, 64-key pages
    ([700, 129], [31, 0], 16, 2, 128, 32),              # G = 8
], ids=["long", "ragged_cached", "g1_page64", "g8"])
def test_prefill_one_wave_per_simd_kernel(cfg, monkeypatch):
    """prefill_w64_kernel (csrc/experiments/paged_prefill_w64.h, MOJO_HIP_PREFILL_W64=1 in an experiments build): one wave per
    SIMD, 64 rows per wave, the loop skewed one tile with the softmax in the MFMA gaps — against the oracle, against the
    default kernel, the same bits run to run, with a hole in a block table."""
    skip_unless_experiments_build()
    monkeypatch.setenv("MOJO_HIP_PREFILL_KSPLIT", "1")         # (the key split belongs to the default kernel)
    q_lens, cached, hq, hkv, d, page = cfg
    q, k, v, cu_q, table, cu_kv, kv_lens = make_prefill_inputs(q_lens, cached, hq, hkv, d, page, seed=43)
    if len(q_lens) == 3:
        table[1, 40] = -1                                       # a hole: rows behind it read as zero K / V
    op, ref = hip_cls("MojoPagedPrefillGQA")(), torch_cls("MojoPagedPrefillGQA")()
    kw = dict(softmax_scale=1.0 / math.sqrt(d), max_q_len=max(q_lens), max_total_seq_len=max(kv_lens), cu_total_seq_lens=cu_kv)
    want = ref(q, k, v, cu_q, table, **kw)
    dkw = {k_: (v_.to(DEV) if isinstance(v_, torch.Tensor) else v_) for k_, v_ in kw.items()}
    args = (q.to(DEV), k.to(DEV), v.to(DEV), cu_q.to(DEV), table.to(DEV))
    monkeypatch.setenv("MOJO_HIP_PREFILL_W64", "0")
    base = op(*args, **dkw)
    monkeypatch.setenv("MOJO_HIP_PREFILL_W64", "1")
    got = op(*args, **dkw)
    again = op(*args, **dkw)
    assert torch.equal(got, again)
    assert_close_tree(to_cpu(got), want, ATOL, RTOL)
    torch.testing.assert_close(got.float(), base.float(), atol=2e-2, rtol=2e-2)


@pytest.mark.parametrize("cfg", [([513, 300, 64], [0, 2000, 100], 32, 8, 128, 16), ([700, 129], [31, 0], 16, 2, 128, 32), ([512], [6000], 8, 8, 128, 16)],
                         ids=["ragged_cached", "g8", "key_split"])
def test_prefill_32x32_mfma_form(cfg, monkeypatch):
    """prefill_m32_kernel (csrc/experiments/paged_prefill_m32.h, MOJO_HIP_PREFILL_M32=1 in an experiments build): the default
    decomposition on v_mfma_f32_32x32x16 — oracle, default kernel, bit stability, key split included."""
    skip_unless_experiments_build()
    q_lens, cached, hq, hkv, d, page = cfg
    q, k, v, cu_q, table, cu_kv, kv_lens = make_prefill_inputs(q_lens, cached, hq, hkv, d, page, seed=47)
    op, ref = hip_cls("MojoPagedPrefillGQA")(), torch_cls("MojoPagedPrefillGQA")()
    kw = dict(softmax_scale=1.0 / math.sqrt(d), max_q_len=max(q_lens), max_total_seq_len=max(kv_lens), cu_total_seq_lens=cu_kv)
    want = ref(q, k, v, cu_q, table, **kw)
    dkw = {k_: (v_.to(DEV) if isinstance(v_, torch.Tensor) else v_) for k_, v_ in kw.items()}
    args = (q.to(DEV), k.to(DEV), v.to(DEV), cu_q.to(DEV), table.to(DEV))
    monkeypatch.setenv("MOJO_HIP_PREFILL_M32", "0")
    base = op(*args, **dkw)
    monkeypatch.setenv("MOJO_HIP_PREFILL_M32", "1")
    got = op(*args, **dkw)
    assert torch.equal(got, op(*args, **dkw))
    assert_close_tree(to_cpu(got), want, ATOL, RTOL)
    torch.testing.assert_close(got.float(), base.float(), atol=2e-2, rtol=2e-2)
