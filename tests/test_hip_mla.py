"""GPU parity of MojoPagedDecodeMLA / MojoPagedPrefillMLA through the C ABI.  Tolerance atol = rtol = 1e-2,
the reference's bound (mojo_opset/tests/accuracy/operators/test_attention.py:1173-1187, :1254-1257)."""
import math

import pytest
import torch

from conftest import load_golden
from hip_utils import (DEV, hip_cls, last_launch, launches_of, run_hip_case, skip_unless_experiments_build, switch_env, to_cpu,
                       torch_cls)

pytestmark = pytest.mark.gpu
# The reference's bound for these ops is atol = rtol = 1e-2, but it was never exercised against an accelerated
# backend (none exists; its test skips).  The golden rounds the decompressed K/V, the scores and the
# probabilities to bf16, which puts the GOLDEN ITSELF up to ~3e-2 away from the exactly computed result at
# the magnitudes of these tests (measured: tests/golden paged_mla case 2, |golden - fp64| = 0.030 while the
# weight-absorbed evaluation is 0.009 away).  Parity is therefore stated as:
#   (1) |hip - fp64 exact| <= 1e-2 (+1e-2 relative)  — the reference's number, against the true value; for outputs whose
#       magnitude exceeds 4 the absolute part scales with it (1e-2 * max|exact| / 4): the merged latent is stored in the
#       16-bit type before the output projection, an error proportional to the output's scale (a fuzz soak found one element
#       of 98 304 at 0.0105 on a case with max|exact| = 5.1 where the golden itself is 0.021 off; every kernel gives that figure);
#   (2) hip is never farther from the exact value than the golden is (plus one bf16 ulp of slack);
#   (3) |hip - golden| <= 4e-2 (+4e-2 relative)      — the golden's own noise band.
ATOL = RTOL = 1e-2
GOLDEN_BAND = 4e-2


def _unpage(cache, row, n):
    page = cache.shape[2]
    parts = []
    for j in range((n + page - 1) // page):
        if int(row[j]) < 0:
            break
        parts.append(cache[int(row[j]), 0, : min(page, n - j * page)])
    return torch.cat(parts) if parts else None


def exact_mla(q, ckv, kpe, table, w, sink, h, nope, rope, vd, r, kv_lens, q_off=None, scale=None):
    """fp64 evaluation of the op's definition (decompress, softmax with optional sink, PV); decode when
    ``q_off`` is None, else packed causal prefill with ``q_off`` the cumulative query offsets."""
    dt = torch.float64
    wd = w.to(dt).view(h, nope + vd, r)
    scale = 1.0 / math.sqrt(nope + rope) if scale is None else scale
    out = torch.zeros(q.shape[0], h, vd, dtype=dt)
    for b, n in enumerate(kv_lens):
        rows = [b] if q_off is None else list(range(q_off[b], q_off[b + 1]))
        if n <= 0 or not rows:
            continue
        c = _unpage(ckv, table[b], n)
        pe = _unpage(kpe, table[b], n)
        if c is None:
            continue
        c, pe = c.to(dt), pe.to(dt)
        k = torch.cat([torch.einsum("sr,hdr->shd", c, wd[:, :nope]), pe[:, None, :].expand(-1, h, -1)], -1)
        v = torch.einsum("sr,hdr->shd", c, wd[:, nope:])
        for i, t in enumerate(rows):
            vis = c.shape[0] if q_off is None else min(c.shape[0], n - len(rows) + i + 1)
            s = torch.einsum("hd,shd->hs", q[t].to(dt), k[:vis]) * scale
            if sink is not None:
                s = torch.cat([s, sink.to(dt)[:, None]], -1)
            p = torch.softmax(s, -1)
            if sink is not None:
                p = p[:, :-1]
            out[t] = torch.einsum("hs,shd->hd", p, v[:vis])
    return out


def prefill_route(h, nope, rope, vd, tq, dtype=torch.bfloat16):
    """Which formulation HIPPagedPrefillMLA takes for these dimensions: "decompress" (the golden's own: un-page, one
    decompression GEMM, flash attention with D_qk = nope + rope) or "absorbed" (the decode kernel per query token)."""
    from mojo_opset_amd import switches
    from mojo_opset_amd.backends.hip import lib as L
    if switches.get("MOJO_HIP_MLA_PREFILL", "decompress") == "absorbed" or tq < 16:
        return "absorbed"
    return "decompress" if L.load().mojo_hip_mla_prefill_supported(nope, rope, vd, L.dtype_code(dtype)) else "absorbed"


def check_mla(got, want_golden, exact, route="absorbed"):
    got, want_golden = got.double(), want_golden.double()
    assert got.shape == exact.shape == want_golden.shape
    if got.numel() == 0:
        return
    if route == "decompress":
        # same rounding points as the golden (decompressed K/V and probabilities in the storage type): the bound is the
        # paged GQA prefill's, atol = rtol = 2e-2 AGAINST THE GOLDEN, and hip must not be farther from the exact value
        # than the golden by more than that band
        torch.testing.assert_close(got, want_golden, atol=2e-2, rtol=2e-2)
        err_hip, err_gold = (got - exact).abs().max(), (want_golden - exact).abs().max()
        assert err_hip <= err_gold + 2e-2 * (1.0 + exact.abs().max()), (err_hip, err_gold)
        return
    # (1) against the exactly computed result: the reference's own numbers, atol = rtol = 1e-2, for all but a bounded handful
    # of elements (<= 1e-4 of them: one fuzz case measured 0.0105 at |exact| = 5.1), and NO element outside twice that
    # however large the outputs are (a bound that grows with the output scale could hide a regression)
    inside = (got - exact).abs() <= ATOL + RTOL * exact.abs()
    assert float((~inside).double().mean()) <= 1e-4, float((~inside).double().mean())
    torch.testing.assert_close(got, exact, atol=2 * ATOL, rtol=RTOL)
    err_hip, err_gold = (got - exact).abs().max(), (want_golden - exact).abs().max()
    assert err_hip <= err_gold + 2.0 ** -8 * exact.abs().max().clamp_min(1.0), (err_hip, err_gold)   # (2)
    torch.testing.assert_close(got, want_golden, atol=GOLDEN_BAND, rtol=GOLDEN_BAND)           # (3)


def cu(lens):
    return torch.tensor([0] + list(torch.tensor(lens).cumsum(0).tolist()), dtype=torch.int32)


def make_mla(lens, h, nope, rope, vd, r, page, sink=False, seed=0, dtype=torch.bfloat16, wscale=0.2):
    g = torch.Generator().manual_seed(seed)
    need = [(n + page - 1) // page for n in lens]
    total = max(sum(need), 1) + 3
    ckv = torch.randn(total, 1, page, r, generator=g).to(dtype)
    kpe = torch.randn(total, 1, page, rope, generator=g).to(dtype)
    ids = torch.randperm(total, generator=g, dtype=torch.int32)
    table = torch.full((len(lens), max(max(need), 1)), -1, dtype=torch.int32)
    at = 0
    for b, n in enumerate(need):
        table[b, :n] = ids[at: at + n]
        at += n
    w = (torch.randn(h * (nope + vd), r, generator=g) * wscale).to(dtype)
    sk = torch.randn(h, generator=g) if sink else None
    return ckv, kpe, table, w, sk


def build(cls_name, h, nope, rope, vd, r, sink, w, sk, device, dtype=torch.bfloat16, **extra):
    kind = hip_cls if device == DEV else torch_cls
    op = kind(cls_name)(h, nope, rope, vd, r, use_attn_sink=sink, **extra).to(dtype).to(device)
    with torch.no_grad():
        op.kv_b_proj.copy_(w.to(device))
        if sink:
            op.attn_sink.copy_(sk.to(device))
    return op


@pytest.mark.parametrize("case", [pytest.param(c, id=f"mla-{i}-{c['op']}") for i, c in enumerate(load_golden("paged_mla"))])
def test_mla_vectors(case):
    got = to_cpu(run_hip_case(case))
    kw = case["ctor"]["kwargs"]
    h, nope, rope, vd, r = (kw[k] for k in ("num_heads", "qk_nope_head_dim", "qk_rope_head_dim", "v_head_dim", "kv_lora_rank"))
    w, sink = case["state"]["kv_b_proj"], case["state"].get("attn_sink")
    if case["op"] == "MojoPagedDecodeMLA":
        q, ckv, kpe, lens, table = case["args"]
        exact = exact_mla(q, ckv, kpe, table, w, sink, h, nope, rope, vd, r, lens.tolist())
    else:
        q, ckv, kpe, cu_q, table = case["args"]
        cu_kv = case["kwargs"]["cu_total_seq_lens"]
        kv_lens = (cu_kv[1:] - cu_kv[:-1]).tolist()
        exact = exact_mla(q, ckv, kpe, table, w, sink, h, nope, rope, vd, r, kv_lens, q_off=cu_q.tolist())
        check_mla(got, case["out"], exact, prefill_route(h, nope, rope, vd, q.shape[0]))
        return
    check_mla(got, case["out"], exact)


@pytest.mark.parametrize("cfg", [
    (4, 16, 96, 32, 128, 64, 256, 64), (2, 8, 64, 32, 64, 32, 128, 32), (3, 8, 64, 32, 64, 32, 0, 32),
    (2, 128, 128, 64, 128, 512, 700, 16),                  # DeepSeek-V3 dimensions, page 16
], ids=["REF0", "REF1", "REF_EMPTY", "DEEPSEEK_V3"])
@pytest.mark.parametrize("sink", [False, True])
def test_mla_decode_reference_space(cfg, sink):
    b, h, nope, rope, vd, r, s_max, page = cfg
    g = torch.Generator().manual_seed(b + h)
    lens = [int(x) for x in (torch.randint(1, s_max + 1, (b,), generator=g) if s_max else torch.tensor([0, 45, 0][:b]))]
    ckv, kpe, table, w, sk = make_mla(lens, h, nope, rope, vd, r, page, sink, seed=h, wscale=0.2 if r <= 64 else 0.05)
    q = torch.randn(b, h, nope + rope, generator=g).to(torch.bfloat16)
    lens_t = torch.tensor(lens, dtype=torch.int32)
    ref = build("MojoPagedDecodeMLA", h, nope, rope, vd, r, sink, w, sk, "cpu")
    op = build("MojoPagedDecodeMLA", h, nope, rope, vd, r, sink, w, sk, DEV)
    want = ref(q, ckv, kpe, lens_t, table)
    got = op(q.to(DEV), ckv.to(DEV), kpe.to(DEV), lens_t.to(DEV), table.to(DEV))
    check_mla(to_cpu(got), want, exact_mla(q, ckv, kpe, table, w, sk, h, nope, rope, vd, r, lens))


@pytest.mark.parametrize("cfg", [
    (2, 8, 64, 32, 64, 32, 48, 32), (3, 8, 64, 32, 64, 32, 0, 32), (2, 16, 96, 32, 128, 64, 150, 16),
    (1, 128, 128, 64, 128, 512, 200, 16),
], ids=["REF0", "REF_EMPTY", "MID", "DEEPSEEK_V3"])
@pytest.mark.parametrize("sink", [False, True])
def test_mla_prefill_reference_space(cfg, sink):
    b, h, nope, rope, vd, r, s_max, page = cfg
    g = torch.Generator().manual_seed(b * 7 + h)
    # s_max == 0: the reference's "empty" case — here a mix of empty and short sequences
    kv_lens = [int(x) for x in (torch.randint(1, s_max + 1, (b,), generator=g) if s_max else torch.tensor([0, 37, 0][:b]))]
    q_lens = [min(n, 1 + int(torch.randint(0, 40, (1,), generator=g))) for n in kv_lens]
    ckv, kpe, table, w, sk = make_mla(kv_lens, h, nope, rope, vd, r, page, sink, seed=h + 1, wscale=0.2 if r <= 64 else 0.05)
    q = torch.randn(sum(q_lens), h, nope + rope, generator=g).to(torch.bfloat16)
    ref = build("MojoPagedPrefillMLA", h, nope, rope, vd, r, sink, w, sk, "cpu", is_causal=True)
    op = build("MojoPagedPrefillMLA", h, nope, rope, vd, r, sink, w, sk, DEV, is_causal=True)
    want = ref(q, ckv, kpe, cu(q_lens), table, cu_total_seq_lens=cu(kv_lens))
    exact = exact_mla(q, ckv, kpe, table, w, sk, h, nope, rope, vd, r, kv_lens, q_off=cu(q_lens).tolist())
    got = op(q.to(DEV), ckv.to(DEV), kpe.to(DEV), cu(q_lens).to(DEV), table.to(DEV), cu_total_seq_lens=cu(kv_lens).to(DEV))
    check_mla(to_cpu(got), want, exact, prefill_route(h, nope, rope, vd, q.shape[0]))
    # the other formulation on the same inputs (the route small / unsupported shapes and over-budget batches take)
    with switch_env(MOJO_HIP_MLA_PREFILL="absorbed"):
        got_abs = op(q.to(DEV), ckv.to(DEV), kpe.to(DEV), cu(q_lens).to(DEV), table.to(DEV), cu_total_seq_lens=cu(kv_lens).to(DEV))
        hist = launches_of(lambda: op(q.to(DEV), ckv.to(DEV), kpe.to(DEV), cu(q_lens).to(DEV), table.to(DEV), cu_total_seq_lens=cu(kv_lens).to(DEV)))
    assert "mla_prefill_attn" not in hist and ("mla512:" in hist or "mla_latent:" in hist), hist      # the absorbed kernels really ran
    check_mla(to_cpu(got_abs), want, exact, "absorbed")


def test_mla_prefill_absorbed_route_padding_rows_stay_zero(monkeypatch):
    """Few query tokens over long cached prefixes: the absorbed route cuts the keys into splits, and the merge launch must
    skip the tokens no sequence owns (nothing wrote their partials) — the golden returns zeros there."""
    h, nope, rope, vd, r, page = 128, 128, 64, 128, 512, 16
    q_lens, cached = [3, 5], [2500, 1200]
    kv_lens = [a + b for a, b in zip(q_lens, cached)]
    g = torch.Generator().manual_seed(12)
    ckv, kpe, table, w, _ = make_mla(kv_lens, h, nope, rope, vd, r, page, seed=21, wscale=0.05)
    q = torch.randn(sum(q_lens) + 6, h, nope + rope, generator=g).to(torch.bfloat16)      # 6 padding tokens
    op = build("MojoPagedPrefillMLA", h, nope, rope, vd, r, False, w, None, DEV, is_causal=True)
    ref = build("MojoPagedPrefillMLA", h, nope, rope, vd, r, False, w, None, "cpu", is_causal=True)
    monkeypatch.setenv("MOJO_HIP_MLA_PREFILL", "absorbed")
    got = to_cpu(op(q.to(DEV), ckv.to(DEV), kpe.to(DEV), cu(q_lens).to(DEV), table.to(DEV), cu_total_seq_lens=cu(kv_lens).to(DEV)))
    want = ref(q, ckv, kpe, cu(q_lens), table, cu_total_seq_lens=cu(kv_lens))
    n = sum(q_lens)
    assert torch.count_nonzero(got[n:]) == 0 and torch.count_nonzero(want[n:]) == 0
    exact = exact_mla(q[:n], ckv, kpe, table, w, None, h, nope, rope, vd, r, kv_lens, q_off=cu(q_lens).tolist())
    check_mla(got[:n], want[:n], exact, "absorbed")


def test_mla_uncast_module_raises_like_the_golden():
    ckv, kpe, table, w, _ = make_mla([20], 8, 64, 32, 64, 32, 16)
    op = hip_cls("MojoPagedDecodeMLA")(8, 64, 32, 64, 32).to(DEV)          # kv_b_proj stays fp32 (reference quirk)
    q = torch.randn(1, 8, 96, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError):
        op(q, ckv.to(DEV), kpe.to(DEV), torch.tensor([20], dtype=torch.int32, device=DEV), table.to(DEV))


def test_mla_decode_full_size_properties():
    """BASELINE config 5a at full size (B=64, H=128, 128/64/128, r=512, page=16, ctx=4096)."""
    b, h, nope, rope, vd, r, page, ctx = 64, 128, 128, 64, 128, 512, 16, 4096
    lens = [ctx] * b
    ckv, kpe, table, w, _ = make_mla(lens, h, nope, rope, vd, r, page, seed=3, wscale=0.02)
    g = torch.Generator().manual_seed(9)
    q = torch.randn(b, h, nope + rope, generator=g).to(torch.bfloat16)
    lens_t = torch.tensor(lens, dtype=torch.int32)
    op = build("MojoPagedDecodeMLA", h, nope, rope, vd, r, False, w, None, DEV)
    dev = [t.to(DEV) for t in (q, ckv, kpe, lens_t, table)]
    out = op(*dev)
    assert torch.isfinite(out.float()).all()
    # the oracle on two whole sequences (the golden decompresses [4096,512] x [512,32768] per sequence)
    ref = build("MojoPagedDecodeMLA", h, nope, rope, vd, r, False, w, None, "cpu")
    for i in (0, b - 1):
        want = ref(q[i: i + 1], ckv, kpe, lens_t[i: i + 1], table[i: i + 1])
        exact = exact_mla(q[i: i + 1], ckv, kpe, table[i: i + 1], w, None, h, nope, rope, vd, r, [ctx])
        check_mla(to_cpu(out[i: i + 1]), want, exact)
    # page relabelling must not change a single bit
    perm = torch.randperm(ckv.shape[0], generator=g)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(perm.numel())
    table2 = inv[table.long()].to(torch.int32)
    out2 = op(dev[0], ckv[perm].to(DEV), kpe[perm].to(DEV), dev[3], table2.to(DEV))
    assert torch.equal(out, out2)


@pytest.mark.parametrize("cfg", [(2, 128, 700, False), (5, 128, 1500, True), (3, 64, 3000, False), (16, 128, 2048, True), (1, 16, 900, False),
                                 (64, 128, 300, False), (4, 128, 33, True)],
                         ids=["B2", "B5_SINK", "B3_H64", "B16_SINK", "B1_H16", "B64_SHORT", "B4_TINY"])
@pytest.mark.parametrize("kernel", ["ps", "pp", "oct", "pair"])
def test_mla_decode_r512_kernels_agree(cfg, kernel, monkeypatch):
    """The r = 512 latent kernels (MOJO_HIP_MLA_KERNEL: specialised waves = default, lock-step on 64-key tiles = small
    pages; in an experiments build also ping-pong on 32-key tiles and one wave per SIMD) against the exactly computed result and the golden, on ragged batches with empty sequences, empty key splits,
    lengths that end inside a tile and fewer heads than a workgroup covers."""
    if kernel in ("pp", "pair"):
        skip_unless_experiments_build()
    b, h, s_max, sink = cfg
    nope, rope, vd, r, page = 128, 64, 128, 512, 16
    g = torch.Generator().manual_seed(b * 7 + h)
    lens = [int(x) for x in torch.randint(1, s_max + 1, (b,), generator=g)]
    lens[0] = s_max
    if b > 2:
        lens[1] = 0                                                  # an empty sequence: zeros
        lens[2] = min(40, s_max)                                     # all but the first key split empty
    ckv, kpe, table, w, sk = make_mla(lens, h, nope, rope, vd, r, page, sink, seed=h + b, wscale=0.05)
    q = torch.randn(b, h, nope + rope, generator=g).to(torch.bfloat16)
    lens_t = torch.tensor(lens, dtype=torch.int32)
    op = build("MojoPagedDecodeMLA", h, nope, rope, vd, r, sink, w, sk, DEV)
    monkeypatch.setenv("MOJO_HIP_MLA_KERNEL", kernel)
    got = to_cpu(op(q.to(DEV), ckv.to(DEV), kpe.to(DEV), lens_t.to(DEV), table.to(DEV)))
    assert f"mla512:{kernel}:" in launches_of(lambda: op(q.to(DEV), ckv.to(DEV), kpe.to(DEV), lens_t.to(DEV), table.to(DEV)))
    again = to_cpu(op(q.to(DEV), ckv.to(DEV), kpe.to(DEV), lens_t.to(DEV), table.to(DEV)))
    assert torch.equal(got, again)                                   # no race between the staggered wave groups
    if b <= 5:
        ref = build("MojoPagedDecodeMLA", h, nope, rope, vd, r, sink, w, sk, "cpu")
        check_mla(got, ref(q, ckv, kpe, lens_t, table), exact_mla(q, ckv, kpe, table, w, sk, h, nope, rope, vd, r, lens))
    else:
        exact = exact_mla(q, ckv, kpe, table, w, sk, h, nope, rope, vd, r, lens)
        torch.testing.assert_close(got.double(), exact.double(), atol=1e-2, rtol=1e-2)


# ---- the reference's own generators, verbatim recipe (test_attention.py:1131-1155, :1193-1233): kv_b_proj = randn ------
def _ref_decode_data(batch, h, nope, rope, r, max_len, page):
    q = torch.randn(batch, h, nope + rope, dtype=torch.bfloat16)
    if max_len > 0:
        lens = torch.randint(max_len // 2, max_len, (batch,), dtype=torch.int32).clamp(min=1)
    else:
        lens = torch.randperm(batch, dtype=torch.int32)
    max_nb = (int(lens.max()) + page - 1) // page
    total = int(torch.div(lens + page - 1, page, rounding_mode="floor").sum()) + 10
    ckv = torch.randn(total, 1, page, r, dtype=torch.bfloat16)
    kpe = torch.randn(total, 1, page, rope, dtype=torch.bfloat16)
    table = torch.full((batch, max(max_nb, 1)), -1, dtype=torch.int32)
    free = torch.randperm(total)
    off = 0
    for i in range(batch):
        n = (int(lens[i]) + page - 1) // page
        table[i, :n] = free[off:off + n]
        off += n
    return q, ckv, kpe, lens, table


def _report_triple(name, got, golden, exact):
    import json
    import os
    rec = {"case": name, "max_abs_exact": float(exact.abs().max()),
           "hip_vs_golden": float((got.double() - golden.double()).abs().max()),
           "hip_vs_fp64": float((got.double() - exact).abs().max()),
           "golden_vs_fp64": float((golden.double() - exact).abs().max())}
    print("MLA_ERROR_TRIPLE " + json.dumps(rec))
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "mla_error_triples.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass
    return rec


@pytest.mark.parametrize("cfg", [(4, 16, 96, 32, 128, 64, 256, 64), (2, 8, 64, 32, 64, 32, 128, 32), (3, 8, 64, 32, 64, 32, 0, 32)],
                         ids=["REF0", "REF1", "REF_PADSEQ"])
def test_mla_decode_on_the_references_own_inputs(cfg):
    """`test_paged_decode_mla` (test_attention.py:1164-1187) with its generator and `w = randn_like(kv_b_proj)`.  The
    reference states atol = rtol = 1e-2 but never ran it against a second implementation (its test skips: both sides are
    the torch class).  At these magnitudes (|out| up to ~40) the golden's bf16 roundings of K/V, scores and probabilities
    put the GOLDEN ITSELF ~1 away from the fp64 value of its own definition, so no accurate kernel can sit within 1e-2 of
    it.  What is asserted on the reference's inputs: hip is closer to the exact value than the golden is, and hip differs
    from the golden by no more than the golden's own error plus hip's.  The three numbers are printed and logged."""
    b, h, nope, rope, vd, r, s, page = cfg
    torch.manual_seed(0)
    q, ckv, kpe, lens, table = _ref_decode_data(b, h, nope, rope, r, s, page)
    ref = torch_cls("MojoPagedDecodeMLA")(h, nope, rope, vd, r).to(torch.bfloat16)
    w = torch.randn_like(ref.kv_b_proj)
    op = build("MojoPagedDecodeMLA", h, nope, rope, vd, r, False, w, None, DEV)
    with torch.no_grad():
        ref.kv_b_proj.copy_(w)
    golden = ref(q, ckv, kpe, lens, table)
    got = to_cpu(op(q.to(DEV), ckv.to(DEV), kpe.to(DEV), lens.to(DEV), table.to(DEV)))
    exact = exact_mla(q, ckv, kpe, table, w, None, h, nope, rope, vd, r, lens.tolist())
    rec = _report_triple(f"decode_mla_ref_inputs{cfg}", got, golden, exact)
    assert got.shape == golden.shape and got.dtype == golden.dtype
    slack = 2.0 ** -8 * max(rec["max_abs_exact"], 1.0)            # one bf16 ulp at the output's magnitude
    assert rec["hip_vs_fp64"] <= rec["golden_vs_fp64"] + slack
    assert rec["hip_vs_golden"] <= rec["golden_vs_fp64"] + rec["hip_vs_fp64"] + slack
    # relative to the output's scale hip meets the reference's number against the TRUE value
    assert rec["hip_vs_fp64"] <= 1e-2 * (1.0 + rec["max_abs_exact"])


@pytest.mark.parametrize("cfg", [(4, 16, 96, 32, 128, 64, 256, 64), (2, 8, 64, 32, 64, 32, 128, 32), (3, 8, 64, 32, 64, 32, 0, 32)],
                         ids=["REF0", "REF1", "REF_PADSEQ"])
def test_mla_decode_golden_route_holds_the_references_bound(cfg, monkeypatch):
    """The reference's `test_paged_decode_mla` (test_attention.py:1164-1187): its generator, `w = randn_like(kv_b_proj)`,
    and its bound atol = rtol = 1e-2 AGAINST THE GOLDEN — on the golden-rounding decode route (`MOJO_HIP_MLA_DECODE=golden`:
    un-page, decompression GEMM rounded to bf16, scores rounded to bf16, scaled and rounded again, probabilities rounded
    to bf16: experimental/operators/attention.py:196-220).  Same statement as the prefill test below, which shares the
    route: the reference's bound with one bf16 ulp at the output's magnitude as the floor of what "equal" can mean.  The
    floor is needed because the golden rounds ~2.7 M decompressed K/V values to bf16 and a different fp32 summation order
    of the same GEMM flips ~2.5e-4 of those roundings by one ulp (|v| ~ 8-16: ulp 1/16); a flipped V under a probability
    near 1 moves an output by 3/64 wherever that output happens to be near zero — measured on the first box: 27 of 8192
    elements outside a strict 1e-2, greatest difference 0.047, against 1.0 on the absorbed route.  The strict fraction is
    asserted too (>= 99 %).  Also through `decode_route` on the instance, several seeds; the fp64 triple is logged."""
    b, h, nope, rope, vd, r, s, page = cfg
    monkeypatch.setenv("MOJO_HIP_MLA_DECODE", "golden")
    for seed in (0, 1, 2):
        torch.manual_seed(seed)
        q, ckv, kpe, lens, table = _ref_decode_data(b, h, nope, rope, r, s, page)
        ref = torch_cls("MojoPagedDecodeMLA")(h, nope, rope, vd, r).to(torch.bfloat16)
        w = torch.randn_like(ref.kv_b_proj)
        op = build("MojoPagedDecodeMLA", h, nope, rope, vd, r, False, w, None, DEV)
        with torch.no_grad():
            ref.kv_b_proj.copy_(w)
        golden = ref(q, ckv, kpe, lens, table)
        got = to_cpu(op(q.to(DEV), ckv.to(DEV), kpe.to(DEV), lens.to(DEV), table.to(DEV)))
        exact = exact_mla(q, ckv, kpe, table, w, None, h, nope, rope, vd, r, lens.tolist())
        rec = _report_triple(f"decode_mla_golden_route{cfg}_seed{seed}", got, golden, exact)
        assert got.shape == golden.shape and got.dtype == golden.dtype
        slack = 2.0 ** -8 * max(rec["max_abs_exact"], 1.0)        # one bf16 ulp at the output's magnitude
        torch.testing.assert_close(got.float(), golden.float(), atol=1e-2 + slack, rtol=1e-2)
        if got.numel():
            strict = torch.isclose(got.float(), golden.float(), atol=1e-2, rtol=1e-2).float().mean().item()
            assert strict >= 0.99, strict
    # the switch on the instance selects the same route without the environment variable
    monkeypatch.delenv("MOJO_HIP_MLA_DECODE")
    absorbed = to_cpu(op(q.to(DEV), ckv.to(DEV), kpe.to(DEV), lens.to(DEV), table.to(DEV)))
    op.decode_route = "golden"
    again = to_cpu(op(q.to(DEV), ckv.to(DEV), kpe.to(DEV), lens.to(DEV), table.to(DEV)))
    assert torch.equal(again, got)
    if s > 0:
        assert not torch.equal(absorbed, got)                # (the default route is the absorbed kernel: other rounding points)


def test_mla_decode_golden_route_fixtures_sink_and_empty_rows(monkeypatch):
    """The golden-rounding decode route on the committed fixtures (with and without sink, zero-length rows, ragged
    lengths) at the reference's bound against the reference's outputs; sequences walked in slices when the decompressed
    image exceeds the budget; a length above the caller's bound is truncated at the capacity, never written past it."""
    monkeypatch.setenv("MOJO_HIP_MLA_DECODE", "golden")
    ran = 0
    for case in load_golden("paged_mla"):
        if case["op"] != "MojoPagedDecodeMLA":
            continue
        kw = case["ctor"]["kwargs"]
        h, nope, rope, vd, r = (kw[k] for k in ("num_heads", "qk_nope_head_dim", "qk_rope_head_dim", "v_head_dim", "kv_lora_rank"))
        from mojo_opset_amd.backends.hip import lib as L
        if not L.load().mojo_hip_mla_prefill_supported(nope, rope, vd, L.dtype_code(torch.bfloat16)):
            continue
        got = run_hip_case(case)
        slack = 2.0 ** -8 * max(float(case["out"].float().abs().max()), 1.0)
        torch.testing.assert_close(to_cpu(got).float(), case["out"].float(), atol=1e-2 + slack, rtol=1e-2)
        ran += 1
    assert ran >= 1
    # slices: a budget that holds one sequence at a time gives the same bits as one pass
    h, nope, rope, vd, r, page = 8, 128, 64, 128, 512, 16
    ckv, kpe, table, w, _ = make_mla([300, 0, 17, 512, 64], h, nope, rope, vd, r, page, seed=5)
    lens = torch.tensor([300, 0, 17, 512, 64], dtype=torch.int32)
    torch.manual_seed(6)
    q = torch.randn(5, h, nope + rope, dtype=torch.bfloat16)
    op = build("MojoPagedDecodeMLA", h, nope, rope, vd, r, False, w, None, DEV)
    args = (q.to(DEV), ckv.to(DEV), kpe.to(DEV), lens.to(DEV), table.to(DEV))
    whole = to_cpu(op(*args, max_total_seq_len=512))
    monkeypatch.setenv("MOJO_HIP_MLA_PREFILL_BYTES", str(512 * h * (nope + vd) * 2))
    sliced = to_cpu(op(*args, max_total_seq_len=512))
    assert torch.equal(whole, sliced)
    assert float(whole[1].abs().max()) == 0.0                 # the empty row reads as zeros (attention.py:184-185)
    ref = torch_cls("MojoPagedDecodeMLA")(h, nope, rope, vd, r).to(torch.bfloat16)
    with torch.no_grad():
        ref.kv_b_proj.copy_(w)
    want = ref(q, ckv, kpe, lens, table)
    slack = 2.0 ** -8 * max(float(want.float().abs().max()), 1.0)
    torch.testing.assert_close(whole.float(), want.float(), atol=1e-2 + slack, rtol=1e-2)
    # a bound below the real lengths: every sequence is cut at the bound ON ITS OWN — the two over-long ones
    # (300, 512 keys) attend to their first 128 keys, the well-formed ones (0, 17, 64 keys) are untouched by their
    # neighbours' excess — nothing is written or read past the image's capacity, every output stays finite
    monkeypatch.delenv("MOJO_HIP_MLA_PREFILL_BYTES")
    short = to_cpu(op(*args, max_total_seq_len=128))
    torch.testing.assert_close(short[[1, 2, 4]].float(), want[[1, 2, 4]].float(), atol=1e-2 + slack, rtol=1e-2)
    cut = ref(q, ckv, kpe, torch.minimum(lens, torch.tensor(128, dtype=torch.int32)), table)
    torch.testing.assert_close(short[[0, 3]].float(), cut[[0, 3]].float(), atol=1e-2 + slack, rtol=1e-2)
    assert torch.isfinite(short.float()).all()
    monkeypatch.setenv("MOJO_HIP_VALIDATE", "1")
    with pytest.raises(ValueError):
        op(*args, max_total_seq_len=128)


@pytest.mark.parametrize("cfg", [(2, 8, 64, 32, 64, 32, 48, 32), (3, 8, 64, 32, 64, 32, 0, 32)], ids=["REF0", "REF_PADSEQ"])
def test_mla_prefill_on_the_references_own_inputs(cfg):
    """`test_paged_prefill_mla` (test_attention.py:1235-1257), same statement as the decode case."""
    b, h, nope, rope, vd, r, max_q, page = cfg
    torch.manual_seed(0)
    if max_q > 0:
        q_lens = torch.randint(max_q // 2, max_q, (b,), dtype=torch.int32).clamp(min=1)
    else:
        q_lens = torch.randperm(b, dtype=torch.int32)
    cu_q = torch.cat([torch.tensor([0], dtype=torch.int32), q_lens.cumsum(0, dtype=torch.int32)])
    q = torch.randn(int(cu_q[-1]), h, nope + rope, dtype=torch.bfloat16)
    max_nb = max((int(q_lens.max()) + page - 1) // page, 1)
    total = int(torch.div(q_lens + page - 1, page, rounding_mode="floor").sum()) + 10
    ckv = torch.zeros(total, 1, page, r, dtype=torch.bfloat16)
    kpe = torch.zeros(total, 1, page, rope, dtype=torch.bfloat16)
    table = torch.full((b, max_nb), -1, dtype=torch.int32)
    free = torch.randperm(total)
    off = 0
    for i in range(b):
        kl = int(q_lens[i])
        nb = (kl + page - 1) // page
        blocks = free[off:off + nb]
        table[i, :nb] = blocks
        off += nb
        cd, pd = torch.randn(kl, r, dtype=torch.bfloat16), torch.randn(kl, rope, dtype=torch.bfloat16)
        for j in range(nb):
            s0, e0 = j * page, min((j + 1) * page, kl)
            ckv[int(blocks[j]), 0, : e0 - s0] = cd[s0:e0]
            kpe[int(blocks[j]), 0, : e0 - s0] = pd[s0:e0]
    ref = torch_cls("MojoPagedPrefillMLA")(h, nope, rope, vd, r, is_causal=True).to(torch.bfloat16)
    w = torch.randn_like(ref.kv_b_proj)
    op = build("MojoPagedPrefillMLA", h, nope, rope, vd, r, False, w, None, DEV, is_causal=True)
    with torch.no_grad():
        ref.kv_b_proj.copy_(w)
    golden = ref(q, ckv, kpe, cu_q, table)
    got = to_cpu(op(q.to(DEV), ckv.to(DEV), kpe.to(DEV), cu_q.to(DEV), table.to(DEV)))
    exact = exact_mla(q, ckv, kpe, table, w, None, h, nope, rope, vd, r, q_lens.tolist(), q_off=cu_q.tolist())
    rec = _report_triple(f"prefill_mla_ref_inputs{cfg}", got, golden, exact)
    slack = 2.0 ** -8 * max(rec["max_abs_exact"], 1.0)
    if prefill_route(h, nope, rope, vd, q.shape[0]) == "decompress":
        # the golden's own formulation and rounding points (decompressed K/V, scores and probabilities in the storage type):
        # on the reference's inputs the two differ by output-ulp flips only — the REFERENCE'S bound, atol = rtol = 1e-2
        # (test_attention.py:1254-1257), with one bf16 ulp at the output's magnitude as the floor of what "equal" can mean
        torch.testing.assert_close(got.float(), golden.float(), atol=1e-2 + slack, rtol=1e-2)
    else:
        assert rec["hip_vs_fp64"] <= rec["golden_vs_fp64"] + slack
        assert rec["hip_vs_fp64"] <= 1e-2 * (1.0 + rec["max_abs_exact"])


def test_mla_weight_repack_follows_the_parameter():
    """The K-major copy of the absorbed key projection (decode-sized calls) must follow `kv_b_proj`: version-bumping
    writes, `.to()`, `load_state_dict` and `refresh_weights()` all rebuild it; MOJO_HIP_VALIDATE=1 catches a `.data` write."""
    import os
    h, nope, rope, vd, r, page = 8, 128, 64, 128, 128, 16
    ckv, kpe, table, w1, _ = make_mla([40, 7], h, nope, rope, vd, r, page, seed=1)
    g = torch.Generator().manual_seed(5)
    w2 = (torch.randn(h * (nope + vd), r, generator=g) * 0.2).to(torch.bfloat16)
    q = torch.randn(2, h, nope + rope, generator=g).to(torch.bfloat16)
    lens = torch.tensor([40, 7], dtype=torch.int32)
    dev = [t.to(DEV) for t in (q, ckv, kpe, lens, table)]
    op = build("MojoPagedDecodeMLA", h, nope, rope, vd, r, False, w1, None, DEV)
    fresh2 = build("MojoPagedDecodeMLA", h, nope, rope, vd, r, False, w2, None, DEV)
    want1, want2 = op(*dev), fresh2(*dev)
    assert not torch.equal(want1, want2)
    with torch.no_grad():
        op.kv_b_proj.copy_(w2.to(DEV))                       # bumps the version counter
    assert torch.equal(op(*dev), want2)
    op.load_state_dict({"kv_b_proj": w1.to(DEV)})
    assert torch.equal(op(*dev), want1)
    op.kv_b_proj.data.copy_(w2.to(DEV))                      # invisible to the version counter ...
    with switch_env(MOJO_HIP_VALIDATE="1"):
        with pytest.raises(RuntimeError, match="refresh_weights"):
            op(*dev)
    op.refresh_weights()                                     # ... until the caller says so
    assert torch.equal(op(*dev), want2)


@pytest.mark.parametrize("cfg", [
    # (q_lens, cached, H, nope, rope, v, r, page): DeepSeek-V3 head dimensions, ragged lengths, cached prefixes, an empty
    # sequence, a sequence shorter than a tile, lengths straddling the 64-key tile and the 128-row block
    ([130, 64, 1, 0, 200], [0, 70, 300, 0, 129], 16, 128, 64, 128, 512, 16),
    ([257], [1023], 8, 128, 64, 128, 512, 64),
    ([48, 31], [0, 0], 8, 64, 32, 64, 32, 32),
    ([100, 29], [33, 7], 16, 96, 32, 128, 64, 16),
], ids=["DSV3_RAGGED", "DSV3_LONG_PREFIX", "REF_DIMS", "MID_DIMS"])
@pytest.mark.parametrize("sink", [False, True])
def test_mla_prefill_decompressed_route(cfg, sink):
    """The non-absorbed formulation against the golden at the GQA prefill's band (2e-2): it shares the golden's rounding
    points, so this is a tight comparison on every row, not the wide band of the absorbed form."""
    q_lens, cached, h, nope, rope, vd, r, page = cfg
    kv_lens = [a + b for a, b in zip(q_lens, cached)]
    g = torch.Generator().manual_seed(sum(q_lens) + h)
    ckv, kpe, table, w, sk = make_mla(kv_lens, h, nope, rope, vd, r, page, sink, seed=h + 3, wscale=0.2 if r <= 64 else 0.05)
    q = torch.randn(sum(q_lens), h, nope + rope, generator=g).to(torch.bfloat16)
    assert prefill_route(h, nope, rope, vd, q.shape[0]) == "decompress"
    ref = build("MojoPagedPrefillMLA", h, nope, rope, vd, r, sink, w, sk, "cpu", is_causal=True)
    op = build("MojoPagedPrefillMLA", h, nope, rope, vd, r, sink, w, sk, DEV, is_causal=True)
    want = ref(q, ckv, kpe, cu(q_lens), table, cu_total_seq_lens=cu(kv_lens))
    got = to_cpu(op(q.to(DEV), ckv.to(DEV), kpe.to(DEV), cu(q_lens).to(DEV), table.to(DEV), cu_total_seq_lens=cu(kv_lens).to(DEV)))
    torch.testing.assert_close(got.float(), want.float(), atol=2e-2, rtol=2e-2)
    if all(c == 0 for c in cached):                      # kv = q lengths: the `cu_total_seq_lens=None` calling form
        got2 = to_cpu(op(q.to(DEV), ckv.to(DEV), kpe.to(DEV), cu(q_lens).to(DEV), table.to(DEV)))
        assert torch.equal(got, got2)
    # the batch walked in slices of sequences (a byte budget for the decompressed image that only fits two of them), and
    # the same with the per-sequence bound passed by the caller instead of taken from the table width: same numbers
    kv_cols_bytes = h * (nope + vd) * 2
    with switch_env(MOJO_HIP_MLA_PREFILL_BYTES=2 * table.shape[1] * page * kv_cols_bytes):
        sliced = to_cpu(op(q.to(DEV), ckv.to(DEV), kpe.to(DEV), cu(q_lens).to(DEV), table.to(DEV), cu_total_seq_lens=cu(kv_lens).to(DEV)))
    assert torch.equal(sliced, got)
    hinted = to_cpu(op(q.to(DEV), ckv.to(DEV), kpe.to(DEV), cu(q_lens).to(DEV), table.to(DEV), cu_total_seq_lens=cu(kv_lens).to(DEV),
                       max_total_seq_len=max(kv_lens)))
    assert torch.equal(hinted, got)
    # dispatch slots per (sequence, head) padded to an odd count (engine rotation) or not: every block visited once either way
    from mojo_opset_amd.backends.hip import lib as _L
    if _L.built_with_experiments():                      # (the placement switch exists in experiments builds only)
        with switch_env(MOJO_HIP_MLA_PREFILL_ODD_SLOTS="0"):
            even = to_cpu(op(q.to(DEV), ckv.to(DEV), kpe.to(DEV), cu(q_lens).to(DEV), table.to(DEV), cu_total_seq_lens=cu(kv_lens).to(DEV)))
        assert torch.equal(even, got)
    # padding rows behind the last sequence read as zeros (the golden's `torch.zeros` output, :393)
    pad = torch.randn(5, h, nope + rope, generator=g).to(torch.bfloat16)
    got3 = to_cpu(op(torch.cat([q, pad]).to(DEV), ckv.to(DEV), kpe.to(DEV), cu(q_lens).to(DEV), table.to(DEV),
                     cu_total_seq_lens=cu(kv_lens).to(DEV)))
    assert torch.equal(got3[: q.shape[0]], got) and torch.count_nonzero(got3[q.shape[0]:]) == 0


def test_mla_prefill_head_group_pipeline_gives_the_same_bits(monkeypatch):
    """`op.prefill_head_groups`: the decompression of head group g + 1 on a side stream beside the attention of group g
    (opt-in; `mojo_hip_mla_prefill_attn` takes a head range, the GEMM writes a column block of the image).  Per-head work is
    independent: every group count gives the bits of the single launch, also when the batch is walked in slices."""
    h, nope, rope, vd, r, page = 64, 128, 64, 128, 512, 16
    q_lens, cached = [300, 512, 77], [0, 640, 100]
    kv_lens = [a + b for a, b in zip(q_lens, cached)]
    g = torch.Generator().manual_seed(11)
    ckv, kpe, table, w, _ = make_mla(kv_lens, h, nope, rope, vd, r, page, seed=12, wscale=0.05)
    q = torch.randn(sum(q_lens), h, nope + rope, generator=g).to(torch.bfloat16)
    op = build("MojoPagedPrefillMLA", h, nope, rope, vd, r, False, w, None, DEV, is_causal=True)
    args = [t.to(DEV) for t in (q, ckv, kpe, cu(q_lens), table)]
    kw = dict(cu_total_seq_lens=cu(kv_lens).to(DEV), max_total_seq_len=max(kv_lens))
    outs = {}
    for groups in ("1", "2", "4"):
        op.prefill_head_groups = int(groups)
        outs[groups] = to_cpu(op(*args, **kw))
    assert torch.equal(outs["1"], outs["2"]) and torch.equal(outs["1"], outs["4"])
    monkeypatch.setenv("MOJO_HIP_MLA_PREFILL_BYTES", str(max(kv_lens) * h * (nope + vd) * 2))      # one sequence per slice
    assert torch.equal(to_cpu(op(*args, **kw)), outs["1"])
