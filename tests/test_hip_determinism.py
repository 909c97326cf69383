"""Run-to-run bit stability of the matrix-core kernels.  A vector instruction that reads an MFMA result too early, a
counted wait that is one short or a missing barrier does not have to break parity: it can show up only as outputs that
differ in the last bit from one launch to the next (found that way in the MLA latent kernels).  Every kernel
here is launched 12 times on the same inputs — sizes that fill the chip, so waves contend for the pipes — and must return
identical bits."""
import math

import pytest
import torch

from hip_utils import DEV, hip_cls, skip_unless_experiments_build
from test_hip_decode_gqa import make_decode_inputs
from test_hip_mla import build, cu, make_mla
from test_hip_prefill_gqa import make_prefill_inputs

pytestmark = pytest.mark.gpu
RUNS = 12


def _stable(fn):
    first = fn()
    for _ in range(RUNS - 1):
        again = fn()
        if isinstance(first, (tuple, list)):
            assert all(torch.equal(a, b) for a, b in zip(first, again))
        else:
            assert torch.equal(first, again)


def test_decode_gqa_is_bit_stable():
    batch, hq, hkv, d, page = 64, 32, 8, 128, 16
    g = torch.Generator().manual_seed(1)
    lens = torch.randint(700, 2049, (batch,), generator=g).tolist()
    q, k, v, lens_t, table = make_decode_inputs(batch, hq, hkv, d, 2048, page, seed=3, lens=lens)
    op = hip_cls("MojoPagedDecodeGQA")(is_causal=True, gqa_layout="AABB")
    args = [t.to(DEV) for t in (q, k, v, lens_t, table)]
    _stable(lambda: op(*args, softmax_scale=1.0 / math.sqrt(d)))


@pytest.mark.parametrize("q_lens,cached", [([700, 1024, 33, 512, 900, 640, 1000, 256], [0] * 8), ([512, 300, 640, 128], [1024, 0, 77, 2000])],
                         ids=["ragged", "cached"])
def test_prefill_gqa_is_bit_stable(q_lens, cached):
    q, k, v, cu_q, table, cu_kv, kv_lens = make_prefill_inputs(q_lens, cached, 32, 8, 128, 16, seed=5)
    op = hip_cls("MojoPagedPrefillGQA")()
    args = [t.to(DEV) for t in (q, k, v, cu_q, table)]
    kw = {} if cu_kv is None else {"cu_total_seq_lens": cu_kv.to(DEV)}
    _stable(lambda: op(*args, max_q_len=max(q_lens), max_total_seq_len=max(kv_lens), **kw))


@pytest.mark.parametrize("kernel", ["ps", "oct", "pp", "pair"])
def test_mla_decode_is_bit_stable(kernel, monkeypatch):
    if kernel in ("pp", "pair"):
        skip_unless_experiments_build()
    b, h, nope, rope, vd, r, page = 16, 128, 128, 64, 128, 512, 16
    g = torch.Generator().manual_seed(2)
    lens = torch.randint(300, 2049, (b,), generator=g).tolist()
    ckv, kpe, table, w, _ = make_mla(lens, h, nope, rope, vd, r, page, seed=7, wscale=0.05)
    q = torch.randn(b, h, nope + rope, generator=g).to(torch.bfloat16)
    op = build("MojoPagedDecodeMLA", h, nope, rope, vd, r, False, w, None, DEV)
    args = [t.to(DEV) for t in (q, ckv, kpe, torch.tensor(lens, dtype=torch.int32), table)]
    monkeypatch.setenv("MOJO_HIP_MLA_KERNEL", kernel)
    _stable(lambda: op(*args))


def test_mla_prefill_is_bit_stable():
    h, nope, rope, vd, r, page = 32, 128, 64, 128, 512, 16
    q_lens, cached = [300, 512, 77, 200], [0, 640, 100, 1000]
    kv_lens = [a + b for a, b in zip(q_lens, cached)]
    g = torch.Generator().manual_seed(4)
    ckv, kpe, table, w, _ = make_mla(kv_lens, h, nope, rope, vd, r, page, seed=9, wscale=0.05)
    q = torch.randn(sum(q_lens), h, nope + rope, generator=g).to(torch.bfloat16)
    op = build("MojoPagedPrefillMLA", h, nope, rope, vd, r, False, w, None, DEV, is_causal=True)
    args = [t.to(DEV) for t in (q, ckv, kpe, cu(q_lens), table)]
    _stable(lambda: op(*args, cu_total_seq_lens=cu(kv_lens).to(DEV)))


@pytest.mark.parametrize("trans", [False, True], ids=["KN", "NK"])
def test_group_gemm_is_bit_stable(trans):
    g = torch.Generator().manual_seed(6)
    m, k, n, groups = 4096, 2048, 4096, 8
    x = torch.randn(m, k, generator=g).to(torch.bfloat16).to(DEV)
    w = (torch.randn(groups, n, k, generator=g) if trans else torch.randn(groups, k, n, generator=g)).to(torch.bfloat16).to(DEV)
    counts = torch.tensor([700, 100, 0, 1300, 512, 17, 955, 512], dtype=torch.int32, device=DEV)
    op = hip_cls("MojoGroupGemm")(w, trans)
    _stable(lambda: op(x, counts))


@pytest.mark.parametrize("qdtype", [torch.int8, torch.float8_e4m3fn], ids=["int8", "fp8"])
@pytest.mark.parametrize("m", [32, 2048])
def test_quant_gemm_is_bit_stable(qdtype, m):
    k, n = 4096, 4096
    g = torch.Generator().manual_seed(8)
    op = hip_cls("MojoQuantGemm")(k, n, trans_weight=True, quant_dtype=qdtype, weight_dtype=qdtype, device=DEV)
    if qdtype == torch.int8:
        op.weight.copy_(torch.randint(-127, 128, (n, k), dtype=torch.int8, generator=g).to(DEV))
        x = torch.randint(-127, 128, (m, k), dtype=torch.int8, generator=g).to(DEV)
    else:
        op.weight.copy_(torch.randn(n, k, generator=g).to(DEV).to(qdtype))
        x = torch.randn(m, k, generator=g).to(DEV).to(qdtype)
    op.weight_scale.fill_(0.01)
    s = torch.rand(m, generator=g).to(DEV)
    _stable(lambda: op(x, s))
