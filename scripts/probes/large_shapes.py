"""Launch forms the fixed tests do not reach (large batches, very long contexts, the split + merge decode path), checked on
sampled sequences against the oracle.  Not part of the suite (minutes of CPU oracle time)."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
torch.set_num_threads(12)
from hip_utils import DEV, hip_cls, to_cpu, torch_cls  # noqa: E402
from test_hip_decode_gqa import make_decode_inputs  # noqa: E402
from test_hip_mla import build, check_mla, exact_mla, make_mla  # noqa: E402
from test_hip_prefill_gqa import make_prefill_inputs  # noqa: E402


def decode_case(batch, hq, hkv, d, page, lens, tag):
    q, k, v, lens_t, table = make_decode_inputs(batch, hq, hkv, d, max(lens), page, seed=1, lens=lens)
    op = hip_cls("MojoPagedDecodeGQA")(is_causal=True, gqa_layout="AABB")
    ref = torch_cls("MojoPagedDecodeGQA")(is_causal=True, gqa_layout="AABB")
    got = to_cpu(op(*[t.to(DEV) for t in (q, k, v, lens_t, table)], softmax_scale=1 / math.sqrt(d)))
    assert torch.isfinite(got.float()).all()
    worst = 0.0
    for i in (0, batch // 2, batch - 1):
        want = ref(q[i:i + 1], k, v, lens_t[i:i + 1], table[i:i + 1], softmax_scale=1 / math.sqrt(d))
        worst = max(worst, (got[i:i + 1].float() - want.float()).abs().max().item())
    print(f"decode {tag}: max |hip - oracle| on 3 sequences {worst:.4f}", flush=True)
    assert worst < 2e-2


decode_case(200, 32, 8, 128, 16, [1500 + 7 * i for i in range(200)], "B=200 ragged (fused, not paired)")
decode_case(4, 32, 8, 128, 16, [40000, 35000, 123, 20000], "ctx 40k (split + merge launches)")
decode_case(300, 8, 1, 128, 64, [64 + i for i in range(300)], "B=300, one kv head, page 64")
decode_case(33, 64, 8, 128, 128, [3000] * 33, "B=33 G=8 page 128")

q, k, v, cu_q, table, cu_kv, kv_lens = make_prefill_inputs([9000, 300], [0, 5000], 32, 8, 128, 16, seed=2)
op = hip_cls("MojoPagedPrefillGQA")()
got = to_cpu(op(q.to(DEV), k.to(DEV), v.to(DEV), cu_q.to(DEV), table.to(DEV), cu_total_seq_lens=cu_kv.to(DEV)))
assert torch.isfinite(got.float()).all()
want = torch_cls("MojoPagedPrefillGQA")()(q[9000:], k, v, torch.tensor([0, 300], dtype=torch.int32), table[1:2], cu_total_seq_lens=torch.tensor([0, 5300], dtype=torch.int32))
print(f"prefill 9000 + (300 over 5000 cached): second sequence max diff {(got[9000:].float() - want.float()).abs().max():.4f}", flush=True)
assert (got[9000:].float() - want.float()).abs().max() < 2e-2

b, h, nope, rope, vd, r, page = 150, 128, 128, 64, 128, 512, 16
lens = [600 + 5 * i for i in range(b)]
ckv, kpe, tbl, w, _ = make_mla(lens, h, nope, rope, vd, r, page, seed=3, wscale=0.05)
qm = torch.randn(b, h, nope + rope, generator=torch.Generator().manual_seed(3)).to(torch.bfloat16)
mop = build("MojoPagedDecodeMLA", h, nope, rope, vd, r, False, w, None, DEV)
mgot = to_cpu(mop(qm.to(DEV), ckv.to(DEV), kpe.to(DEV), torch.tensor(lens, dtype=torch.int32).to(DEV), tbl.to(DEV)))
assert torch.isfinite(mgot.float()).all()
for i in (0, b - 1):
    ex = exact_mla(qm[i:i + 1], ckv, kpe, tbl[i:i + 1], w, None, h, nope, rope, vd, r, [lens[i]])
    print(f"mla decode B=150 sequence {i}: max |hip - exact| {(mgot[i:i + 1].double() - ex.double()).abs().max():.4f}", flush=True)
    assert (mgot[i:i + 1].double() - ex.double()).abs().max() < 1e-2
print("large shapes ok")
