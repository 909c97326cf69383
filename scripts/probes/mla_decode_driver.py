"""A few MLA decode calls at one shape, for a kernel trace:  python3 scripts/probes/mla_decode_driver.py B ctx"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__  # noqa
from benchmarks.extras import hip
b, ctx = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
h, nope, rope, vd, r, page = 128, 128, 64, 128, 512, 16
op = hip("MojoPagedDecodeMLA")(h, nope, rope, vd, r).to(torch.bfloat16).to(dev)
with torch.no_grad():
    op.kv_b_proj.copy_(torch.randn_like(op.kv_b_proj) * 0.02)
pages = ctx // page
total = b * pages + 4
ckv = torch.randn(total, 1, page, r, device=dev, dtype=torch.bfloat16)
kpe = torch.randn(total, 1, page, rope, device=dev, dtype=torch.bfloat16)
table = torch.randperm(total, dtype=torch.int32)[: b * pages].view(b, pages).to(dev)
lens = torch.full((b,), ctx, dtype=torch.int32, device=dev)
q = torch.randn(b, h, nope + rope, device=dev, dtype=torch.bfloat16)
for _ in range(30):
    op(q, ckv, kpe, lens, table)
torch.cuda.synchronize()
