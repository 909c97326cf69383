"""MojoPagedPrefillMLA, decompressed route only, 4 x 512 new tokens (+ CACHED cached ones): a profiling target."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.extras import _time, hip  # noqa: E402

dev = torch.device("cuda", 0)
h, nope, rope, vd, r, page = 128, 128, 64, 128, 512, 16
op = hip("MojoPagedPrefillMLA")(h, nope, rope, vd, r).to(torch.bfloat16).to(dev)
with torch.no_grad():
    op.kv_b_proj.copy_(torch.randn_like(op.kv_b_proj) * 0.02)
q_lens, cached = [512] * 4, [int(os.environ.get("CACHED", "0"))] * 4
kv = [a + b for a, b in zip(q_lens, cached)]
need = [(n + page - 1) // page for n in kv]
total = sum(need) + 4
ckv = torch.randn(total, 1, page, r, device=dev, dtype=torch.bfloat16)
kpe = torch.randn(total, 1, page, rope, device=dev, dtype=torch.bfloat16)
table = torch.randperm(total, dtype=torch.int32)[: sum(need)].view(len(kv), need[0]).to(dev)
cu = lambda l: torch.tensor([0] + torch.tensor(l).cumsum(0).tolist(), dtype=torch.int32, device=dev)  # noqa: E731
cu_q, cu_kv = cu(q_lens), cu(kv)
q = torch.randn(sum(q_lens), h, nope + rope, device=dev, dtype=torch.bfloat16)
t = _time(lambda: op(q, ckv, kpe, cu_q, table, cu_total_seq_lens=cu_kv, max_total_seq_len=max(kv)), 20, 3)
print(f"cached {cached[0]}: {t * 1e6:.1f} us")
