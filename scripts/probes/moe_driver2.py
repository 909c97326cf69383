"""MoE dispatch + combine at many experts, for a kernel trace:  python3 scripts/probes/moe_driver2.py T E k H"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__  # noqa
from benchmarks.extras import hip
t, e, k, h = (int(v) for v in sys.argv[1:5])
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.rand(t, h, device=dev, dtype=torch.bfloat16)
gates, ids = torch.topk(torch.rand(t, e, device=dev), k, dim=-1)
ids = ids.to(torch.int32).contiguous(); gates = gates.contiguous()
dd, cd = hip("MojoMoEDispatch")(num_experts=e), hip("MojoMoECombine")()
buf = torch.empty_like(x)
for _ in range(12):
    rows, cnt, sg, tok = dd(x, gates, ids)
    cd(buf, rows, sg, tok)
torch.cuda.synchronize()
