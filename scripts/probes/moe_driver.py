"""Mixtral-routing dispatch + combine a few times, for a kernel trace:  python3 scripts/probes/moe_driver.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__  # noqa
from benchmarks.extras import hip
dev = torch.device("cuda:0")
torch.manual_seed(20260716)
t_, e_, k_, h_ = 8192, 8, 2, 4096
x = torch.rand(t_, h_, device=dev, dtype=torch.bfloat16)
gating = hip("MojoMoEGating")(hidden_size=h_, num_experts=e_, top_k=k_).to(dev)
with torch.no_grad():
    gating.gate_weight.copy_(torch.randn(h_, e_) * 0.02)
idx, gates = gating(x)
dispatch, combine = hip("MojoMoEDispatch")(num_experts=e_), hip("MojoMoECombine")()
buf = torch.empty_like(x)
for _ in range(12):
    sh, counts, sg, tok = dispatch(x, gates, idx)
    combine(buf, sh, sg, tok)
torch.cuda.synchronize()
