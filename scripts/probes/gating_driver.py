"""A few router calls at one decode-sized shape, for a kernel trace:  python3 scripts/probes/gating_driver.py T E k H"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__  # noqa
from benchmarks.extras import hip
t, e, k, h = (int(v) for v in sys.argv[1:5])
dev = torch.device("cuda:0")
x = torch.rand(t, h, device=dev, dtype=torch.bfloat16)
g = hip("MojoMoEGating")(hidden_size=h, num_experts=e, top_k=k).to(dev)
with torch.no_grad():
    g.gate_weight.copy_(torch.randn(h, e) * 0.02)
for _ in range(40):
    g(x)
torch.cuda.synchronize()
