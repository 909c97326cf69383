"""The MoE layer at decode (64 tokens, top-8 of 64 experts, hidden 4096, inter 2048) a few times, for a kernel trace."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__  # noqa
from benchmarks.extras import hip
dev = torch.device("cuda:0")
torch.manual_seed(20260716)
td, ed, kd, hd, idm = 64, 64, 8, 4096, 2048
xd = torch.rand(td, hd, device=dev, dtype=torch.bfloat16)
gd = hip("MojoMoEGating")(hidden_size=hd, num_experts=ed, top_k=kd).to(dev)
exd = hip("MojoExperts")(num_experts=ed, hidden_size=hd, intermediate_size=idm).to(torch.bfloat16).to(dev)
with torch.no_grad():
    gd.gate_weight.copy_(torch.randn(hd, ed) * 0.02)
    exd.up_proj_weight.normal_(std=0.02)
    exd.down_proj_weight.normal_(std=0.02)
dd, cd = hip("MojoMoEDispatch")(num_experts=ed), hip("MojoMoECombine")()
bufd = torch.empty_like(xd)
for _ in range(30):
    i2, g2 = gd(xd)
    a, c, b, d = dd(xd, g2, i2)
    cd(bufd, exd(a, c), b, d)
torch.cuda.synchronize()
