"""Decode-sized MoE layer (64 tokens, top-8 of 64 experts): fused gemm+swiglu epilogue vs the two-kernel path, A/B in one process."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.extras import _time_graph, hip  # noqa: E402

dev = torch.device("cuda", 0)
td, ed, kd, hd, idm = 64, 64, 8, 4096, 2048
gd = hip("MojoMoEGating")(hidden_size=hd, num_experts=ed, top_k=kd).to(dev)
exd = hip("MojoExperts")(num_experts=ed, hidden_size=hd, intermediate_size=idm).to(torch.bfloat16).to(dev)
with torch.no_grad():
    for p in exd.parameters():
        p.normal_(std=0.02)
    gd.gate_weight.normal_(std=0.02)
dd, cd = hip("MojoMoEDispatch")(num_experts=ed), hip("MojoMoECombine")()
x = torch.randn(td, hd, device=dev, dtype=torch.bfloat16)
buf = torch.zeros_like(x)


def layer():
    idx, g = gd(x)
    a, c, b, d = dd(x, g, idx)
    return cd(buf, exd(a, c), b, d)


nbytes = ed * 3 * hd * idm * 2
for rep in range(3):
    for fused in ("1", "0"):
        os.environ["MOJO_HIP_EXPERTS_FUSED"] = fused
        t = _time_graph(layer, reps=5, replays=5)
        print(f"fused={fused}  {t * 1e6:7.1f} us  {nbytes / t / 1e9:6.0f} GB/s")
