"""A few launches of the decode-sized dense GEMM for a profiler pass:  python3 scripts/probes/skinny_driver.py M K N"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__  # noqa
from mojo_opset_amd.backends.hip.operators.gemm import dense_gemm
m, k, n = (int(v) for v in sys.argv[1:4])
dev = torch.device("cuda:0")
x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
ws = [torch.randn(n, k, device=dev, dtype=torch.bfloat16) for _ in range(max(2, min(8, int(600e6 // (n * k * 2)))))]
for i in range(12):
    dense_gemm(x, ws[i % len(ws)], None, False)
torch.cuda.synchronize()
