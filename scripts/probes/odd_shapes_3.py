"""Third sweep: dense bf16 and int8 GEMMs at mid-sized M (between the decode-sized streaming kernels and the big-tile regime),
next to the vendor library (torch F.linear -> hipBLASLt) as calibration."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__  # noqa
from benchmarks.extras import _time_graph, hip
from mojo_opset_amd.backends.hip.operators.gemm import dense_gemm
dev = torch.device("cuda:0")
for m in (129, 192, 256, 512, 1024, 2048):
    for k, n in ((4096, 4096), (4096, 14336), (8192, 8192)):
        x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
        w = torch.randn(n, k, device=dev, dtype=torch.bfloat16)
        t = _time_graph(lambda: dense_gemm(x, w, None, False), reps=5, replays=3)
        tl = _time_graph(lambda: F.linear(x, w), reps=5, replays=3)
        fl = 2.0 * m * k * n
        print(f"dense bf16 M={m} K={k} N={n}: {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF | library {tl*1e6:8.1f} us {fl/tl/1e12:7.1f} TF | ratio {t/tl:4.2f}", flush=True)
        del x, w
q = hip("MojoQuantGemm")
for m in (129, 256, 512, 1024):
    for k, n in ((7168, 4096), (4096, 7168)):
        op = q(k, n).to(dev)
        with torch.no_grad():
            op.weight.copy_(torch.randint(-127, 128, tuple(op.weight.shape), dtype=torch.int8))
            op.weight_scale.copy_(torch.rand(n) * 0.01)
        xq = torch.randint(-127, 128, (m, k), dtype=torch.int8, device=dev)
        s = torch.rand(m, device=dev)
        t = _time_graph(lambda: op(xq, s), reps=5, replays=3)
        print(f"int8 M={m} K={k} N={n}: {t*1e6:8.1f} us {2.0*m*k*n/t/1e12:7.1f} TOP/s  weights {k*n/t/1e12:5.2f} TB/s", flush=True)
