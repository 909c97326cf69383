"""Decode-sized dense bf16 GEMM ([N,K] weights) under graph replay: skinny path vs the 256-tile path."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.extras import _time_graph  # noqa: E402
from mojo_opset_amd.backends.hip.operators.gemm import dense_gemm  # noqa: E402

dev = torch.device("cuda", 0)
for m, k, n in ((1, 8192, 8192), (16, 8192, 8192), (64, 8192, 8192), (128, 8192, 8192), (64, 4096, 14336), (64, 14336, 4096)):
    x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    w = torch.randn(n, k, device=dev, dtype=torch.bfloat16)
    t = _time_graph(lambda: dense_gemm(x, w, None, False))
    print(f"M={m} K={k} N={n}: {t * 1e6:.1f} us  {n * k * 2 / t / 1e12:.2f} TB/s weight stream  {2.0 * m * k * n / t / 1e12:.0f} TF", flush=True)
