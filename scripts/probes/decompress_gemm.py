"""The MLA prefill decompression GEMM on its own: [T_kv, 512] x [H * 256, 512]^T, one group (DeepSeek-V3 dims)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.extras import _time, hip  # noqa: E402

dev = torch.device("cuda", 0)
for m in (2048, 4096, 10240):
    k, n = 512, 128 * 256
    x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    w = torch.randn(1, n, k, device=dev, dtype=torch.bfloat16) * 0.05
    op = hip("MojoGroupGemm")(w, True)
    counts = torch.tensor([m], dtype=torch.int32, device=dev)
    t = _time(lambda: op(x, counts), 20, 5)
    tt = _time(lambda: torch.matmul(x, w[0].t()), 20, 5)
    fl = 2.0 * m * k * n
    print(f"M {m:6d}: grouped GEMM {t * 1e6:7.1f} us = {fl / t / 1e12:6.0f} TF/s, output {m * n * 2 / t / 1e9:6.0f} GB/s written | torch.matmul {tt * 1e6:7.1f} us = {fl / tt / 1e12:6.0f} TF/s", flush=True)
