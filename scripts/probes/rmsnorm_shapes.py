"""ResidualAddRMSNorm / RMSNorm / SwiGLU bandwidth over hidden sizes (the bench line is hidden 4096 only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.extras import _time, hip  # noqa: E402

dev = torch.device("cuda", 0)
for rows, d in ((16384, 4096), (16384, 5120), (8192, 7168), (8192, 8192), (4096, 16384), (8192, 7338)):
    x, r = torch.randn(rows, d, device=dev, dtype=torch.bfloat16), torch.randn(rows, d, device=dev, dtype=torch.bfloat16)
    norm = hip("MojoResidualAddRMSNorm")(d, 1e-5, "pre", dtype=torch.bfloat16, device=dev)
    t = _time(lambda: norm(x, r), 50, 10)
    plain = hip("MojoRMSNorm")(d, 1e-5, dtype=torch.bfloat16, device=dev)
    t2 = _time(lambda: plain(x), 50, 10)
    print(f"{rows:6d} x {d:6d}  residual_add_rmsnorm {t * 1e6:7.1f} us {4 * rows * d * 2 / t / 1e9:6.0f} GB/s   rmsnorm {t2 * 1e6:7.1f} us {2 * rows * d * 2 / t2 / 1e9:6.0f} GB/s")
