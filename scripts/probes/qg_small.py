"""Small-M QuantGemm timing target for rocprofv3 (kernel durations of the skinny kernel and the finalize)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mojo_opset_amd as mo  # noqa: E402

dev = torch.device("cuda", 0)
m, k, n = int(os.environ.get("QG_M", 32)), 7168, 4096
op = mo.MojoQuantGemm.get_backend_impl("hip", strict=True)(k, n, trans_weight=True, device=dev)
op.weight.copy_(torch.randint(-127, 128, (n, k), dtype=torch.int8, device=dev))
op.weight_scale.fill_(0.01)
x = torch.randint(-127, 128, (m, k), dtype=torch.int8, device=dev)
s = torch.rand(m, device=dev)
for _ in range(30):
    op(x, s)
torch.cuda.synchronize()
