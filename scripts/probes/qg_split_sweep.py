"""Sweep the K split of the decode-sized QuantGemm (MOJO_HIP_QGEMM_SPLITK, read per call) under graph replay, weight copies in rotation."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mojo_opset_amd as mo  # noqa: E402
from benchmarks.extras import _time_graph  # noqa: E402

dev = torch.device("cuda", 0)
for m, k, n in ((32, 7168, 4096), (128, 7168, 4096), (64, 7168, 4096), (32, 18432, 7168), (16, 4096, 7168)):
    ops = []
    for _ in range(max(2, min(8, int(400e6 // (n * k))))):
        op = mo.MojoQuantGemm.get_backend_impl("hip", strict=True)(k, n, trans_weight=True, device=dev)
        op.weight.copy_(torch.randint(-127, 128, (n, k), dtype=torch.int8, device=dev))
        op.weight_scale.fill_(0.01)
        ops.append(op)
    x = torch.randint(-127, 128, (m, k), dtype=torch.int8, device=dev)
    s = torch.rand(m, device=dev)
    row = []
    for sk in (0, 1, 2, 3, 4, 5, 6, 7, 8, 12, 16):
        if sk:
            os.environ["MOJO_HIP_QGEMM_SPLITK"] = str(sk)
        else:
            os.environ.pop("MOJO_HIP_QGEMM_SPLITK", None)
        if sk > k // 256 // 2:
            continue
        it = [0]

        def fn():
            it[0] += 1
            return ops[it[0] % len(ops)](x, s)

        fn()
        t = _time_graph(fn, reps=len(ops) * 4, replays=5)
        row.append(f"sk={sk or 'auto'}:{t * 1e6:5.1f}")
    os.environ.pop("MOJO_HIP_QGEMM_SPLITK", None)
    print(f"int8 {m}x{k}x{n}: " + "  ".join(row), flush=True)
