"""A/B in one process: decode-sized QuantGemm with the round-3 rule (four-wave workgroups, MOJO_HIP_QGEMM_WAVES=4) and with the
balanced (split, waves) plan of round 4 (unset), graph replay, weight copies in rotation."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mojo_opset_amd as mo  # noqa: E402
from benchmarks.extras import _time_graph  # noqa: E402

dev = torch.device("cuda", 0)
for dt, lo, hi in ((torch.int8, -127, 128),):
    for m, k, n in ((32, 18432, 7168), (64, 18432, 7168), (128, 18432, 7168), (32, 7168, 4096), (128, 7168, 4096), (32, 7168, 36864), (32, 7168, 2112), (16, 4096, 7168), (64, 7168, 18432)):
        ops = []
        for _ in range(max(2, min(6, int(400e6 // (n * k))))):
            op = mo.MojoQuantGemm.get_backend_impl("hip", strict=True)(k, n, trans_weight=True, device=dev)
            op.weight.copy_(torch.randint(lo, hi, (n, k), dtype=dt, device=dev))
            op.weight_scale.fill_(0.01)
            ops.append(op)
        x = torch.randint(lo, hi, (m, k), dtype=dt, device=dev)
        s = torch.rand(m, device=dev)
        row, outs = [], {}
        for rep in range(2):
            for w in ("4", None):
                if w:
                    os.environ["MOJO_HIP_QGEMM_WAVES"] = w
                else:
                    os.environ.pop("MOJO_HIP_QGEMM_WAVES", None)
                it = [0]

                def fn():
                    it[0] += 1
                    return ops[it[0] % len(ops)](x, s)

                outs[w] = ops[0](x, s).clone()
                t = _time_graph(fn, reps=len(ops) * 4, replays=5)
                row.append(f"waves={w or 'auto'}:{t * 1e6:6.1f}")
        assert torch.equal(outs["4"], outs[None])
        print(f"int8 {m}x{k}x{n}: " + "  ".join(row), flush=True)
os.environ.pop("MOJO_HIP_QGEMM_WAVES", None)
