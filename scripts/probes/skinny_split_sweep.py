"""Sweep the split-K factor of the decode-sized dense GEMM (gemm_skinny.hip) over the projections of one Llama-3-8B layer
and the bench's decode shapes: time under graph replay per forced split, next to the split the library picks by itself.

    python3 scripts/probes/skinny_split_sweep.py          (GPU box)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__  # noqa: F401,E402  (puts the repo on the path, checks the library is built)
from benchmarks.extras import _time_graph  # noqa: E402
from mojo_opset_amd.backends.hip.operators.gemm import dense_gemm  # noqa: E402

dev = torch.device("cuda:0")
shapes = [(64, 4096, 6144), (64, 4096, 4096), (64, 4096, 28672), (64, 14336, 4096), (64, 8192, 8192), (1, 8192, 8192),
          (16, 4096, 6144), (128, 8192, 8192), (32, 7168, 2048), (64, 2048, 7168)]
for m, k, n in shapes:
    x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    # four weight copies used in rotation: 4 x >= 33 MB, so a replay does not find its weights in the MALL
    ws = [torch.randn(n, k, device=dev, dtype=torch.bfloat16) for _ in range(max(2, min(8, int(600e6 // (n * k * 2)))))]
    ref = None
    row = []
    for sk in (0, 1, 2, 3, 4, 5, 6, 8, 10, 12, 16):
        if sk > 0:
            os.environ["MOJO_HIP_GEMM_SKINNY_SPLITK"] = str(sk)
        else:
            os.environ.pop("MOJO_HIP_GEMM_SKINNY_SPLITK", None)
        if sk > max(1, k // 128 // 2):
            continue
        it = [0]

        def fn():
            it[0] += 1
            return dense_gemm(x, ws[it[0] % len(ws)], None, False)

        out = dense_gemm(x, ws[0], None, False)
        if ref is None:
            ref = out.float()
        err = (out.float() - ref).abs().max().item()
        t = _time_graph(fn, reps=len(ws) * 4, replays=5)
        row.append(f"sk={sk or 'auto'}: {t * 1e6:6.1f} us {n * k * 2 / t / 1e12:4.2f} TB/s" + (f" (max diff {err:.3g})" if err > 0.5 else ""))
    print(f"M={m} K={k} N={n}\n   " + "\n   ".join(row), flush=True)
    del ws, x
    torch.cuda.empty_cache()
os.environ.pop("MOJO_HIP_GEMM_SKINNY_SPLITK", None)

# calibration: what the vendor library (torch.nn.functional.linear -> hipBLASLt) takes for the same products; not a product path
import torch.nn.functional as F  # noqa: E402
print("calibration: torch F.linear (hipBLASLt) under the same replay timing")
for m, k, n in shapes:
    x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    ws = [torch.randn(n, k, device=dev, dtype=torch.bfloat16) for _ in range(max(2, min(8, int(600e6 // (n * k * 2)))))]
    it = [0]

    def fn():
        it[0] += 1
        return F.linear(x, ws[it[0] % len(ws)])

    t = _time_graph(fn, reps=len(ws) * 4, replays=5)
    print(f"M={m} K={k} N={n}: {t * 1e6:6.1f} us {n * k * 2 / t / 1e12:4.2f} TB/s", flush=True)
    del ws, x
    torch.cuda.empty_cache()
