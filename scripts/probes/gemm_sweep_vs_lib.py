"""Round 5: dense bf16 products over every regime of M (decode-sized to prefill-sized), both weight layouts, with and without a
bias: this backend's `mojo_hip_gemm` (kernel form recorded) against hipBLASLt on the same box; device times (HIP graphs).  Lines
with time_vs_lib > 1.25 are the fall-offs to look at."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.extras import _time_graph
from mojo_opset_amd.backends.hip import lib as L
from mojo_opset_amd.backends.hip.operators.gemm import dense_gemm
dev = torch.device("cuda", 0)
shapes = ((4096, 4096), (4096, 14336), (14336, 4096), (8192, 1024), (1024, 8192), (5120, 5120), (4096, 128256), (7168, 2048), (2048, 7168), (3584, 8192))
ms = (1, 8, 32, 64, 96, 128, 160, 200, 256, 512, 1024, 2048, 4096, 8192, 16384)
for k, n in shapes:
    for layout in ("NK", "KN"):
        w = torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.02
        trans = layout == "KN"
        if trans:
            w = w.t().contiguous()
        b = torch.randn(n, device=dev, dtype=torch.bfloat16)
        for m in ms:
            if m * n > 2 ** 29:
                continue
            x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
            for bias in ((None, b) if m in (64, 1024, 8192) else (None,)):
                reps = 10 if m * k * n < 2 ** 36 else 3
                t = _time_graph(lambda: dense_gemm(x, w, bias, trans), reps=reps)
                form = L.last_launch()
                if trans:
                    t_lib = _time_graph((lambda: x @ w) if bias is None else (lambda: torch.addmm(bias, x, w)), reps=reps)
                else:
                    t_lib = _time_graph(lambda: torch.nn.functional.linear(x, w, bias), reps=reps)
                print(json.dumps({"m": m, "k": k, "n": n, "layout": layout, "bias": bias is not None, "us": round(t * 1e6, 1), "lib_us": round(t_lib * 1e6, 1),
                                  "time_vs_lib": round(t / t_lib, 2), "form": form}), flush=True)
