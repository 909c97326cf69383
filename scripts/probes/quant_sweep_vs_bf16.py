"""Round 5: MojoQuantGemm (int8, bf16 output) over every regime of M and both weight layouts against this backend's own bf16
product of the same shape (the 8-bit product moves half the weight bytes and multiplies at twice the rate: a ratio above 1
is a fall-off); device times (HIP graphs)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.extras import _time_graph
from mojo_opset_amd.backends.hip import lib as L
from mojo_opset_amd.backends.hip.operators.gemm import HIPQuantGemm, dense_gemm
dev = torch.device("cuda", 0)
shapes = ((4096, 4096), (7168, 4096), (4096, 14336), (18432, 7168), (7168, 2048), (2048, 7168), (1024, 8192), (8192, 1024))
ms = (1, 4, 8, 32, 64, 96, 128, 160, 256, 512, 1024, 2048, 4096, 8192)
for k, n in shapes:
    for trans in (True, False):                       # trans_weight=True: (N, K) weights
        op = HIPQuantGemm(k, n, output_dtype=torch.bfloat16, trans_weight=trans, device=dev)
        w_nk = torch.randint(-127, 128, (n, k), dtype=torch.int8, device=dev)
        op.weight.copy_(w_nk if trans else w_nk.t())
        op.weight_scale.fill_(0.01)
        wb = (torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.02)
        wb = wb if trans else wb.t().contiguous()
        for m in ms:
            x = torch.randint(-127, 128, (m, k), dtype=torch.int8, device=dev)
            sc = torch.rand(m, device=dev)
            xb = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
            reps = 10 if m * k * n < 2 ** 36 else 3
            t = _time_graph(lambda: op(x, sc), reps=reps)
            form = L.last_launch()
            tb = _time_graph(lambda: dense_gemm(xb, wb, None, not trans), reps=reps)
            print(json.dumps({"m": m, "k": k, "n": n, "layout": "NK" if trans else "KN", "int8_us": round(t * 1e6, 1), "bf16_us": round(tb * 1e6, 1),
                              "int8_vs_bf16": round(t / tb, 2), "form": form, "bf16_form": L.last_launch()}), flush=True)
        del op
