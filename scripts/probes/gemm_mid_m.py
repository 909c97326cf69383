"""Round 5: mid-size M dense products (chunked prefill of 256..2048 tokens; config 4 at M 1024): this backend's dense_gemm against
hipBLASLt (torch.matmul) on the same box.  [N, K] weights (F.linear layout), bf16."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.extras import _time
from mojo_opset_amd.backends.hip import lib as L
from mojo_opset_amd.backends.hip.operators.gemm import dense_gemm
dev = torch.device("cuda", 0)
for k, n in ((4096, 4096), (4096, 6144), (4096, 28672), (14336, 4096), (3584, 8192), (8192, 8192)):
    w = torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.02
    row = {}
    for m in (256, 512, 1024, 2048, 4096):
        x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
        t_mine = _time(lambda: dense_gemm(x, w, None, False), 20, 5, repeats=3)
        form = L.last_launch()
        t_lib = _time(lambda: torch.nn.functional.linear(x, w), 20, 5, repeats=3)
        row[m] = {"mine_us": round(t_mine * 1e6, 1), "hipblaslt_us": round(t_lib * 1e6, 1), "mine_tflops": round(2.0 * m * k * n / t_mine / 1e12),
                  "hipblaslt_tflops": round(2.0 * m * k * n / t_lib / 1e12), "form": form}
    print(json.dumps({f"K{k}_N{n}": row}), flush=True)
