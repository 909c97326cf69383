"""Round 5: chip-filling dense products with and without a bias ([N,K] weights = F.linear, bf16): until round 5 a bias sent the
256 x 256 kernel's tiles through the direct 8-byte stores with the bias fetched in the epilogue; now the row-staged epilogue takes it
from LDS.  MOJO_HIP_GEMM_STAGE_ROWS=0 is the old route for the bias case.  Device times (HIP graphs)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.extras import _time_graph
from mojo_opset_amd import switches
from mojo_opset_amd.backends.hip import lib as L
from mojo_opset_amd.backends.hip.operators.gemm import dense_gemm
dev = torch.device("cuda", 0)
for m, k, n in ((8192, 4096, 6144), (4096, 4096, 4096), (8192, 3584, 4608), (16384, 1024, 4096), (8192, 8192, 8192)):
    x = torch.randn(m, k, device=dev, dtype=torch.bfloat16); w = torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.02; b = torch.randn(n, device=dev, dtype=torch.bfloat16)
    t0 = _time_graph(lambda: dense_gemm(x, w, None, False), reps=6); f0 = L.last_launch()
    t1 = _time_graph(lambda: dense_gemm(x, w, b, False), reps=6); f1 = L.last_launch()
    os.environ["MOJO_HIP_GEMM_STAGE_ROWS"] = "0"; switches.reload()
    t2 = _time_graph(lambda: dense_gemm(x, w, b, False), reps=6); f2 = L.last_launch()
    os.environ.pop("MOJO_HIP_GEMM_STAGE_ROWS"); switches.reload()
    t_lib = _time_graph(lambda: torch.nn.functional.linear(x, w, b), reps=6)
    print(json.dumps({"shape": [m, k, n], "no_bias_us": round(t0 * 1e6, 1), "no_bias_form": f0, "bias_us": round(t1 * 1e6, 1), "bias_form": f1,
                      "bias_direct_stores_us": round(t2 * 1e6, 1), "bias_direct_form": f2, "hipblaslt_bias_us": round(t_lib * 1e6, 1)}), flush=True)
