"""Round 5: gate|up projection + SwiGLU at prefill-sized M: the 256 x 256 kernel's fused epilogue (one group) against the product +
`mojo_hip_swiglu_rows` (MOJO_HIP_GEMM_SKINNY=27); bf16, device time (HIP graphs)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.extras import _time_graph
from mojo_opset_amd import switches
from mojo_opset_amd.backends.hip import lib as L
from mojo_opset_amd.backends.hip.operators.gemm import dense_gemm_swiglu
dev = torch.device("cuda", 0)
for k, inter in ((4096, 14336), (8192, 28672), (5120, 13824), (2048, 8192)):
    w = torch.randn(2 * inter, k, device=dev, dtype=torch.bfloat16) * 0.02
    row = {}
    for m in (256, 512, 1024, 2048, 4096, 8192):
        x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
        res = {}
        for leg, mask in (("fused", "31"), ("unfused", "27")):
            os.environ["MOJO_HIP_GEMM_SKINNY"] = mask
            switches.reload()
            L.launch_history(clear=True)
            dense_gemm_swiglu(x, w)
            form = L.launch_history()
            res[leg] = (_time_graph(lambda: dense_gemm_swiglu(x, w), reps=4), form)
        row[m] = {"fused_us": round(res["fused"][0] * 1e6, 1), "unfused_us": round(res["unfused"][0] * 1e6, 1), "fused_form": res["fused"][1], "unfused_form": res["unfused"][1],
                  "glu_tiles": -(-m // 256) * (inter // 128)}
    print(json.dumps({f"K{k}_I{inter}": row}), flush=True)
