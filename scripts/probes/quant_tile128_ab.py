"""Round 5: QuantGemm (int8 / fp8, [N,K] weights = trans_weight=True) at mid-size M: the 128-row-tile kernel (gemm_tile128_core.h)
against the 256 x 256 kernel (+ split-K slabs where it splits), forced in one process; bf16 output, random data."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.extras import _time_graph
from mojo_opset_amd import switches
from mojo_opset_amd.backends.hip import lib as L
from mojo_opset_amd.backends.hip.operators.gemm import HIPQuantGemm
dev = torch.device("cuda", 0)
KN = len(sys.argv) > 1 and sys.argv[1] == "KN"            # the operator's default (K, N) weight layout (int8 only on the 128-row tiles)
shapes = ((4096, 4096), (7168, 4096), (7168, 1536), (4096, 6144), (18432, 7168), (2048, 7168), (1024, 8192), (8192, 8192))
ms = (192, 256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096)
for qd in ((torch.int8,) if KN else (torch.int8, torch.float8_e4m3fn)):
    for k, n in shapes:
        op = HIPQuantGemm(k, n, output_dtype=torch.bfloat16, trans_weight=not KN, quant_dtype=qd, weight_dtype=qd, device=dev)
        w_nk = torch.randint(-127, 128, (n, k), dtype=torch.int8, device=dev) if qd == torch.int8 else torch.randn(n, k, device=dev).to(qd)
        op.weight.copy_(w_nk.t() if KN else w_nk)
        op.weight_scale.fill_(0.01)
        row = {}
        for m in ms:
            x = torch.randint(-127, 128, (m, k), dtype=torch.int8, device=dev) if qd == torch.int8 else torch.randn(m, k, device=dev).to(qd)
            sc = torch.rand(m, device=dev)
            res = {}
            for leg in ("0", "1"):
                os.environ["MOJO_HIP_GEMM_TILE128"] = leg
                switches.reload()
                t = _time_graph(lambda: op(x, sc), reps=10)
                res[leg] = (t, L.last_launch())
            row[m] = {"t256_us": round(res["0"][0] * 1e6, 1), "t128_us": round(res["1"][0] * 1e6, 1), "f256": res["0"][1], "f128": res["1"][1],
                      "tops128": round(2.0 * m * k * n / res["1"][0] / 1e12), "tops256": round(2.0 * m * k * n / res["0"][0] / 1e12),
                      "tiles256": -(-m // 256) * -(-n // 256), "tiles128": -(-m // 128) * -(-n // 128)}
        print(json.dumps({f"{'i8' if qd == torch.int8 else 'f8'}_K{k}_N{n}": row}), flush=True)
        del op
