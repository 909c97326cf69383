"""Round 5: the 128 x 128-tile dense GEMM (csrc/gemm_tile128.hip) against the 256 x 256 kernel (+ split-K where it splits) and
hipBLASLt (torch F.linear / x @ w) on mid-size M; bf16, random data; device time (ten calls per HIP graph, sustained medians):
the eager figures of the first runs were host-bound below ~25 us."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.extras import _time_graph
from mojo_opset_amd import switches
from mojo_opset_amd.backends.hip import lib as L
from mojo_opset_amd.backends.hip.operators.gemm import dense_gemm
dev = torch.device("cuda", 0)
shapes = ((4096, 4096), (4096, 6144), (4096, 28672), (14336, 4096), (3584, 8192), (8192, 8192), (1024, 8192), (8192, 1024))
ms = (192, 256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096)


TRANS = len(sys.argv) > 1 and sys.argv[1] == "KN"          # [K, N] weights (`x @ w`) instead of F.linear's [N, K]


def leg(x, w, value):
    os.environ["MOJO_HIP_GEMM_TILE128"] = value
    switches.reload()
    t = _time_graph(lambda: dense_gemm(x, w, None, TRANS), reps=10)
    return t, L.last_launch()


for k, n in shapes:
    w = torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.02
    if TRANS:
        w = w.t().contiguous()
    row = {}
    for m in ms:
        x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
        t256, f256 = leg(x, w, "0")
        t128, f128 = leg(x, w, "1")
        t_lib = _time_graph((lambda: x @ w) if TRANS else (lambda: torch.nn.functional.linear(x, w)), reps=10)
        tf = lambda t: round(2.0 * m * k * n / t / 1e12)
        row[m] = {"t256_us": round(t256 * 1e6, 1), "t128_us": round(t128 * 1e6, 1), "hipblaslt_us": round(t_lib * 1e6, 1),
                  "tf256": tf(t256), "tf128": tf(t128), "tf_lib": tf(t_lib), "f256": f256, "f128": f128,
                  "tiles256": -(-m // 256) * -(-n // 256), "tiles128": -(-m // 128) * -(-n // 128)}
    print(json.dumps({f"K{k}_N{n}": row}), flush=True)
