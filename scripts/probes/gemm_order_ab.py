"""GroupGemm tile-order experiment, interleaved rounds in ONE process:
    python3 scripts/probes/gemm_order_ab.py            # TFLOP/s per MOJO_HIP_GEMM_ORDER in {0, 1, 2}
    python3 scripts/probes/gemm_order_ab.py one <n>    # a few launches of one order (for a counter pass)"""
import json, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__  # noqa
from benchmarks.extras import hip, _time
dev = torch.device("cuda:0")
m, k, n, groups = 16384, 4096, 28672, 8
x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
w = torch.randn(groups, k, n, device=dev, dtype=torch.bfloat16)
counts = torch.full((groups,), m // groups, dtype=torch.int32, device=dev)
op = hip("MojoGroupGemm")(w, False)
if len(sys.argv) > 2 and sys.argv[1] == "one":
    os.environ["MOJO_HIP_GEMM_ORDER"] = sys.argv[2]
    for _ in range(12):
        op(x, counts)
    torch.cuda.synchronize()
    sys.exit(0)
res = {o: [] for o in "012"}
for rnd in range(5):
    for o in "012":
        os.environ["MOJO_HIP_GEMM_ORDER"] = o
        res[o].append(2.0 * m * k * n / _time(lambda: op(x, counts), 10, 2, repeats=3) / 1e12)
print(json.dumps({"case": "mixtral_up_16384x4096x28672_G8_KN", **{"order_" + o: {"median_tflops": round(statistics.median(v), 1), "min": round(min(v), 1), "max": round(max(v), 1)} for o, v in res.items()}}))
