"""Round 5 (reused for the ring-depth A/B: D = ring stages 4 / 5): the 128-row-tile kernel's prefetch wave (gemm_tile128_core.h: a fifth wave touching K-tile t + S + D) with COLD weights
(every call another copy, copies x bytes >= 768 MB) and with one warm weight: distance D = 0 (off) / 2 / 4 / 6 / 8 / 12, the
four-wave shape forced, unsplit and in the model's split; bf16."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.extras import _time_graph
from mojo_opset_amd import switches
from mojo_opset_amd.backends.hip import lib as L
from mojo_opset_amd.backends.hip.operators.gemm import dense_gemm
dev = torch.device("cuda", 0)
TRANS = len(sys.argv) > 1 and sys.argv[1] == "KN"
cases = ((1024, 4096, 4096), (512, 4096, 4096), (256, 4096, 4096), (1024, 14336, 4096), (512, 8192, 8192), (256, 8192, 1024),
         (128, 4096, 28672), (512, 2048, 7168), (1024, 4096, 1024), (100, 14336, 4096))
dists = (4, 5)


def leg(x, ws, **env):
    for key in ("MOJO_HIP_GEMM_TILE128", "MOJO_HIP_GEMM_SPLITK", "MOJO_HIP_GEMM_RING"):
        os.environ.pop(key, None)
    os.environ.update(env)
    switches.reload()
    i = [0]

    def fn():
        i[0] += 1
        return dense_gemm(x, ws[i[0] % len(ws)], None, TRANS)
    t = _time_graph(fn, reps=max(10, len(ws)))
    return round(t * 1e6, 1), L.last_launch()


for m, k, n in cases:
    copies = max(2, -(-768 * 2 ** 20 // (k * n * 2)))
    ws = [torch.randn((k, n) if TRANS else (n, k), device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(copies)]
    x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    row = {"m": m, "k": k, "n": n}
    for d in dists:
        row[f"cold_unsplit_D{d}"], _ = leg(x, ws, MOJO_HIP_GEMM_TILE128="128", MOJO_HIP_GEMM_SPLITK="1", MOJO_HIP_GEMM_RING=str(d))
        row[f"cold_model_D{d}"], f = leg(x, ws, MOJO_HIP_GEMM_TILE128="128", MOJO_HIP_GEMM_RING=str(d))
        row[f"warm_unsplit_D{d}"], _ = leg(x, ws[:1], MOJO_HIP_GEMM_TILE128="128", MOJO_HIP_GEMM_SPLITK="1", MOJO_HIP_GEMM_RING=str(d))
    row["form_model"] = f
    row["t256_cold"], row["f256"] = leg(x, ws, MOJO_HIP_GEMM_TILE128="0")
    j = [0]

    def lib():
        j[0] += 1
        w = ws[j[0] % len(ws)]
        return x @ w if TRANS else torch.nn.functional.linear(x, w)
    row["lib_cold"] = round(_time_graph(lib, reps=max(10, len(ws))) * 1e6, 1)
    print(json.dumps(row), flush=True)
    del ws
