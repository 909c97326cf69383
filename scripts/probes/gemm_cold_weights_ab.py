"""Round 5: the same A/B as gemm_tile128_splitk_ab.py with COLD weights: every call of the captured graph multiplies by another
copy of the weight (copies x bytes >= 768 MB, three times the 256 MB last-level cache), as the layers of a model do — the
one-weight graphs of the other probes re-read a weight of up to 256 MB from the last-level cache.  bf16, [N,K] (or KN)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.extras import _time_graph
from mojo_opset_amd import switches
from mojo_opset_amd.backends.hip import lib as L
from mojo_opset_amd.backends.hip.operators.gemm import dense_gemm
dev = torch.device("cuda", 0)
TRANS = len(sys.argv) > 1 and sys.argv[1] == "KN"
shapes = ((4096, 4096), (8192, 1024), (14336, 4096), (4096, 6144), (4096, 14336), (8192, 8192), (4096, 28672), (2048, 7168),
          (7168, 2048), (5120, 5120), (4096, 1024), (3584, 8192), (1024, 8192))
ms = (8, 32, 64, 100, 128, 160, 256, 384, 512, 768, 1024, 1536, 2048, 4096)
splits = (2, 3, 4, 6, 8, 12, 16)


def leg(x, ws, **env):
    for key in ("MOJO_HIP_GEMM_TILE128", "MOJO_HIP_GEMM_SPLITK"):
        os.environ.pop(key, None)
    os.environ.update(env)
    switches.reload()
    i = [0]

    def fn():
        i[0] += 1
        return dense_gemm(x, ws[i[0] % len(ws)], None, TRANS)
    t = _time_graph(fn, reps=max(10, len(ws)))
    return round(t * 1e6, 1), L.last_launch()


for k, n in shapes:
    copies = max(2, -(-768 * 2 ** 20 // (k * n * 2)))
    ws = [torch.randn((k, n) if TRANS else (n, k), device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(copies)]
    for m in ms:
        tiles = -(-m // 128) * -(-n // 128)
        if -(-m // 128) * -(-n // 256) > 512:
            continue
        x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
        row = {"m": m, "k": k, "n": n, "tiles": tiles, "copies": copies}
        row["t256"], row["f256"] = leg(x, ws, MOJO_HIP_GEMM_TILE128="0")
        if tiles <= 512:
            row["t128_1"], _ = leg(x, ws, MOJO_HIP_GEMM_TILE128="128", MOJO_HIP_GEMM_SPLITK="1")
        if tiles > 128:
            row["t128w"], _ = leg(x, ws, MOJO_HIP_GEMM_TILE128="256", MOJO_HIP_GEMM_SPLITK="1")
        for sk in splits:
            if tiles * sk <= 256 and k // 64 >= 4 * sk:
                row[f"t128_{sk}"], f = leg(x, ws, MOJO_HIP_GEMM_TILE128="1", MOJO_HIP_GEMM_SPLITK=str(sk))
        row["default"], row["form"] = leg(x, ws)
        w0 = ws[:]
        j = [0]

        def lib():
            j[0] += 1
            w = w0[j[0] % len(w0)]
            return x @ w if TRANS else torch.nn.functional.linear(x, w)
        row["lib"] = round(_time_graph(lib, reps=max(10, len(ws))) * 1e6, 1)
        print(json.dumps(row), flush=True)
    del ws
