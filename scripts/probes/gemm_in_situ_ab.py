"""Round 5: dense GEMM forms measured IN SITU: every call sits between other work, as in a model — a graph of reps x (flush, GEMM),
where the flush streams 1 GiB (a copy of 512 MiB: what the attention's K/V does to the caches between two projections) and
each GEMM uses another copy of the weight; the time of a graph of reps x (flush) is subtracted.  Neither the one-weight graphs
(weights from the last-level cache) nor the back-to-back cold graphs (ten launches of one kernel overlapping tail and head, HBM in
steady state) of the other probes are what a layer sees.  bf16, [N,K] weights (argument KN: [K,N])."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.extras import _time_graph
from mojo_opset_amd import switches
from mojo_opset_amd.backends.hip import lib as L
from mojo_opset_amd.backends.hip.operators.gemm import dense_gemm
dev = torch.device("cuda", 0)
TRANS = len(sys.argv) > 1 and sys.argv[1] == "KN"
shapes = ((4096, 4096), (4096, 6144), (14336, 4096), (4096, 28672), (8192, 8192), (8192, 1024), (7168, 2048), (2048, 7168))
ms = (16, 64, 128, 256, 512, 1024, 2048)
splits = (2, 3, 4, 6, 8, 12)
REPS = 8
src = torch.empty(512 * 2 ** 20, dtype=torch.uint8, device=dev)
dst = torch.empty_like(src)


def flush():
    dst.copy_(src)


t_flush = _time_graph(flush, reps=REPS)


def leg(x, ws, **env):
    for key in ("MOJO_HIP_GEMM_TILE128", "MOJO_HIP_GEMM_SPLITK"):
        os.environ.pop(key, None)
    os.environ.update(env)
    switches.reload()
    i = [0]

    def fn():
        flush()
        i[0] += 1
        return dense_gemm(x, ws[i[0] % len(ws)], None, TRANS)
    t = _time_graph(fn, reps=REPS) - t_flush
    return round(t * 1e6, 1), L.last_launch()


print(json.dumps({"flush_us": round(t_flush * 1e6, 1)}), flush=True)
for k, n in shapes:
    ws = [torch.randn((k, n) if TRANS else (n, k), device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(REPS)]
    for m in ms:
        tiles = -(-m // 128) * -(-n // 128)
        if -(-m // 128) * -(-n // 256) > 512:
            continue
        x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
        row = {"m": m, "k": k, "n": n, "tiles": tiles}
        row["t256"], row["f256"] = leg(x, ws, MOJO_HIP_GEMM_TILE128="0")
        if tiles <= 512:
            row["t128_1"], _ = leg(x, ws, MOJO_HIP_GEMM_TILE128="128", MOJO_HIP_GEMM_SPLITK="1")
        if tiles > 128:
            row["t128w"], _ = leg(x, ws, MOJO_HIP_GEMM_TILE128="256", MOJO_HIP_GEMM_SPLITK="1")
        for sk in splits:
            if tiles * sk <= 256 and k // 64 >= 4 * sk:
                row[f"t128_{sk}"], f = leg(x, ws, MOJO_HIP_GEMM_TILE128="1", MOJO_HIP_GEMM_SPLITK=str(sk))
        row["default"], row["form"] = leg(x, ws)
        j = [0]

        def lib():
            flush()
            j[0] += 1
            w = ws[j[0] % len(ws)]
            return x @ w if TRANS else torch.nn.functional.linear(x, w)
        row["lib"] = round((_time_graph(lib, reps=REPS) - t_flush) * 1e6, 1)
        print(json.dumps(row), flush=True)
    del ws
