"""Round 5: K split of the 128 x 128-tile dense GEMM (csrc/gemm_tile128_core.h) on few-tile / long-K products — a chunk of
129-768 rows against a 1024-5120-wide projection — against the same kernel unsplit, the 256 x 256 kernel with its own best
split, the library's default choice and hipBLASLt (F.linear / x @ w); bf16, random data; device time (ten calls per HIP graph,
sustained medians).  One JSON line per shape: the time of every forced split, the default's time and form."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.extras import _time_graph
from mojo_opset_amd import switches
from mojo_opset_amd.backends.hip import lib as L
from mojo_opset_amd.backends.hip.operators.gemm import dense_gemm
dev = torch.device("cuda", 0)
shapes = ((4096, 4096), (8192, 1024), (7168, 2048), (14336, 4096), (5120, 5120), (4096, 1024), (2048, 7168), (4096, 14336), (8192, 8192))
if len(sys.argv) > 2 and sys.argv[2] == "small":
    shapes += ((8192, 28672), (28672, 8192), (4096, 28672))
TRANS = len(sys.argv) > 1 and sys.argv[1] == "KN"
ms = (8, 32, 64) if len(sys.argv) > 2 and sys.argv[2] == "small" else (100, 160, 256, 384, 512, 768, 1024)
splits = (2, 3, 4, 6, 8, 12, 16)


def leg(x, w, **env):
    for key in ("MOJO_HIP_GEMM_TILE128", "MOJO_HIP_GEMM_SPLITK"):
        os.environ.pop(key, None)
    os.environ.update(env)
    switches.reload()
    t = _time_graph(lambda: dense_gemm(x, w, None, TRANS), reps=10)
    return round(t * 1e6, 1), L.last_launch()


for k, n in shapes:
    w = torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.02
    if TRANS:
        w = w.t().contiguous()
    for m in ms:
        tiles = -(-m // 128) * -(-n // 128)
        if tiles > 256:
            continue
        x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
        row = {"m": m, "k": k, "n": n, "tiles": tiles}
        row["t256"], row["f256"] = leg(x, w, MOJO_HIP_GEMM_TILE128="0")
        row["t128_1"], _ = leg(x, w, MOJO_HIP_GEMM_TILE128="1", MOJO_HIP_GEMM_SPLITK="1")
        for sk in splits:
            if tiles * sk <= 512 and k // 64 >= 2 * sk:
                row[f"t128_{sk}"], f = leg(x, w, MOJO_HIP_GEMM_TILE128="1", MOJO_HIP_GEMM_SPLITK=str(sk))
                assert f.endswith(":splitk"), f
        row["default"], row["form"] = leg(x, w)
        row["lib"] = round(_time_graph((lambda: x @ w) if TRANS else (lambda: torch.nn.functional.linear(x, w)), reps=10) * 1e6, 1)
        print(json.dumps(row), flush=True)
