"""Round 5: K split of the 128 x 128-tile QuantGemm (gemm_tile128_core.h, int32 / fp32 slabs + quant_finalize_kernel) on few-tile /
long-K products, against the same kernel unsplit, the 256 x 256 kernel with its own split and the library's default; int8 (fp8 with
the third argument "fp8", [N,K] only), bf16 output, random data; device time (ten calls per HIP graph, sustained medians)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.extras import _time_graph
from mojo_opset_amd import switches
from mojo_opset_amd.backends.hip import lib as L
from mojo_opset_amd.backends.hip.operators.gemm import HIPQuantGemm
dev = torch.device("cuda", 0)
KN = len(sys.argv) > 1 and sys.argv[1] == "KN"
qd = torch.float8_e4m3fn if len(sys.argv) > 2 and sys.argv[2] == "fp8" else torch.int8
shapes = ((4096, 4096), (8192, 1024), (7168, 2048), (7168, 4096), (18432, 7168), (4096, 1024), (2048, 7168), (4096, 14336), (8192, 8192))
ms = ((8, 32, 64) if KN else ()) + (100, 160, 256, 384, 512, 768, 1024)
splits = (2, 3, 4, 6, 8, 12, 16)


def leg(op, x, sc, **env):
    for key in ("MOJO_HIP_GEMM_TILE128", "MOJO_HIP_GEMM_SPLITK"):
        os.environ.pop(key, None)
    os.environ.update(env)
    switches.reload()
    t = _time_graph(lambda: op(x, sc), reps=10)
    return round(t * 1e6, 1), L.last_launch()


for k, n in shapes:
    op = HIPQuantGemm(k, n, output_dtype=torch.bfloat16, trans_weight=not KN, quant_dtype=qd, weight_dtype=qd, device=dev)
    w_nk = torch.randint(-127, 128, (n, k), dtype=torch.int8, device=dev) if qd == torch.int8 else torch.randn(n, k, device=dev).to(qd)
    op.weight.copy_(w_nk.t() if KN else w_nk)
    op.weight_scale.fill_(0.01)
    for m in ms:
        tiles = -(-m // 128) * -(-n // 128)
        if tiles > 256:
            continue
        x = torch.randint(-127, 128, (m, k), dtype=torch.int8, device=dev) if qd == torch.int8 else torch.randn(m, k, device=dev).to(qd)
        sc = torch.rand(m, device=dev)
        row = {"m": m, "k": k, "n": n, "tiles": tiles}
        row["t256"], row["f256"] = leg(op, x, sc, MOJO_HIP_GEMM_TILE128="0")
        if m > 64 or KN:
            row["t128_1"], f = leg(op, x, sc, MOJO_HIP_GEMM_TILE128="1", MOJO_HIP_GEMM_SPLITK="1")
            if f.startswith("gemm128"):
                for sk in splits:
                    if tiles * sk <= 512 and k // 128 >= 2 * sk:
                        row[f"t128_{sk}"], f = leg(op, x, sc, MOJO_HIP_GEMM_TILE128="1", MOJO_HIP_GEMM_SPLITK=str(sk))
                        assert f.endswith(":splitk"), f
            else:
                del row["t128_1"]
        row["default"], row["form"] = leg(op, x, sc)
        print(json.dumps(row), flush=True)
    del op
