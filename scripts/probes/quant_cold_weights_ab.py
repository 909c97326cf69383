"""Round 5: QuantGemm A/B with COLD weights (every call of the captured graph on another operator instance: copies x weight bytes
>= 768 MB): the 256 x 256 kernel (+ its split) / the weight-streaming kernel, the 128-row tiles unsplit and in every K split, the
default.  int8, bf16 out; [N,K] weights (trans_weight=True) or, with the argument KN, the operator's default (K, N) layout."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.extras import _time_graph
from mojo_opset_amd import switches
from mojo_opset_amd.backends.hip import lib as L
from mojo_opset_amd.backends.hip.operators.gemm import HIPQuantGemm
dev = torch.device("cuda", 0)
KN = len(sys.argv) > 1 and sys.argv[1] == "KN"
shapes = ((4096, 4096), (8192, 1024), (7168, 2048), (7168, 4096), (18432, 7168), (4096, 1024), (2048, 7168), (4096, 14336), (8192, 8192),
          (4096, 6144), (7168, 1536))
ms = ((8, 32, 64, 100, 128) if KN else ()) + (160, 256, 384, 512, 768, 1024, 1536, 2048, 4096)
splits = (2, 3, 4, 6, 8, 12, 16)


def leg(ops, x, sc, **env):
    for key in ("MOJO_HIP_GEMM_TILE128", "MOJO_HIP_GEMM_SPLITK"):
        os.environ.pop(key, None)
    os.environ.update(env)
    switches.reload()
    i = [0]

    def fn():
        i[0] += 1
        return ops[i[0] % len(ops)](x, sc)
    t = _time_graph(fn, reps=max(10, len(ops)))
    return round(t * 1e6, 1), L.last_launch()


for k, n in shapes:
    copies = max(2, -(-768 * 2 ** 20 // (k * n)))
    ops = []
    for _ in range(copies):
        op = HIPQuantGemm(k, n, output_dtype=torch.bfloat16, trans_weight=not KN, device=dev)
        op.weight.copy_(torch.randint(-127, 128, (k, n) if KN else (n, k), dtype=torch.int8, device=dev))
        op.weight_scale.fill_(0.01)
        ops.append(op)
    for m in ms:
        tiles = -(-m // 128) * -(-n // 128)
        if -(-m // 128) * -(-n // 256) > 512:
            continue
        x = torch.randint(-127, 128, (m, k), dtype=torch.int8, device=dev)
        sc = torch.rand(m, device=dev)
        row = {"m": m, "k": k, "n": n, "tiles": tiles, "copies": copies}
        row["t256"], row["f256"] = leg(ops, x, sc, MOJO_HIP_GEMM_TILE128="0")
        if tiles <= 512:
            row["t128_1"], _ = leg(ops, x, sc, MOJO_HIP_GEMM_TILE128="128", MOJO_HIP_GEMM_SPLITK="1")
        if tiles > 128:
            row["t128w"], _ = leg(ops, x, sc, MOJO_HIP_GEMM_TILE128="256", MOJO_HIP_GEMM_SPLITK="1")
        for sk in splits:
            if tiles * sk <= 256 and k // 128 >= 4 * sk:
                row[f"t128_{sk}"], f = leg(ops, x, sc, MOJO_HIP_GEMM_TILE128="1", MOJO_HIP_GEMM_SPLITK=str(sk))
        row["default"], row["form"] = leg(ops, x, sc)
        print(json.dumps(row), flush=True)
    del ops
