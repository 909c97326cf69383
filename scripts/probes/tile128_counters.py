"""Round 5: counter targets for the 128-row-tile GEMM (csrc/gemm_tile128_core.h), the one kernel of the round without counters yet.
    python3 scripts/probes/tile128_counters.py one <case>        # 14 launches of one case (for a rocprofv3 --pmc pass)
    python3 scripts/probes/tile128_counters.py wall              # wall time per case under graph replay + the form each call took
cases: d1024 (dense bf16 1024 x 4096 x 4096, [N,K], ONE weight: last-level-cache resident), d1024c (the same with every launch on
another copy of the weight: cold, as a model's layers are), d2048w (2048 x 4096 x 4096: the 128 x 256 eight-wave shape), d256sk
(256 x 8192 x 1024: the tiles' own K split + finalize), q1024 (MojoQuantGemm int8 1024 x 4096 x 4096 on the same core)"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__  # noqa
from benchmarks.extras import hip, _time
from mojo_opset_amd.backends.hip import lib as L
from mojo_opset_amd.backends.hip.operators.gemm import dense_gemm
dev = torch.device("cuda:0")
SHAPES = {"d1024": (1024, 4096, 4096), "d1024c": (1024, 4096, 4096), "d2048w": (2048, 4096, 4096), "d256sk": (256, 8192, 1024),
          "q1024": (1024, 4096, 4096)}


def make(case, copies=14):
    m, k, n = SHAPES[case]
    if case == "q1024":
        op = hip("MojoQuantGemm")(k, n, trans_weight=True, quant_dtype=torch.int8, weight_dtype=torch.int8, device=dev)
        op.weight.copy_(torch.randint(-127, 128, (n, k), dtype=torch.int8, device=dev))
        op.weight_scale.fill_(0.01)
        x = torch.randint(-127, 128, (m, k), dtype=torch.int8, device=dev)
        s = torch.rand(m, device=dev)
        return (lambda i: op(x, s)), 2.0 * m * k * n
    x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    ws = [torch.randn(n, k, device=dev, dtype=torch.bfloat16) for _ in range(copies if case.endswith("c") else 1)]
    return (lambda i: dense_gemm(x, ws[i % len(ws)], None, False)), 2.0 * m * k * n


if len(sys.argv) > 2 and sys.argv[1] == "one":
    fn, _ = make(sys.argv[2])
    for i in range(14):
        fn(i)
    torch.cuda.synchronize()
    sys.exit(0)

out = {}
for case in SHAPES:
    fn, flops = make(case)
    fn(0)
    form = L.last_launch()
    box = {"i": 0}

    def call():
        box["i"] += 1
        return fn(box["i"])
    t = _time(call, 14, 2, repeats=3)
    out[case] = {"shape": SHAPES[case], "form": form, "us": round(t * 1e6, 2), "tflops": round(flops / t / 1e12, 1)}
    del fn
    torch.cuda.empty_cache()
print(json.dumps(out))
