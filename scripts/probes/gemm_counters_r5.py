"""Round-5 counter targets for the MFMA-bound GEMM rows:
    python3 scripts/probes/gemm_counters_r5.py wall              # wall A/B in one process: shipped / zeros / hipBLASLt per case
    python3 scripts/probes/gemm_counters_r5.py one <case>        # 12 launches of one case (for a rocprofv3 --pmc pass)
cases: gg_kn (MojoGroupGemm bf16 16384 x 4096 x 28672, 8 experts, [G,K,N]), qg_fp8 / qg_i8 (MojoQuantGemm 4096 x 7168 x 36864 [N,K])"""
import json, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__  # noqa
from benchmarks.extras import hip, _time, group_gemm_case
dev = torch.device("cuda:0")


def make(case, zeros=False):
    if case == "gg_kn":
        m, k, n, groups = 16384, 4096, 28672, 8
        mk = torch.zeros if zeros else torch.randn
        x = mk(m, k, device=dev, dtype=torch.bfloat16)
        w = mk(groups, k, n, device=dev, dtype=torch.bfloat16)
        counts = torch.full((groups,), m // groups, dtype=torch.int32, device=dev)
        op = hip("MojoGroupGemm")(w, False)
        return (lambda: op(x, counts)), 2.0 * m * k * n
    m, k, n = 4096, 7168, 36864
    qd = torch.int8 if case == "qg_i8" else torch.float8_e4m3fn
    op = hip("MojoQuantGemm")(k, n, trans_weight=True, quant_dtype=qd, weight_dtype=qd, device=dev)
    if zeros:
        op.weight.zero_()
        x = torch.zeros(m, k, device=dev).to(qd)
    elif qd == torch.int8:
        op.weight.copy_(torch.randint(-127, 128, (n, k), dtype=torch.int8, device=dev))
        x = torch.randint(-127, 128, (m, k), dtype=torch.int8, device=dev)
    else:
        op.weight.copy_(torch.randn(n, k, device=dev).to(qd))
        x = torch.randn(m, k, device=dev).to(qd)
    op.weight_scale.fill_(0.01)
    s = torch.rand(m, device=dev)
    return (lambda: op(x, s)), 2.0 * m * k * n


if len(sys.argv) > 2 and sys.argv[1] == "one":
    fn, _ = make(sys.argv[2])
    for _ in range(12):
        fn()
    torch.cuda.synchronize()
    sys.exit(0)

out = {}
for case in ("gg_kn", "qg_fp8", "qg_i8"):
    rec = {}
    for arm in ("random", "zeros"):
        fn, flops = make(case, zeros=(arm == "zeros"))
        v = [flops / _time(fn, 10, 2, repeats=3) / 1e12 for _ in range(3)]
        rec[arm + "_tflops"] = {"median": round(statistics.median(v), 1), "min": round(min(v), 1), "max": round(max(v), 1)}
        del fn
        torch.cuda.empty_cache()
    out[case] = rec
# hipBLASLt (torch.matmul per group) on the same box for the headline shape, both weight layouts: calibration, not a product path
for trans in (False, True):
    v = [group_gemm_case(dev, 16384, 4096, 28672, 8, trans, data="torch")["tflops"] for _ in range(3)]
    out["gg_" + ("nk" if trans else "kn") + "_hipblaslt_tflops"] = {"median": round(statistics.median(v), 1), "min": round(min(v), 1), "max": round(max(v), 1)}
    torch.cuda.empty_cache()
print(json.dumps(out))
