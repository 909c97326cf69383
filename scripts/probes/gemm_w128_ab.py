"""A/B (experiments build): MojoGroupGemm bf16 on the shipped 8-wave 256x256 kernel against experiments/gemm_w128.h (four waves of
128x128, MOJO_HIP_GEMM_W128=1) on the Mixtral up case with [N,K] weights (the experiment has that layout only), random data, one
process, arms alternated.  `one <0|1>`: a few launches of one arm (for rocprofv3 --pmc passes)."""
import json, os, sys, torch
sys.path.insert(0, ".")
from benchmarks import extras as X
from benchmarks.extras import hip
dev = torch.device("cuda:0")
SW = "MOJO_HIP_GEMM_W128"
m, k, n, g = 16384, 4096, 28672, 8
if len(sys.argv) > 2 and sys.argv[1] == "one":
    os.environ[SW] = sys.argv[2]
    x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    w = torch.randn(g, n, k, device=dev, dtype=torch.bfloat16)
    counts = torch.full((g,), m // g, dtype=torch.int32, device=dev)
    op = hip("MojoGroupGemm")(w, True)
    for _ in range(8):
        op(x, counts)
    torch.cuda.synchronize()
    sys.exit(0)
# correctness: same accumulation order per output element -> the two kernels must agree bit for bit
torch.manual_seed(4)
for (mm, kk, nn, gg) in ((1024, 512, 768, 2), (2048 + 48, 4096, 1024 + 16, 3), (512, 128, 256, 1)):
    x = torch.randn(mm, kk, device=dev, dtype=torch.bfloat16)
    w = torch.randn(gg, nn, kk, device=dev, dtype=torch.bfloat16)
    counts = torch.full((gg,), mm // gg, dtype=torch.int32)
    counts[-1] += mm - int(counts.sum())
    counts = counts.to(dev)
    op = hip("MojoGroupGemm")(w, True)
    os.environ[SW] = "0"; a = op(x, counts).clone()
    os.environ[SW] = "1"; b = op(x, counts).clone()
    print("bit-identical", (mm, kk, nn, gg), bool(torch.equal(a, b)), float((a.float() - b.float()).abs().max()), flush=True)
    assert torch.equal(a, b)
res = {}
for rnd in range(3):
    for arm in ("0", "1"):
        os.environ[SW] = arm
        for name, trans in (("mixtral_up_16384x4096x28672_G8_NK", True),):
            r = X.group_gemm_case(dev, m, k, n, g, trans)
            res.setdefault(name, {}).setdefault(arm, []).append((r["us"], r["tflops"]))
        r = X.group_gemm_case(dev, 16384, 14336, 4096, 8, True)
        res.setdefault("mixtral_down_16384x14336x4096_G8_NK", {}).setdefault(arm, []).append((r["us"], r["tflops"]))
rec = {"switch": SW, "note": "arm 0 = gemm256_kernel (8 waves, 128x64 per wave), arm 1 = gemm_w128_kernel (4 waves, 128x128 per wave)"}
for k_, v in res.items():
    rec[k_] = {("w128" if a == "1" else "shipped"): {"us_min": round(min(t[0] for t in xs), 1), "tflops_max": round(max(t[1] for t in xs)),
                                                      "us_all": [round(t[0], 1) for t in xs]} for a, xs in v.items()}
print(json.dumps(rec))
json.dump(rec, open("gpurun_out/gemm_w128_ab.json", "w"), indent=1)
