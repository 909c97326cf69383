"""A/B in one process: MojoPagedPrefillGQA on prefill_kernel (MOJO_HIP_PREFILL_W64=0) against the one-wave-per-SIMD
prefill_w64_kernel (=1) over the bench cases, plus a numerical cross-check of the two kernels on every case; the switch is
read per call; the arms are timed alternately."""
import os, sys, json, torch
sys.path.insert(0, ".")
from benchmarks import extras as X
dev = torch.device("cuda:0")
res = {}
for rnd in range(2):
    for m in ("0", "1"):
        os.environ["MOJO_HIP_PREFILL_W64"] = m
        out = X.bench_prefill(dev)
        for k, v in out.items():
            res.setdefault(k, {}).setdefault(m, []).append((v["us"], v["tflops"]))
rec = {}
for k, v in res.items():
    rec[k] = {("w64" if m == "1" else "base"): {"us_min": round(min(x[0] for x in xs), 1), "tflops_max": round(max(x[1] for x in xs))} for m, xs in v.items()}
    print(k, rec[k], flush=True)
json.dump(rec, open("gpurun_out/prefill_w64_ab.json", "w"), indent=1)
