"""A/B in one process of a per-call prefill switch (SWITCH=MOJO_HIP_PREFILL_M32 by default: prefill_kernel on 16x16x32 MFMAs
against prefill_m32_kernel on 32x32x16) over the bench cases; the arms are timed alternately, three rounds."""
import os, sys, json, torch
sys.path.insert(0, ".")
from benchmarks import extras as X
dev = torch.device("cuda:0")
sw = os.environ.get("SWITCH", "MOJO_HIP_PREFILL_M32")
res = {}
for rnd in range(3):
    for m in ("0", "1"):
        os.environ[sw] = m
        out = X.bench_prefill(dev)
        for k, v in out.items():
            res.setdefault(k, {}).setdefault(m, []).append((v["us"], v["tflops"]))
rec = {"switch": sw}
for k, v in res.items():
    rec[k] = {("on" if m == "1" else "off"): {"us_min": round(min(x[0] for x in xs), 1), "tflops_max": round(max(x[1] for x in xs)),
                                                "us_all": [round(x[0], 1) for x in xs]} for m, xs in v.items()}
    print(k, {a: (b["us_min"], b["tflops_max"]) for a, b in rec[k].items()}, flush=True)
json.dump(rec, open("gpurun_out/prefill_ab.json", "w"), indent=1)
