"""A/B in one process: MojoPagedPrefillGQA on prefill_kernel (MOJO_HIP_PREFILL_PP=0) against the phase-alternating
prefill_pp_kernel (=1) over the bench cases; the switch is read per call; the arms are timed alternately."""
import os, sys, json, torch
sys.path.insert(0, ".")
from benchmarks import extras as X
dev = torch.device("cuda:0")
res = {}
for rnd in range(2):
    for m in ("0", "1"):
        os.environ["MOJO_HIP_PREFILL_PP"] = m
        out = X.bench_prefill(dev)
        for k, v in out.items():
            res.setdefault(k, {}).setdefault(m, []).append((v["us"], v["tflops"]))
for k, v in res.items():
    print(k, {m: (round(min(x[0] for x in xs), 1), round(max(x[1] for x in xs))) for m, xs in v.items()}, flush=True)
