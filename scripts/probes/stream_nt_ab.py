"""A/B in one process: the streaming operators with non-temporal loads / stores forced off and on (MOJO_HIP_STREAM_NT=0 / 1, read
per call) over the bench's streaming cases — which sizes want the by-pass?"""
import os, sys, json, torch
sys.path.insert(0, ".")
from benchmarks import extras as X
dev = torch.device("cuda:0")
res = {}
for rnd in range(2):
    for m in ("0", "1"):
        os.environ["MOJO_HIP_STREAM_NT"] = m
        out = X.bench_streaming(dev)
        for k, v in out.items():
            res.setdefault(k, {}).setdefault(m, []).append(v["us"])
for k, v in res.items():
    a, b = min(v["0"]), min(v["1"])
    print(f"{k:48s} cached {a:8.1f} us   non-temporal {b:8.1f} us   {a / b:5.3f}x", flush=True)
