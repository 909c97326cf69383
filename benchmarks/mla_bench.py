"""MLA decode benchmark (profiling target): python benchmarks/mla_bench.py [iters]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks import extras  # noqa: E402

if __name__ == "__main__":
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    _time = extras._time
    extras._time = lambda fn, it=10, warmup=2: _time(fn, max(it, iters), max(warmup, iters // 4))
    print(json.dumps(extras.bench_mla_decode(torch.device("cuda", 0))))
