"""Run selected extras (profiling / iteration target):  python benchmarks/one.py bench_mla_prefill bench_decode_variants ..."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks import extras  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    for name in sys.argv[1:]:
        fn = getattr(extras, name)
        res = fn(dev, 1, 0) if name == "bench_compute_comm" else fn(dev)
        print(json.dumps({name: res}), flush=True)
