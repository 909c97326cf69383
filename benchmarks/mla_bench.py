"""MLA decode benchmark (profiling target): python benchmarks/mla_bench.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks.extras import bench_mla_decode  # noqa: E402

if __name__ == "__main__":
    print(json.dumps(bench_mla_decode(torch.device("cuda", 0))))
