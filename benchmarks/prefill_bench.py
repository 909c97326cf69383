"""Prefill GQA benchmark (profiling target): python benchmarks/prefill_bench.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks.extras import bench_prefill  # noqa: E402

if __name__ == "__main__":
    print(json.dumps(bench_prefill(torch.device("cuda", 0))))
