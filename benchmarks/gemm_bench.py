"""Single-configuration grouped-GEMM benchmark (profiling target).

    python benchmarks/gemm_bench.py --m 16384 --k 4096 --n 28672 --groups 8 [--trans] [--iters 20]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks.extras import group_gemm_case  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=16384)
    ap.add_argument("--k", type=int, default=4096)
    ap.add_argument("--n", type=int, default=28672)
    ap.add_argument("--groups", type=int, default=8)
    ap.add_argument("--trans", action="store_true")
    ap.add_argument("--split", default="balanced")
    ap.add_argument("--data", default="randn", choices=["randn", "zeros", "torch"])
    ns = ap.parse_args()
    r = group_gemm_case(torch.device("cuda", 0), ns.m, ns.k, ns.n, ns.groups, ns.trans, ns.split, data=ns.data)
    r["config"] = vars(ns)
    print(json.dumps(r))
