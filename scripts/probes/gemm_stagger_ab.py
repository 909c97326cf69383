"""Start stagger of the 256x256 GEMM (MOJO_HIP_GEMM_STAGGER=<10-ns ticks per phase>, read per call) on the MLA decompression
product [T_kv, 512] x [H * 256, 512]^T and on two long-K controls.  python scripts/probes/gemm_stagger_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.extras import _time, hip  # noqa: E402

dev = torch.device("cuda", 0)
for m, k, n in ((2048, 512, 32768), (10240, 512, 32768), (4096, 1024, 16384), (16384, 4096, 4096)):
    x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    w = torch.randn(1, n, k, device=dev, dtype=torch.bfloat16) * 0.05
    op = hip("MojoGroupGemm")(w, True)
    counts = torch.tensor([m], dtype=torch.int32, device=dev)
    ref = None
    row = []
    for rep in range(2):
        for st in (0, 60, 120, 180, 240, 320, 480):
            os.environ["MOJO_HIP_GEMM_STAGGER"] = str(st)
            out = op(x, counts)
            if ref is None:
                ref = out.clone()
            assert torch.equal(out, ref)
            t = _time(lambda: op(x, counts), 20, 5)
            row.append(f"stagger {st:4d}: {t * 1e6:7.1f} us = {2.0 * m * k * n / t / 1e12:6.0f} TF")
    os.environ.pop("MOJO_HIP_GEMM_STAGGER")
    print(f"M {m} K {k} N {n}\n   " + "\n   ".join(row), flush=True)
