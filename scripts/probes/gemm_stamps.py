"""Cycle anatomy of the 256x256 GEMM's K loop from in-kernel stamps.  Diagnostic build only:
    MOJO_HIP_EXTRA_CXXFLAGS=-DGEMM_STAMPS python -m mojo_opset_amd.csrc.build --force && python scripts/probes/gemm_stamps.py
Per phase p of a K-tile (4 phases): R = stage 2 LDS-DMA + this phase's fragment reads (issue only), B1 = first barrier,
M = 16 MFMAs, W = counted vmcnt wait, B2 = second barrier.  Waves 4-7 run one barrier behind waves 0-3."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mojo_opset_amd as mo  # noqa: E402
from benchmarks.extras import hip  # noqa: E402

dev = torch.device("cuda", 0)
m, k, n, g = 16384, 4096, 28672, 8
trans = len(sys.argv) > 1 and sys.argv[1] == "nk"
x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
w = torch.randn(g, n, k, device=dev, dtype=torch.bfloat16) if trans else torch.randn(g, k, n, device=dev, dtype=torch.bfloat16)
counts = torch.full((g,), m // g, dtype=torch.int32, device=dev)
op = hip("MojoGroupGemm")(w, trans)
for _ in range(40):                      # settle the clock
    op(x, counts)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.path.join(os.path.dirname(mo.__file__), "lib", "libmojo_hip.so"))
cnt = 8192 * 8 * 32
buf = np.zeros(cnt, dtype=np.uint32)
assert lib.mojo_hip_debug_gemm_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(cnt)) == 0
a = buf.reshape(8192, 8, 32).astype(np.float64)
nkt = a[:, :, 31]
sel = nkt[:, 0] > 0
a = a[sel]
print("workgroups", int(sel.sum()), "K-tiles per tile", a[0, 0, 31])
per = a[:, :, :20] / a[:, :, 31:32]
names = ["R issue", "barrier 1", "MFMA x16", "vmcnt(6)", "barrier 2"]
for grp, label in ((slice(0, 4), "waves 0-3"), (slice(4, 8), "waves 4-7 (one barrier behind)")):
    print(label)
    tot = 0.0
    for p in range(4):
        row = [per[:, grp, 5 * p + i].mean() for i in range(5)]
        tot += sum(row)
        print(f"  P{p + 1}: " + "  ".join(f"{nm} {v:6.0f}" for nm, v in zip(names, row)) + f"   | phase {sum(row):6.0f}")
    print(f"  cycles per K-tile {tot:7.0f}   (1024 of them MFMA issue: 64 x 16)")
