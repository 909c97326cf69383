"""MLA r=512 latent-kernel tile-loop cycle anatomy from in-kernel stamps.
    MOJO_HIP_EXTRA_CXXFLAGS=-DMLA_STAMPS python -m mojo_opset_amd.csrc.build && python scripts/probes/mla_stamps.py
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mojo_opset_amd as mo  # noqa: E402
from benchmarks.extras import bench_mla_decode  # noqa: E402

print(bench_mla_decode(torch.device("cuda", 0)))
torch.cuda.synchronize()
lib = ctypes.CDLL(os.path.join(os.path.dirname(mo.__file__), "lib", "libmojo_hip.so"))
n = 1024 * 8 * 16
buf = np.zeros(n, dtype=np.uint32)
assert lib.mojo_hip_debug_mla_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(n)) == 0
a = buf.reshape(1024 * 8, 16).astype(np.float64)
tiles = a[:, 15]
sel = tiles > 4
names = ["barrier(3) -> top", "stage_prep (table read, addresses)", "QK^T: 36 reads + 36 mfma + 9 DMA pieces", "(unused)",
         "(unused)", "own max (VALU), softmax, write P + ref", "barrier (2)", "read partner P + ref, joint ref, rescale", "PV: 64 tr reads + 64 mfma",
         "vmcnt(0)", "barrier (3)"]
per = a[sel, :11] / tiles[sel, None]
print("waves", int(sel.sum()), "mean tiles", tiles[sel].mean())
for i, nm in enumerate(names):
    print(f"  {i:2d} {nm:42s} {per[:, i].mean():8.1f} cycles/tile   (p10 {np.percentile(per[:, i], 10):7.1f}  p90 {np.percentile(per[:, i], 90):7.1f})")
print("  total", per.sum(1).mean())
