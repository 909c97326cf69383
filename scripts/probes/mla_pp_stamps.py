"""Ping-pong MLA latent kernel: cycles per 32-key tile between the in-kernel stamps, leading and late waves apart.
    MOJO_HIP_EXTRA_CXXFLAGS=-DMLA_STAMPS python -m mojo_opset_amd.csrc.build && MOJO_HIP_MLA_KERNEL=pp python scripts/probes/mla_pp_stamps.py
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mojo_opset_amd as mo  # noqa: E402
from benchmarks.extras import bench_mla_decode  # noqa: E402

os.environ["MOJO_HIP_MLA_KERNEL"] = "pp"
print(bench_mla_decode(torch.device("cuda", 0)))
torch.cuda.synchronize()
lib = ctypes.CDLL(os.path.join(os.path.dirname(mo.__file__), "lib", "libmojo_hip.so"))
n = 1024 * 8 * 16
buf = np.zeros(n, dtype=np.uint32)
assert lib.mojo_hip_debug_mla_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(n)) == 0
a = buf.reshape(1024, 8, 16).astype(np.float64)
names = ["loop top", "QK^T: 18 reads + 18 mfma (+ nops)", "mask, max, exp, publish P + ref", "late waves: stage tile + wait landed", "barrier 1",
         "read partner, joint ref, rescale", "PV: 32 tr reads + 16 mfma", "leading waves: stage tile + wait landed", "barrier 2"]
for grp_name, waves in (("leading waves 0-3", slice(0, 4)), ("late waves 4-7", slice(4, 8))):
    x = a[:, waves, :].reshape(-1, 16)
    tiles = x[:, 15]
    sel = tiles > 8
    per = x[sel, :9] / tiles[sel, None]
    print(f"{grp_name}: {int(sel.sum())} waves, mean tiles {tiles[sel].mean():.1f}")
    for i, nm in enumerate(names):
        print(f"  {i:2d} {nm:44s} {per[:, i].mean():8.1f} cycles/tile   (p10 {np.percentile(per[:, i], 10):7.1f}  p90 {np.percentile(per[:, i], 90):7.1f})")
    print(f"     total per 32-key tile {per.sum(1).mean():8.1f}   (x2 = {2 * per.sum(1).mean():.0f} per 64 keys; the lock-step kernel: 5 745)")
