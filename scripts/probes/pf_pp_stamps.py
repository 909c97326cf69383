"""Phase anatomy of prefill_pp_kernel from in-kernel stamps.  Build with MOJO_HIP_EXTRA_CXXFLAGS=-DPF_STAMPS first:
    MOJO_HIP_EXTRA_CXXFLAGS=-DPF_STAMPS python -m mojo_opset_amd.csrc.build --force && MOJO_HIP_PREFILL_PP=1 python scripts/probes/pf_pp_stamps.py
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mojo_opset_amd as mo  # noqa: E402
from benchmarks.extras import _paged, hip  # noqa: E402

dev = torch.device("cuda", 0)
hq, hkv, d, page = 32, 8, 128, 16
op = hip("MojoPagedPrefillGQA")()
q_lens, kv = [16384], [16384]
k, v, table = _paged(dev, kv, hkv, d, page)
q = torch.randn(sum(q_lens), hq, d, device=dev, dtype=torch.bfloat16)
cu = torch.tensor([0, 16384], dtype=torch.int32, device=dev)
for _ in range(3):
    op(q, k, v, cu, table, cu_total_seq_lens=cu, max_q_len=16384, max_total_seq_len=16384)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.path.join(os.path.dirname(mo.__file__), "lib", "libmojo_hip.so"))
n = 8192 * 4 * 16
buf = np.zeros(n, dtype=np.uint32)
rc = lib.mojo_hip_debug_prefill_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(n))
assert rc == 0
a = buf.reshape(-1, 16)[: 2048 * 8].astype(np.float64)
tiles = a[:, 15]
names = ["loop top", "softmax (+4 DMA pieces)", "V0 issue + vmcnt wait", "barrier 1", "V0 wait, V1 issue, PV0 mfma", "V1 wait, PV1 mfma",
         "K reads issue", "K wait + QK mfma", "barrier 2"]
for g, gname in ((0, "group A (waves 0-3)"), (1, "group B (waves 4-7)")):
    wave = np.arange(a.shape[0]) % 8
    sel = (tiles > 32) & ((wave >> 2) == g)
    per = a[sel, :9] / tiles[sel, None]
    print(gname, "waves", int(sel.sum()), "mean tiles", tiles[sel].mean())
    for i, nm in enumerate(names):
        print(f"  {i} {nm:32s} {per[:, i].mean():8.1f} cycles/tile   (p10 {np.percentile(per[:, i], 10):7.1f}  p90 {np.percentile(per[:, i], 90):7.1f})")
    print("  total", per.sum(1).mean())
