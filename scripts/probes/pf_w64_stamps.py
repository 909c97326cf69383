"""Cycle anatomy of prefill_w64_kernel's interleaved iteration from in-kernel stamps.  Build with -DPF_STAMPS first (the
OUTPUT of such a build is not valid: s_memtime shares lgkmcnt with the kernel's counted LDS waits):
    MOJO_HIP_EXTRA_CXXFLAGS=-DPF_STAMPS python -m mojo_opset_amd.csrc.build && MOJO_HIP_PREFILL_W64=1 python scripts/probes/pf_w64_stamps.py
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mojo_opset_amd as mo  # noqa: E402
from benchmarks.extras import _paged, hip  # noqa: E402

os.environ["MOJO_HIP_PREFILL_W64"] = "1"
dev = torch.device("cuda", 0)
hq, hkv, d, page = 32, 8, 128, 16
op = hip("MojoPagedPrefillGQA")()
n_tok = int(os.environ.get("W64_STAMP_TOKENS", "16384"))
k, v, table = _paged(dev, [n_tok], hkv, d, page)
q = torch.randn(n_tok, hq, d, device=dev, dtype=torch.bfloat16)
cu = torch.tensor([0, n_tok], dtype=torch.int32, device=dev)
for _ in range(3):
    op(q, k, v, cu, table, cu_total_seq_lens=cu, max_q_len=n_tok, max_total_seq_len=n_tok)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.path.join(os.path.dirname(mo.__file__), "lib", "libmojo_hip.so"))
n = 8192 * 4 * 16
buf = np.zeros(n, dtype=np.uint32)
assert lib.mojo_hip_debug_prefill_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(n)) == 0
a = buf.reshape(8192 * 4, 16).astype(np.float64)
iters = a[:, 15]
sel = iters > 32
names = ["barrier -> top of iteration (loop control)", "page ids, first V group, reference check, scalar address work",
         "PV half: 32 MFMAs + softmax rows 0-31 + K pieces", "QK^T half: 32 MFMAs + softmax rows 32-63 + V pieces + maxima",
         "settle + last maxima", "vmcnt(4)", "barrier"]
per = a[sel, :7] / iters[sel, None]
print("waves", int(sel.sum()), "mean interleaved iterations", iters[sel].mean())
for i, nm in enumerate(names):
    print(f"  {i} {nm:70s} {per[:, i].mean():8.1f} cycles/iteration   (p10 {np.percentile(per[:, i], 10):7.1f}  p90 {np.percentile(per[:, i], 90):7.1f})")
print("  total", per.sum(1).mean())
