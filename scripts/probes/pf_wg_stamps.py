"""Prefill workgroup-level phase times (prologue / hot loop / remaining loops / epilogue) for the short-sequence cases.
    MOJO_HIP_EXTRA_CXXFLAGS=-DPF_WG_STAMPS python -m mojo_opset_amd.csrc.build && python scripts/probes/pf_wg_stamps.py
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
lib = ctypes.CDLL(os.path.join(os.path.dirname(mo.__file__), "lib", "libmojo_hip.so"))
g = torch.Generator().manual_seed(20260716)
ragged = torch.randint(512, 1025, (16,), generator=g).tolist()
for name, q_lens in {"16 ragged 512..1024": ragged, "4 x 2048": [2048] * 4}.items():
    k, v, table = _paged(dev, q_lens, hkv, d, page)
    q = torch.randn(sum(q_lens), hq, d, device=dev, dtype=torch.bfloat16)
    cu = torch.tensor([0] + torch.tensor(q_lens).cumsum(0).tolist(), dtype=torch.int32, device=dev)
    n = 8192 * 4 * 16
    zero = np.zeros(n, dtype=np.uint32)
    for _ in range(3):
        op(q, k, v, cu, table, cu_total_seq_lens=cu, max_q_len=max(q_lens), max_total_seq_len=max(q_lens))
    torch.cuda.synchronize()
    buf = np.zeros(n, dtype=np.uint32)
    assert lib.mojo_hip_debug_prefill_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(n)) == 0
    raw = buf.reshape(-1, 16)
    n_units = ((max(q_lens) + 127) // 32) * hkv * len(q_lens)          # an upper bound on this case's block ids (G = 4)
    raw = raw[:n_units]
    a = raw.astype(np.float64)
    live = a[:, 5] > 0
    live &= a[:, 7] > a[live, 7].max() - 2e5           # rows this launch did not write keep stamps of an earlier case: drop them
    hw, xcc = raw[live, 8], raw[live, 9] & 0xF
    cu_key = (xcc.astype(np.int64) << 16) | (hw & 0xFF00)               # xcc | se, sh, cu bits of HW_ID
    a = a[live]
    pro, hot, rest, end = a[:, 0], a[:, 1] - a[:, 0], a[:, 2] - a[:, 1], a[:, 3] - a[:, 2]
    print(f"{name}: {int(live.sum())} live workgroups, tiles/WG {a[:, 5].mean():.1f} (hot-loop tiles {a[:, 4].mean():.1f})")
    t0, t1 = a[:, 6] / 100.0, a[:, 7] / 100.0                      # us (100 MHz counter, low 32 bits)
    base = t0.min()
    t0, t1 = t0 - base, t1 - base
    span = t1.max()
    grid = np.linspace(0, span, 21)
    occ = [int(((t0 <= g) & (t1 > g)).sum()) for g in grid]
    print(f"   kernel span {span:6.1f} us; live workgroups resident at 5 % steps: {occ}")
    print(f"   distinct CUs seen: {len(np.unique(cu_key))}")
    for g in grid[1:7]:
        res = (t0 <= g) & (t1 > g)
        _, cnt = np.unique(cu_key[res], return_counts=True)
        print(f"   t = {g:6.1f} us: CUs holding 1 / 2 / 3+ workgroups: {(cnt == 1).sum()} / {(cnt == 2).sum()} / {(cnt >= 3).sum()};"
              f" started so far {(t0 <= g).sum()}, finished {(t1 <= g).sum()}, tiles of the resident ones: mean {a[res, 5].mean():.1f}")
    order = np.argsort(t0)
    print("   start times of the first 600 workgroups by start order (us), every 40th:", np.round(t0[order][:600:40], 2).tolist())
    print(f"   start of the last workgroup {t0.max():6.1f} us; sum of workgroup times / 512 slots = {(t1 - t0).sum() / 512:6.1f} us")
    print(f"   prologue {pro.mean():8.0f} cycles   hot loop {hot.mean():8.0f} ({(hot / np.maximum(a[:, 4], 1)).mean():6.0f}/tile)"
          f"   other loops {rest.mean():8.0f} ({(rest / np.maximum(a[:, 5] - a[:, 4], 1)).mean():6.0f}/tile)   epilogue {end.mean():6.0f}   total {a[:, 3].mean():8.0f}")
