"""Sweep the K split of the fused decode forms (mojo_hip_gemm_residual_rmsnorm, mojo_hip_qkv_rope_store) and the waves per
workgroup of mojo_hip_gemm_swiglu on the Llama-3-8B decode shapes, under graph replay with weight copies in rotation.

    python3 scripts/probes/fused_split_sweep.py > profiles/r4_fused_split_sweep.txt          (GPU box)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.extras import _time_graph  # noqa: E402
from mojo_opset_amd.backends.hip.operators.gemm import dense_gemm_residual_rmsnorm, dense_gemm_swiglu, qkv_rope_store  # noqa: E402

dev = torch.device("cuda:0")
dt = torch.bfloat16


def copies(n, k):
    return [torch.randn(n, k, device=dev, dtype=dt) * 0.02 for _ in range(max(2, min(8, int(600e6 // (n * k * 2)))))]


def sweep(name, make_fn, ws, values, env):
    row = []
    for v in values:
        if v:
            os.environ[env] = str(v)
        else:
            os.environ.pop(env, None)
        it = [0]

        def fn():
            it[0] += 1
            return make_fn(ws[it[0] % len(ws)])

        fn()
        t = _time_graph(fn, reps=len(ws) * 4, replays=5)
        row.append(f"{env.split('_')[-1].lower()}={v or 'auto'}: {t * 1e6:6.1f} us")
    os.environ.pop(env, None)
    print(name + "\n   " + "\n   ".join(row), flush=True)


m = 64
for k, n in ((4096, 4096), (14336, 4096), (8192, 8192)):
    x = torch.randn(m, k, device=dev, dtype=dt)
    r = torch.randn(m, n, device=dev, dtype=dt)
    nw = torch.ones(n, device=dev, dtype=dt)
    ws = copies(n, k)
    sweep(f"gemm_residual_rmsnorm M={m} K={k} N={n}", lambda w: dense_gemm_residual_rmsnorm(x, w, None, r, nw, 1e-5), ws,
          (0, 2, 3, 4, 5, 6, 8, 10, 12, 16), "MOJO_HIP_GEMM_SKINNY_SPLITK")
    del ws
hq, hkv, d, page, ctx = 32, 8, 128, 16, 4096
x = torch.randn(m, 4096, device=dev, dtype=dt)
cos, sin = torch.randn(m, d, device=dev), torch.randn(m, d, device=dev)
nb = m * (ctx // page + 1)
kc = torch.zeros(nb, hkv, page, d, device=dev, dtype=dt)
vc = torch.zeros_like(kc)
table = torch.randperm(nb, device=dev).view(m, -1).to(torch.int32)
ctx_t = torch.full((m,), ctx, dtype=torch.int32, device=dev)
ws = copies((hq + 2 * hkv) * d, 4096)
sweep("qkv_rope_store M=64 K=4096 N=6144", lambda w: qkv_rope_store(x, w, None, cos, sin, kc, vc, table, ctx_t, hq, hkv), ws,
      (0, 2, 3, 4, 5, 6, 8, 10, 12, 16), "MOJO_HIP_GEMM_SKINNY_SPLITK")
del ws
for inter in (14336, 28672, 18432, 8192):
    ws = copies(2 * inter, 4096)
    sweep(f"gemm_swiglu M=64 K=4096 inter={inter}", lambda w: dense_gemm_swiglu(x, w), ws, (0, 4, 5, 6, 7, 8), "MOJO_HIP_GEMM_GLU_WAVES")
    del ws
    torch.cuda.empty_cache()
