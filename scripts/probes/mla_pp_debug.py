"""Ping-pong latent kernel called through the C ABI with two key splits: which of the partial results (reference maximum,
row sum, un-normalised O) vary from run to run?  (debug aid)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mojo_opset_amd.backends.hip import lib as L  # noqa: E402

dev = torch.device("cuda", 0)
os.environ["MOJO_HIP_MLA_SPLITS"] = "2"
lib = L.load()
r, rope, page = 512, 64, 16
for kernel in ("oct", "pp"):
    os.environ["MOJO_HIP_MLA_KERNEL"] = kernel
    for (t, h, n) in ((1, 32, 256), (1, 64, 256), (2, 128, 700), (8, 128, 2000)):
        g = torch.Generator().manual_seed(5)
        pages = (n + page - 1) // page
        q = (torch.randn(t, h, r + rope, generator=g) * 0.3).to(torch.bfloat16).to(dev)
        ckv = torch.randn(pages * t + 2, 1, page, r, generator=g).to(torch.bfloat16).to(dev)
        kpe = torch.randn(pages * t + 2, 1, page, rope, generator=g).to(torch.bfloat16).to(dev)
        table = torch.arange(pages * t, dtype=torch.int32).view(t, pages).to(dev)
        lens = torch.full((t,), n, dtype=torch.int32, device=dev)
        o = torch.empty(t, h, r, dtype=torch.bfloat16, device=dev)
        nbytes = lib.mojo_hip_mla_latent_attn_workspace_bytes(t, h, r, n)
        runs = []
        for _ in range(30):
            ws = torch.zeros(max(nbytes, 64), dtype=torch.uint8, device=dev)
            L.check(lib.mojo_hip_mla_latent_attn(
                L.ptr(q), r + rope, L.ptr(None), 0, L.ptr(ckv), L.ptr(kpe), L.ptr(lens), L.ptr(None), L.ptr(None), L.ptr(table), L.ptr(None),
                L.ptr(o), L.ptr(ws), ws.numel(), t, t, h, r, rope, page, pages, table.stride(0), ckv.stride(0), ckv.stride(2),
                kpe.stride(0), kpe.stride(2), n, 0.07, L.dtype_code(torch.bfloat16), L.stream_of(q)), "latent")
            torch.cuda.synchronize()
            f = ws.view(torch.float32)
            slots = t * 2 * h
            po = f[: slots * r].view(t, 2, h, r).clone().cpu()
            ml = f[slots * r: slots * r + slots * 2].view(t, 2, h, 2).clone().cpu()
            runs.append((po, ml, o.float().cpu().clone()))
        po_var = torch.stack([x[0] for x in runs]).std(0)
        ml_var = torch.stack([x[1] for x in runs]).std(0)
        o_var = torch.stack([x[2] for x in runs]).std(0)
        print(f"{kernel} T={t} H={h} keys={n}: partial O varies in {(po_var.amax(-1) > 0).sum().item()} (split, head) rows, "
              f"max m varies {(ml_var[..., 0] > 0).sum().item()}, row sum varies {(ml_var[..., 1] > 0).sum().item()}, out rows {(o_var.amax(-1) > 0).sum().item()}; "
              f"rel size of O variation {(po_var.amax() / runs[0][0].abs().amax()).item():.2e}", flush=True)
        if kernel == "pp":
            mls = torch.stack([x[1] for x in runs])          # [runs, t, 2, h, 2]
            rows = (ml_var[..., 0] > 0).nonzero().tolist()[:8]
            for (tt, sp, hh) in rows:
                vals = sorted(set(round(v, 4) for v in mls[:, tt, sp, hh, 0].tolist()))
                ls = sorted(set(round(v, 4) for v in mls[:, tt, sp, hh, 1].tolist()))
                print(f"    split {sp} head {hh}: m in {vals}   l in {ls}")

            rows = (po_var.amax(-1) > 0).nonzero().tolist()[:6]
            for (tt, sp, hh) in rows:
                dims = (po_var[tt, sp, hh] > 0).nonzero().flatten().tolist()
                print(f"    split {sp} head {hh}: {len(dims)} dims vary, e.g. {dims[:12]}")
