"""A/B in one process: paged decode GQA on small grids, split + merge form vs the grouped form (MOJO_HIP_DECODE_GROUPED, read per
call).  Cases: 8q/1kv B 64 ctx 4096 (Llama-3-70B under TP 8), 32q/8kv B 8 ctx 4096, 64q/8kv B 8 ctx 8192, 8q/1kv B 16 ctx 16384."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.extras import _paged, _time_graph, hip  # noqa: E402

dev = torch.device("cuda", 0)
page = 16
for hq, hkv, d, bsz, ctx in ((8, 1, 128, 64, 4096), (32, 8, 128, 8, 4096), (64, 8, 128, 8, 8192), (8, 1, 128, 16, 16384), (32, 8, 64, 8, 8192)):
    k, v, table = _paged(dev, [ctx] * bsz, hkv, d, page)
    q = torch.randn(bsz, hq, d, device=dev, dtype=torch.bfloat16)
    lens = torch.full((bsz,), ctx, dtype=torch.int32, device=dev)
    op = hip("MojoPagedDecodeGQA")(is_causal=True, gqa_layout="AABB")
    outs, row = {}, []
    for rep in range(2):
        for mode in ("0", "1"):
            os.environ["MOJO_HIP_DECODE_GROUPED"] = mode
            if mode == "1":
                os.environ["MOJO_HIP_DECODE_MFMA"] = "1"
            outs[mode] = op(q, k, v, lens, table, max_total_seq_len=ctx).float()
            t = _time_graph(lambda: op(q, k, v, lens, table, max_total_seq_len=ctx), reps=8)
            os.environ.pop("MOJO_HIP_DECODE_MFMA", None)
            row.append(f"grouped={mode}: {t * 1e6:6.1f} us")
    err = (outs["0"] - outs["1"]).abs().max().item()
    gb = bsz * ctx * hkv * d * 2 * 2 / 1e9
    print(f"{hq}q/{hkv}kv d{d} B{bsz} ctx{ctx} ({gb * 1e3:.0f} MB): " + "  ".join(row) + f"   max |diff| {err:.3g}", flush=True)
os.environ.pop("MOJO_HIP_DECODE_GROUPED", None)
