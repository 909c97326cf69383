"""A/B in one process: MojoPagedPrefillMLA with and without the start stagger of its decompression GEMM (MOJO_HIP_GEMM_STAGGER),
eager timing (the bench's) and graph replay (launch overhead out of the way)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.extras import _time, _time_graph, hip  # noqa: E402

dev = torch.device("cuda", 0)
h, nope, rope, vd, r, page = 128, 128, 64, 128, 512, 16
op = hip("MojoPagedPrefillMLA")(h, nope, rope, vd, r).to(torch.bfloat16).to(dev)
with torch.no_grad():
    op.kv_b_proj.copy_(torch.randn_like(op.kv_b_proj) * 0.02)
for name, (q_lens, cached) in {"4x512_nocache": ([512] * 4, [0] * 4), "4x512_cached2048": ([512] * 4, [2048] * 4)}.items():
    kv = [a + b for a, b in zip(q_lens, cached)]
    need = [(n + page - 1) // page for n in kv]
    total = sum(need) + 4
    ckv = torch.randn(total, 1, page, r, device=dev, dtype=torch.bfloat16)
    kpe = torch.randn(total, 1, page, rope, device=dev, dtype=torch.bfloat16)
    table = torch.randperm(total, dtype=torch.int32)[: sum(need)].view(len(kv), need[0]).to(dev)
    cu = lambda l: torch.tensor([0] + torch.tensor(l).cumsum(0).tolist(), dtype=torch.int32, device=dev)  # noqa: E731
    cu_q, cu_kv = cu(q_lens), cu(kv)
    q = torch.randn(sum(q_lens), h, nope + rope, device=dev, dtype=torch.bfloat16)
    fn = lambda: op(q, ckv, kpe, cu_q, table, cu_total_seq_lens=cu_kv, max_total_seq_len=max(kv))  # noqa: E731
    for rep in range(2):
        for st in ("0", None):
            if st is None:
                os.environ.pop("MOJO_HIP_GEMM_STAGGER", None)
            else:
                os.environ["MOJO_HIP_GEMM_STAGGER"] = st
            te = _time(fn, 5, 1)
            tg = _time_graph(fn, reps=4, replays=5)
            print(f"{name} stagger {st or 'auto'}: eager {te * 1e6:7.1f} us, graph replay {tg * 1e6:7.1f} us", flush=True)
