"""A/B of the r = 512 latent kernels in ONE process (interleaved rounds, median and min per arm):
    python3 scripts/probes/mla_kernels_ab.py [B ctx] [arms...]"""
import json, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__  # noqa
from benchmarks.extras import hip, _time_graph
b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
arms = sys.argv[3:] or ["oct", "ps"]
dev = torch.device("cuda:0")
h, nope, rope, vd, r, page = 128, 128, 64, 128, 512, 16
op = hip("MojoPagedDecodeMLA")(h, nope, rope, vd, r).to(torch.bfloat16).to(dev)
with torch.no_grad():
    op.kv_b_proj.copy_(torch.randn_like(op.kv_b_proj) * 0.02)
pages = ctx // page
total = b * pages + 4
ckv = torch.randn(total, 1, page, r, device=dev, dtype=torch.bfloat16)
kpe = torch.randn(total, 1, page, rope, device=dev, dtype=torch.bfloat16)
table = torch.randperm(total, dtype=torch.int32)[: b * pages].view(b, pages).to(dev)
lens = torch.full((b,), ctx, dtype=torch.int32, device=dev)
q = torch.randn(b, h, nope + rope, device=dev, dtype=torch.bfloat16)
res = {a: [] for a in arms}
outs = {}
for rnd in range(5):
    for a in arms:
        os.environ["MOJO_HIP_MLA_KERNEL"] = a
        outs[a] = op(q, ckv, kpe, lens, table)
        res[a].append(_time_graph(lambda: op(q, ckv, kpe, lens, table), reps=10) * 1e6)
ref = outs[arms[0]].float()
print(json.dumps({"B": b, "ctx": ctx, **{a: {"median_us": round(statistics.median(v), 2), "min_us": round(min(v), 2),
                                            "max_abs_diff_vs_first": float((outs[a].float() - ref).abs().max())} for a, v in res.items()}}))
