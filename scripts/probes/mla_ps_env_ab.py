"""A/B of environment switches of the MLA decode op in ONE process: python3 scripts/probes/mla_ps_env_ab.py VAR=a,b [B ctx]"""
import json, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__  # noqa
from benchmarks.extras import hip, _time_graph
var, vals = sys.argv[1].split("=")
vals = vals.split(",")
b = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ctx = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
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
res = {v: [] for v in vals}
for rnd in range(5):
    for v in vals:
        os.environ[var] = v
        res[v].append(_time_graph(lambda: op(q, ckv, kpe, lens, table), reps=10) * 1e6)
print(json.dumps({"var": var, "B": b, "ctx": ctx, **{v: {"median_us": round(statistics.median(t), 2), "min_us": round(min(t), 2)} for v, t in res.items()}}))
