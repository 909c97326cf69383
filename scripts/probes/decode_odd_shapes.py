"""Sanity sweep of shape classes away from the benchmarked ones: time and achieved HBM rate (graph replay), to catch a
kernel that is absurdly slow at some size (the router at decode sizes was: 204 us for a 7 MB problem)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__  # noqa
from benchmarks.extras import _time_graph, _paged, hip
dev = torch.device("cuda:0")
hq, hkv, d, page = 32, 8, 128, 16
op = hip("MojoPagedDecodeGQA")(is_causal=True, gqa_layout="AABB")
for b, ctx in ((1, 16384), (1, 131072), (4, 32768), (8, 2048), (256, 512), (512, 1024), (3, 100), (128, 300)):
    k, v, table = _paged(dev, [ctx] * b, hkv, d, page)
    q = torch.randn(b, hq, d, device=dev, dtype=torch.bfloat16)
    lens = torch.full((b,), ctx, dtype=torch.int32, device=dev)
    t = _time_graph(lambda: op(q, k, v, lens, table), reps=10, replays=3)
    byt = b * ctx * hkv * d * 2 * 2
    print(f"decode GQA B={b} ctx={ctx}: {t*1e6:8.1f} us  {byt/t/1e12:5.2f} TB/s", flush=True)
    del k, v, table
    torch.cuda.empty_cache()
for t_, e, k_, h in ((300, 64, 8, 4096), (511, 256, 8, 7168), (512, 256, 8, 7168), (2048, 64, 8, 4096), (300, 8, 2, 4096), (16, 8, 2, 4096)):
    x = torch.rand(t_, h, device=dev, dtype=torch.bfloat16)
    g = hip("MojoMoEGating")(hidden_size=h, num_experts=e, top_k=k_).to(dev)
    with torch.no_grad():
        g.gate_weight.copy_(torch.randn(h, e) * 0.02)
    t = _time_graph(lambda: g(x), reps=10, replays=3)
    print(f"router T={t_} E={e} H={h}: {t*1e6:8.1f} us", flush=True)
for rows, dim in ((64, 4096), (64, 7168), (1, 8192)):
    x = torch.randn(rows, dim, device=dev, dtype=torch.bfloat16)
    r = torch.randn(rows, dim, device=dev, dtype=torch.bfloat16)
    n = hip("MojoResidualAddRMSNorm")(dim, 1e-5, "pre", dtype=torch.bfloat16, device=dev)
    dq = hip("MojoDynamicQuant")()
    print(f"rows={rows} dim={dim}: norm {_time_graph(lambda: n(x, r))*1e6:6.1f} us, dynamic quant {_time_graph(lambda: dq(x))*1e6:6.1f} us, swiglu {_time_graph(lambda: hip('MojoSwiGLU')()(x, r))*1e6:6.1f} us", flush=True)
