"""Fourth sweep: the bookkeeping kernels at large counts (many experts, many tokens, many sequences) and the streaming
operators at awkward sizes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__  # noqa
import mojo_opset_amd as mo
from benchmarks.extras import _time_graph, _time, hip
dev = torch.device("cuda:0")
torch.manual_seed(0)
# MoE bookkeeping at scale
for t, e, k, h in ((8192, 256, 8, 1024), (65536, 64, 8, 256), (4096, 1024, 8, 256), (16384, 8, 2, 1024)):
    x = torch.rand(t, h, device=dev, dtype=torch.bfloat16)
    probs = torch.rand(t, e, device=dev)
    gates, ids = torch.topk(probs, k, dim=-1)
    ids = ids.to(torch.int32).contiguous(); gates = gates.contiguous()
    dd, cd = hip("MojoMoEDispatch")(num_experts=e), hip("MojoMoECombine")()
    td = _time(lambda: dd(x, gates, ids), 5, 2)
    rows, cnt, sg, tok = dd(x, gates, ids)
    buf = torch.empty_like(x)
    tc = _time(lambda: cd(buf, rows, sg, tok), 5, 2)
    byt = t * h * 2 + t * k * h * 2
    print(f"MoE T={t} E={e} k={k} H={h}: dispatch {td*1e6:8.1f} us ({byt/td/1e12:4.2f} TB/s)  combine {tc*1e6:8.1f} us ({byt/tc/1e12:4.2f} TB/s)", flush=True)
    del x, rows, buf
# group GEMM prefix with many groups
for g_ in (1024, 4096):
    w = torch.randn(g_, 64, 128, device=dev, dtype=torch.bfloat16)
    counts = torch.randint(0, 4, (g_,), dtype=torch.int32)
    x = torch.randn(int(counts.sum()), 128, device=dev, dtype=torch.bfloat16)
    gg = hip("MojoGroupGemm")(w, True)
    cd_ = counts.to(dev)
    print(f"group gemm G={g_} (tiny groups): {_time(lambda: gg(x, cd_), 5, 2)*1e6:8.1f} us", flush=True)
# streaming operators at awkward sizes
for rows, dim in ((4096, 5120), (333, 7168), (100000, 128), (8192, 12288)):
    x = torch.randn(rows, dim, device=dev, dtype=torch.bfloat16); r = torch.randn_like(x)
    n = hip("MojoResidualAddRMSNorm")(dim, 1e-5, "pre", dtype=torch.bfloat16, device=dev)
    tn = _time(lambda: n(x, r), 10, 2); ts = _time(lambda: hip("MojoSwiGLU")()(x, r), 10, 2)
    print(f"rows={rows} dim={dim}: norm {tn*1e6:8.1f} us ({4*rows*dim*2/tn/1e12:4.2f} TB/s)  swiglu {ts*1e6:8.1f} us ({3*rows*dim*2/ts/1e12:4.2f} TB/s)", flush=True)
    del x, r
# paged KV store, many sequences at decode; block allocator with many sequences
hkv, d, page = 8, 128, 16
for b in (1024, 4096):
    blocks = b * 4 + 8
    kc = torch.zeros(blocks, hkv, page, d, device=dev, dtype=torch.bfloat16); vc = torch.zeros_like(kc)
    table = torch.randperm(blocks, dtype=torch.int32)[: b * 4].view(b, 4).to(dev)
    ks = torch.randn(b, hkv, d, device=dev, dtype=torch.bfloat16); vs = torch.randn_like(ks)
    ctx = torch.randint(0, 60, (b,), dtype=torch.int32, device=dev)
    st = hip("MojoStorePagedKVCache")()
    print(f"store decode B={b}: {_time_graph(lambda: st(ks, vs, kc, vc, table, None, ctx))*1e6:8.1f} us", flush=True)

