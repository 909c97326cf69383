"""Second sweep of shape classes away from the benchmarked ones: MLA decode at small batches, GQA prefill with many short /
one very long / chunked sequences, grouped GEMM with mid-sized groups."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__  # noqa
from benchmarks.extras import _time_graph, _time, _paged, hip
dev = torch.device("cuda:0")
# ---- MLA decode, DeepSeek-V3 dims
h, nope, rope, vd, r, page = 128, 128, 64, 128, 512, 16
op = hip("MojoPagedDecodeMLA")(h, nope, rope, vd, r).to(torch.bfloat16).to(dev)
with torch.no_grad():
    op.kv_b_proj.copy_(torch.randn_like(op.kv_b_proj) * 0.02)
for b, ctx in ((1, 4096), (1, 32768), (4, 8192), (16, 4096), (128, 1024), (256, 2048)):
    pages = ctx // page
    total = b * pages + 4
    ckv = torch.randn(total, 1, page, r, device=dev, dtype=torch.bfloat16)
    kpe = torch.randn(total, 1, page, rope, device=dev, dtype=torch.bfloat16)
    table = torch.randperm(total, dtype=torch.int32)[: b * pages].view(b, pages).to(dev)
    lens = torch.full((b,), ctx, dtype=torch.int32, device=dev)
    q = torch.randn(b, h, nope + rope, device=dev, dtype=torch.bfloat16)
    t = _time_graph(lambda: op(q, ckv, kpe, lens, table), reps=5, replays=3)
    flops = 2.0 * b * h * ctx * (2 * r + rope)
    print(f"MLA decode B={b} ctx={ctx}: {t*1e6:8.1f} us  {b*ctx*(r+rope)*2/t/1e12:5.2f} TB/s  {flops/t/1e12:6.1f} TF", flush=True)
    del ckv, kpe, table
    torch.cuda.empty_cache()
# ---- GQA prefill
hq, hkv, d = 32, 8, 128
pf = hip("MojoPagedPrefillGQA")()
for name, q_lens, cached in (("64x128", [128] * 64, [0] * 64), ("256x32", [32] * 256, [0] * 256), ("1x65536", [65536], [0]),
                             ("chunk 1x512 + 16384 cached", [512], [16384]), ("8x512 + 4096 cached", [512] * 8, [4096] * 8), ("1x100", [100], [0])):
    kv = [a + c for a, c in zip(q_lens, cached)]
    k, v, table = _paged(dev, kv, hkv, d, page)
    q = torch.randn(sum(q_lens), hq, d, device=dev, dtype=torch.bfloat16)
    cu = lambda l: torch.tensor([0] + list(torch.tensor(l).cumsum(0).tolist()), dtype=torch.int32, device=dev)  # noqa: E731
    cu_q, cu_kv = cu(q_lens), cu(kv)
    flops = sum(4.0 * hq * d * (a * c - a * a / 2.0) for a, c in zip(q_lens, kv))
    t = _time(lambda: pf(q, k, v, cu_q, table, cu_total_seq_lens=cu_kv, max_q_len=max(q_lens), max_total_seq_len=max(kv)), 5, 2)
    print(f"prefill {name}: {t*1e6:9.1f} us  {flops/t/1e12:7.1f} TF", flush=True)
    del k, v, table, q
    torch.cuda.empty_cache()
# ---- grouped GEMM, mid-sized ragged groups ([G, N, K] weights)
for g_, rows, k_, n_ in ((64, 128, 4096, 4096), (64, 96, 4096, 4096), (16, 300, 4096, 14336), (256, 40, 7168, 4096)):
    w = torch.randn(g_, n_, k_, device=dev, dtype=torch.bfloat16)
    counts = torch.randint(rows // 2, rows * 3 // 2 + 1, (g_,), dtype=torch.int32)
    m = int(counts.sum())
    x = torch.randn(m, k_, device=dev, dtype=torch.bfloat16)
    gg = hip("MojoGroupGemm")(w, True)
    cd = counts.to(dev)
    t = _time(lambda: gg(x, cd), 5, 2)
    print(f"group gemm G={g_} rows~{rows} K={k_} N={n_}: {t*1e6:9.1f} us  {2.0*m*k_*n_/t/1e12:7.1f} TF  weights {g_*k_*n_*2/t/1e12:5.2f} TB/s", flush=True)
    del w, x
    torch.cuda.empty_cache()
