"""Fifth sweep: other head geometries (MHA, G = 8, head_dim 64 / 96) for decode and prefill GQA, and MLA prefill shape classes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__  # noqa
from benchmarks.extras import _time_graph, _time, _paged, hip
dev = torch.device("cuda:0")
page = 16
dec = hip("MojoPagedDecodeGQA")(is_causal=True, gqa_layout="AABB")
pf = hip("MojoPagedPrefillGQA")()
for hq, hkv, d in ((32, 32, 128), (64, 8, 128), (32, 8, 64), (32, 4, 96), (16, 2, 128), (8, 1, 128)):
    b, ctx = 64, 4096
    k, v, table = _paged(dev, [ctx] * b, hkv, d, page)
    q = torch.randn(b, hq, d, device=dev, dtype=torch.bfloat16)
    lens = torch.full((b,), ctx, dtype=torch.int32, device=dev)
    t = _time_graph(lambda: dec(q, k, v, lens, table), reps=5, replays=3)
    line = f"Hq={hq} Hkv={hkv} D={d}: decode B64 ctx4096 {t*1e6:8.1f} us {b*ctx*hkv*d*4/t/1e12:5.2f} TB/s"
    del k, v, table
    q_lens = [2048] * 4
    k, v, table = _paged(dev, q_lens, hkv, d, page)
    qq = torch.randn(sum(q_lens), hq, d, device=dev, dtype=torch.bfloat16)
    cu = torch.tensor([0, 2048, 4096, 6144, 8192], dtype=torch.int32, device=dev)
    flops = sum(4.0 * hq * d * (a * a / 2.0) for a in q_lens)
    tp = _time(lambda: pf(qq, k, v, cu, table, cu_total_seq_lens=cu, max_q_len=2048, max_total_seq_len=2048), 5, 2)
    print(line + f" | prefill 4x2048 {tp*1e6:8.1f} us {flops/tp/1e12:7.1f} TF", flush=True)
    del k, v, table, qq
    torch.cuda.empty_cache()
h, nope, rope, vd, r = 128, 128, 64, 128, 512
mp = hip("MojoPagedPrefillMLA")(h, nope, rope, vd, r).to(torch.bfloat16).to(dev)
with torch.no_grad():
    mp.kv_b_proj.copy_(torch.randn_like(mp.kv_b_proj) * 0.02)
for name, q_lens, cached in (("64x64", [64] * 64, [0] * 64), ("1x8192", [8192], [0]), ("1x512 + 16384 cached", [512], [16384]), ("16x256 + 1024 cached", [256] * 16, [1024] * 16)):
    kv = [a + c for a, c in zip(q_lens, cached)]
    need = [(n + page - 1) // page for n in kv]
    total = sum(need) + 4
    ckv = torch.randn(total, 1, page, r, device=dev, dtype=torch.bfloat16)
    kpe = torch.randn(total, 1, page, rope, device=dev, dtype=torch.bfloat16)
    perm = torch.randperm(total, dtype=torch.int32)
    table = torch.full((len(kv), max(need)), -1, dtype=torch.int32)
    at = 0
    for i, n in enumerate(need):
        table[i, :n] = perm[at: at + n]; at += n
    table = table.to(dev)
    q = torch.randn(sum(q_lens), h, nope + rope, device=dev, dtype=torch.bfloat16)
    cu = lambda l: torch.tensor([0] + list(torch.tensor(l).cumsum(0).tolist()), dtype=torch.int32, device=dev)  # noqa: E731
    cu_q, cu_kv = cu(q_lens), cu(kv)
    flops = sum(2.0 * h * (nope + rope + vd) * (a * c2 - a * a / 2.0) for a, c2 in zip(q_lens, kv))
    t = _time(lambda: mp(q, ckv, kpe, cu_q, table, cu_total_seq_lens=cu_kv), 3, 1)
    print(f"MLA prefill {name}: {t*1e6:9.1f} us  {flops/t/1e12:7.1f} TF (attention FLOPs only)", flush=True)
    del ckv, kpe, q
    torch.cuda.empty_cache()
