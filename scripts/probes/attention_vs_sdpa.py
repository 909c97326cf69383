"""Round 5 calibration: paged prefill / decode GQA (Llama-3-8B heads, bf16) against torch's own scaled_dot_product_attention on the
same box (the flash kernel of the PyTorch-ROCm wheel, contiguous [B, H, S, D] tensors, enable_gqa) — the vendor-stack number a
user would get without this backend.  Causal; device times (prefill eager medians over long kernels, decode under graph replay)."""
import json, os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.extras import _time_graph, _time, _paged, hip
dev = torch.device("cuda:0"); dt = torch.bfloat16
hq, hkv, d, page = 32, 8, 128, 16
pre = hip("MojoPagedPrefillGQA")()
for name, b, s in (("4x2048", 4, 2048), ("1x16384", 1, 16384), ("16x512", 16, 512), ("1x8192", 1, 8192)):
    q_lens = [s] * b
    k, v, table = _paged(dev, q_lens, hkv, d, page)
    q = torch.randn(b * s, hq, d, device=dev, dtype=dt)
    cu = torch.tensor([0] + list(torch.tensor(q_lens).cumsum(0).tolist()), dtype=torch.int32, device=dev)
    flops = b * 4.0 * hq * d * (s * s / 2.0)
    t = _time_graph(lambda: pre(q, k, v, cu, table, cu_total_seq_lens=cu, max_q_len=s, max_total_seq_len=s), reps=3, replays=3)
    qs = torch.randn(b, hq, s, d, device=dev, dtype=dt); ks = torch.randn(b, hkv, s, d, device=dev, dtype=dt); vs = torch.randn(b, hkv, s, d, device=dev, dtype=dt)
    fn = lambda: F.scaled_dot_product_attention(qs, ks, vs, is_causal=True, enable_gqa=True)
    fn(); torch.cuda.synchronize()
    t_lib = _time(fn, 5, 2)
    print(json.dumps({"op": "prefill", "case": name, "paged_prefill_us": round(t * 1e6, 1), "paged_prefill_PFLOPs": round(flops / t / 1e15, 3),
                      "torch_sdpa_us": round(t_lib * 1e6, 1), "torch_sdpa_PFLOPs": round(flops / t_lib / 1e15, 3), "time_vs_sdpa": round(t / t_lib, 2)}), flush=True)
    del k, v, table, qs, ks, vs
    torch.cuda.empty_cache()
dec = hip("MojoPagedDecodeGQA")(is_causal=True, gqa_layout="AABB")
for b, ctx in ((64, 4096), (16, 16384), (256, 1024), (8, 32768)):
    k, v, table = _paged(dev, [ctx] * b, hkv, d, page)
    q = torch.randn(b, hq, d, device=dev, dtype=dt)
    lens = torch.full((b,), ctx, dtype=torch.int32, device=dev)
    t = _time_graph(lambda: dec(q, k, v, lens, table), reps=6, replays=3)
    nbytes = b * ctx * hkv * d * 2 * 2
    qs = torch.randn(b, hq, 1, d, device=dev, dtype=dt); ks = torch.randn(b, hkv, ctx, d, device=dev, dtype=dt); vs = torch.randn(b, hkv, ctx, d, device=dev, dtype=dt)
    fn = lambda: F.scaled_dot_product_attention(qs, ks, vs, enable_gqa=True)
    fn(); torch.cuda.synchronize()
    t_lib = _time_graph(fn, reps=4, replays=3)
    print(json.dumps({"op": "decode", "b": b, "ctx": ctx, "paged_decode_us": round(t * 1e6, 1), "paged_decode_TBps": round(nbytes / t / 1e12, 2),
                      "torch_sdpa_us": round(t_lib * 1e6, 1), "torch_sdpa_TBps": round(nbytes / t_lib / 1e12, 2), "time_vs_sdpa": round(t / t_lib, 2)}), flush=True)
    del k, v, table, qs, ks, vs
    torch.cuda.empty_cache()
