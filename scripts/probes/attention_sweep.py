"""Round 5: paged decode / prefill GQA over head geometries, head dims, page sizes and batch shapes away from the benchmarked
ones: achieved HBM rate (decode) / MFMA rate (prefill) under graph replay.  A sanity sweep: lines far below their neighbours are
the fall-offs to look at."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.extras import _time_graph, _paged, hip
from mojo_opset_amd.backends.hip import lib as L
dev = torch.device("cuda:0")
dt = torch.bfloat16
dec = hip("MojoPagedDecodeGQA")(is_causal=True, gqa_layout="AABB")
for hq, hkv, d in ((32, 8, 128), (64, 8, 128), (8, 1, 128), (16, 16, 128), (40, 8, 128), (28, 4, 128), (12, 2, 128), (32, 8, 64), (32, 4, 256), (96, 8, 128)):
    for page in (16, 64, 128):
        for b, ctx in ((64, 4096), (16, 16384), (256, 1024), (8, 32768), (1, 65536), (32, 777)):
            nbytes = b * ctx * hkv * d * 2 * 2
            if nbytes > 12e9 or nbytes < 2e6:
                continue
            try:
                k, v, table = _paged(dev, [ctx] * b, hkv, d, page)
                q = torch.randn(b, hq, d, device=dev, dtype=dt)
                lens = torch.full((b,), ctx, dtype=torch.int32, device=dev)
                t = _time_graph(lambda: dec(q, k, v, lens, table), reps=6, replays=3)
                print(json.dumps({"op": "decode", "hq": hq, "hkv": hkv, "d": d, "page": page, "b": b, "ctx": ctx, "us": round(t * 1e6, 1),
                                  "TBps": round(nbytes / t / 1e12, 2), "form": L.last_launch()}), flush=True)
            except Exception as e:
                print(json.dumps({"op": "decode", "hq": hq, "hkv": hkv, "d": d, "page": page, "b": b, "ctx": ctx, "error": repr(e)[:120]}), flush=True)
            del k, v, table
            torch.cuda.empty_cache()
pre = hip("MojoPagedPrefillGQA")()
for hq, hkv, d in ((32, 8, 128), (64, 8, 128), (8, 1, 128), (16, 16, 128), (28, 4, 128), (32, 8, 64), (32, 4, 256)):
    for page in (16, 128):
        for q_lens, cached in (([2048] * 4, [0] * 4), ([8192], [0]), ([512] * 16, [0] * 16), ([1024], [8192]), ([333, 777, 1500, 64], [100, 0, 2000, 4000])):
            kv = [a + c for a, c in zip(q_lens, cached)]
            try:
                k, v, table = _paged(dev, kv, hkv, d, page)
                q = torch.randn(sum(q_lens), hq, d, device=dev, dtype=dt)
                cu = lambda l: torch.tensor([0] + list(torch.tensor(l).cumsum(0).tolist()), dtype=torch.int32, device=dev)  # noqa: E731
                cu_q, cu_kv = cu(q_lens), cu(kv)
                flops = sum(4.0 * hq * d * (a * b_ - a * a / 2.0) for a, b_ in zip(q_lens, kv))
                t = _time_graph(lambda: pre(q, k, v, cu_q, table, cu_total_seq_lens=cu_kv, max_q_len=max(q_lens), max_total_seq_len=max(kv)), reps=3, replays=3)
                print(json.dumps({"op": "prefill", "hq": hq, "hkv": hkv, "d": d, "page": page, "q": q_lens if len(q_lens) < 5 else f"{len(q_lens)}x{q_lens[0]}", "cached": cached[0],
                                  "us": round(t * 1e6, 1), "PFLOPs": round(flops / t / 1e15, 3), "form": L.last_launch()}), flush=True)
            except Exception as e:
                print(json.dumps({"op": "prefill", "hq": hq, "hkv": hkv, "d": d, "page": page, "q": str(q_lens)[:30], "error": repr(e)[:120]}), flush=True)
            del k, v, table
            torch.cuda.empty_cache()
