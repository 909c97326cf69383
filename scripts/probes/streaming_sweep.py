"""Round 5: the streaming operators (RMSNorm family, SwiGLU, RoPE, dynamic quant, norm + quant) over rows x dims away from the
benchmarked ones: achieved HBM rate under graph replay.  Lines far below their neighbours are the fall-offs to look at."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.extras import _time_graph, hip
dev = torch.device("cuda:0")
def rec(op, rows, d, dtype, t, nbytes, extra=""):
    print(json.dumps({"op": op, "rows": rows, "dim": d, "dtype": str(dtype)[6:], "us": round(t * 1e6, 1), "TBps": round(nbytes / t / 1e12, 2), "MB": round(nbytes / 1e6, 1), "x": extra}), flush=True)
for dtype in (torch.bfloat16, torch.float16, torch.float32):
    es = torch.tensor([], dtype=dtype).element_size()
    for d in (128, 1000, 1536, 2048, 4096, 5120, 7168, 8192, 16384):
        for rows in (1, 64, 300, 4096, 32768):
            if rows * d * es > 1.2e9 or rows * d * es < 8192:
                continue
            x, r = torch.randn(rows, d, device=dev, dtype=dtype), torch.randn(rows, d, device=dev, dtype=dtype)
            try:
                n1 = hip("MojoRMSNorm")(d, 1e-5, dtype=dtype, device=dev) if "dtype" in hip("MojoRMSNorm").__init__.__code__.co_varnames else hip("MojoRMSNorm")(d, 1e-5).to(dtype).to(dev)
                rec("rmsnorm", rows, d, dtype, _time_graph(lambda: n1(x), reps=8, replays=3), 2 * rows * d * es)
            except Exception as e:
                print(json.dumps({"op": "rmsnorm", "rows": rows, "dim": d, "error": repr(e)[:100]}), flush=True)
            n2 = hip("MojoResidualAddRMSNorm")(d, 1e-5, "pre", dtype=dtype, device=dev)
            rec("residual_add_rmsnorm", rows, d, dtype, _time_graph(lambda: n2(x, r), reps=8, replays=3), 4 * rows * d * es)
            act = hip("MojoSwiGLU")()
            rec("swiglu", rows, d, dtype, _time_graph(lambda: act(x, r), reps=8, replays=3), 3 * rows * d * es)
            if dtype != torch.float32:
                dq = hip("MojoDynamicQuant")()
                rec("dynamic_quant", rows, d, dtype, _time_graph(lambda: dq(x), reps=8, replays=3), rows * d * (es + 1))
            del x, r
rope = hip("MojoApplyRoPE")()
for tokens in (64, 1024, 8192, 65536):
    for hq, hkv, d in ((32, 8, 128), (64, 8, 128), (28, 4, 128), (32, 8, 64), (128, 128, 64)):
        for head_first in (False, True):
            if tokens * (hq + hkv) * d * 2 > 2e9:
                continue
            q = torch.randn((1, hq, tokens, d) if head_first else (1, tokens, hq, d), device=dev, dtype=torch.bfloat16)
            k = torch.randn((1, hkv, tokens, d) if head_first else (1, tokens, hkv, d), device=dev, dtype=torch.bfloat16)
            cos, sin = torch.randn(tokens, d, device=dev), torch.randn(tokens, d, device=dev)
            t = _time_graph(lambda: rope(q, k, cos, sin, head_first=head_first), reps=8, replays=3)
            rec("apply_rope", tokens, d, torch.bfloat16, t, 2 * (q.numel() + k.numel()) * 2 + 2 * cos.numel() * 4, f"hq{hq} hkv{hkv} head_first={head_first}")
