"""Round 5 calibration: the streaming operators against torch's own kernels for the same math on the same box (what a model file
runs without this backend): F.rms_norm, residual add + rms_norm, silu(gate) * up, rotate-half RoPE; bf16, device times (graphs)."""
import json, os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.extras import _time_graph, hip
dev = torch.device("cuda:0"); dt = torch.bfloat16
for rows, d in ((64, 4096), (8192, 4096), (65536, 4096), (8192, 7168)):
    x, r = torch.randn(rows, d, device=dev, dtype=dt), torch.randn(rows, d, device=dev, dtype=dt)
    w = torch.randn(d, device=dev, dtype=dt)
    n1 = hip("MojoRMSNorm")(d, 1e-5).to(dt).to(dev)
    n2 = hip("MojoResidualAddRMSNorm")(d, 1e-5, "pre", dtype=dt, device=dev)
    act = hip("MojoSwiGLU")()
    rec = {"rows": rows, "dim": d}
    rec["rmsnorm_us"] = round(_time_graph(lambda: n1(x), reps=8, replays=3) * 1e6, 1)
    rec["torch_rms_norm_us"] = round(_time_graph(lambda: F.rms_norm(x, (d,), w, 1e-5), reps=8, replays=3) * 1e6, 1)
    rec["residual_add_rmsnorm_us"] = round(_time_graph(lambda: n2(x, r), reps=8, replays=3) * 1e6, 1)
    def torch_res():
        s_ = x + r
        return F.rms_norm(s_, (d,), w, 1e-5), s_
    rec["torch_add_then_rms_norm_us"] = round(_time_graph(torch_res, reps=8, replays=3) * 1e6, 1)
    rec["swiglu_us"] = round(_time_graph(lambda: act(x, r), reps=8, replays=3) * 1e6, 1)
    rec["torch_silu_mul_us"] = round(_time_graph(lambda: F.silu(x) * r, reps=8, replays=3) * 1e6, 1)
    print(json.dumps(rec), flush=True)
rope = hip("MojoApplyRoPE")()
for tokens in (64, 8192):
    q = torch.randn(1, tokens, 32, 128, device=dev, dtype=dt); k = torch.randn(1, tokens, 8, 128, device=dev, dtype=dt)
    cos, sin = torch.randn(tokens, 128, device=dev), torch.randn(tokens, 128, device=dev)
    def torch_rope():
        c, s_ = cos[None, :, None, :], sin[None, :, None, :]
        def rot(t):
            t1, t2 = t[..., :64], t[..., 64:]
            return torch.cat((-t2, t1), dim=-1)
        return (q.float() * c + rot(q.float()) * s_).to(dt), (k.float() * c + rot(k.float()) * s_).to(dt)
    print(json.dumps({"tokens": tokens, "apply_rope_us": round(_time_graph(lambda: rope(q, k, cos, sin, head_first=False), reps=8, replays=3) * 1e6, 1),
                      "torch_rotate_half_us": round(_time_graph(torch_rope, reps=4, replays=3) * 1e6, 1)}), flush=True)
