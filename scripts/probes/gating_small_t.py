"""MoE router at decode sizes: the vector kernel (mfma=0), the matrix-core route (hi/lo split of the fp32 gate weight,
mfma=1), the few-token kernel (small) and the library's own choice (auto); graph-replay timing."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__  # noqa
from benchmarks.extras import _time_graph, hip
dev = torch.device("cuda:0")
for t, e, k, h in ((1, 64, 8, 4096), (8, 64, 8, 4096), (64, 64, 8, 4096), (256, 64, 8, 4096), (64, 256, 8, 7168), (256, 256, 8, 7168), (64, 8, 2, 4096), (64, 32, 4, 2048), (128, 384, 8, 3584)):
    x = torch.rand(t, h, device=dev, dtype=torch.bfloat16)
    g = hip("MojoMoEGating")(hidden_size=h, num_experts=e, top_k=k).to(dev)
    with torch.no_grad():
        g.gate_weight.copy_(torch.randn(h, e) * 0.02)
    row = []
    ref = None
    for mode in ("0", "1", "small", "auto"):
        os.environ.pop("MOJO_HIP_GATING_MFMA", None)
        os.environ.pop("MOJO_HIP_GATING_SMALL", None)
        if mode in ("0", "1"):
            os.environ["MOJO_HIP_GATING_MFMA"] = mode
            os.environ["MOJO_HIP_GATING_SMALL"] = "0"
        elif mode == "small":
            os.environ["MOJO_HIP_GATING_SMALL"] = "1"
        idx, gates = g(x)
        if ref is None:
            ref = (idx.clone(), gates.clone())
        agree = (idx == ref[0]).float().mean().item()
        gerr = (gates - ref[1]).abs().max().item() if agree == 1.0 else float("nan")
        tt = _time_graph(lambda: g(x), reps=10, replays=5)
        row.append(f"{mode}: {tt * 1e6:6.1f} us (top-k agreement {agree:.3f}, gate diff {gerr:.2g})")
    print(f"T={t} E={e} k={k} H={h}: " + "; ".join(row), flush=True)
os.environ.pop("MOJO_HIP_GATING_MFMA", None)
os.environ.pop("MOJO_HIP_GATING_SMALL", None)
