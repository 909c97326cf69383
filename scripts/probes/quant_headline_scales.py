"""Round 5: QuantGemm headline shapes (M 4096; [N,K] weights): device time per call.  Run on the shipped library and on one built
with MOJO_HIP_EXTRA_CXXFLAGS=-DQG_NO_SCALE_LOADS (the 256 x 256 kernel's row-staged epilogue multiplies by constants instead of
fetching its scales: wrong results, timing only) — the difference is what the epilogue's scale fetches cost."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.extras import _time_graph
from mojo_opset_amd.backends.hip import lib as L
from mojo_opset_amd.backends.hip.operators.gemm import HIPQuantGemm
dev = torch.device("cuda", 0)
out = {"library": L.load().mojo_hip_version().decode()}
for qd in (torch.int8, torch.float8_e4m3fn):
    for m, k, n in ((4096, 7168, 36864), (4096, 18432, 7168), (8192, 4096, 4096)):
        op = HIPQuantGemm(k, n, output_dtype=torch.bfloat16, trans_weight=True, quant_dtype=qd, weight_dtype=qd, device=dev)
        op.weight.copy_(torch.randint(-127, 128, (n, k), dtype=torch.int8, device=dev) if qd == torch.int8 else torch.randn(n, k, device=dev).to(qd))
        op.weight_scale.fill_(0.01)
        x = torch.randint(-127, 128, (m, k), dtype=torch.int8, device=dev) if qd == torch.int8 else torch.randn(m, k, device=dev).to(qd)
        sc = torch.rand(m, device=dev)
        t = _time_graph(lambda: op(x, sc), reps=4)
        out[f"{'i8' if qd == torch.int8 else 'f8'}_M{m}_K{k}_N{n}"] = {"us": round(t * 1e6, 1), "pops": round(2.0 * m * k * n / t / 1e15, 3), "form": L.last_launch()}
        del op
print(json.dumps(out))
