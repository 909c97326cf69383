"""Round 5: the 256 x 256 kernel's K-major LDS-DMA pieces — 8 whole rows of 128 bytes (GEMM256_LINE_PIECES=1) against 16 rows x 64
bytes (=0; build the library with MOJO_HIP_EXTRA_CXXFLAGS=-DGEMM256_LINE_PIECES=0) on the headline products; device times (graphs)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.extras import _time_graph, hip
from mojo_opset_amd.backends.hip import lib as L
from mojo_opset_amd.backends.hip.operators.gemm import dense_gemm, HIPQuantGemm
dev = torch.device("cuda", 0)
out = {"library": L.load().mojo_hip_version().decode()}
for trans in (True, False):
    m_, k_, n_, g_ = 16384, 4096, 28672, 8
    x = torch.randn(m_, k_, device=dev, dtype=torch.bfloat16)
    w = (torch.randn(g_, n_, k_, device=dev, dtype=torch.bfloat16) if trans else torch.randn(g_, k_, n_, device=dev, dtype=torch.bfloat16))
    counts = torch.full((g_,), m_ // g_, dtype=torch.int32, device=dev)
    op = hip("MojoGroupGemm")(w, trans)
    flops = 2.0 * m_ * k_ * n_
    t = _time_graph(lambda: op(x, counts), reps=2, replays=4)
    out[f"group_16384x4096x28672_G8_{'NK' if trans else 'KN'}"] = {"us": round(t * 1e6, 1), "tflops": round(flops / t / 1e12), "form": L.last_launch()}
    del op, x, w
    torch.cuda.empty_cache()
x = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16); w = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16) * 0.02
t = _time_graph(lambda: dense_gemm(x, w, None, False), reps=3, replays=4)
out["dense_8192^3_NK"] = {"us": round(t * 1e6, 1), "tflops": round(2.0 * 8192 ** 3 / t / 1e12), "form": L.last_launch()}
del x, w
for qd in (torch.int8, torch.float8_e4m3fn):
    m, k, n = 4096, 7168, 36864
    op = HIPQuantGemm(k, n, output_dtype=torch.bfloat16, trans_weight=True, quant_dtype=qd, weight_dtype=qd, device=dev)
    op.weight.copy_(torch.randint(-127, 128, (n, k), dtype=torch.int8, device=dev) if qd == torch.int8 else torch.randn(n, k, device=dev).to(qd))
    op.weight_scale.fill_(0.01)
    xq = torch.randint(-127, 128, (m, k), dtype=torch.int8, device=dev) if qd == torch.int8 else torch.randn(m, k, device=dev).to(qd)
    sc = torch.rand(m, device=dev)
    t = _time_graph(lambda: op(xq, sc), reps=3, replays=4)
    out[f"quant_{'i8' if qd == torch.int8 else 'f8'}_4096x7168x36864_NK"] = {"us": round(t * 1e6, 1), "pops": round(2.0 * m * k * n / t / 1e15, 3), "form": L.last_launch()}
    del op
print(json.dumps(out))
