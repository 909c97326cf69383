"""Round 5: one dense product in a loop for `rocprofv3 --kernel-trace --stats` (scripts/profile_r5_gemm_splitk.sh): the kernels a
K-split product of the 128-row tiles launches (gemm128_kernel + the finalize) and their durations, warm (one weight) or cold
(argument `cold`: every call another copy of the weight, copies x bytes >= 768 MB).  usage: gemm_splitk_trace.py M K N [cold]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mojo_opset_amd.backends.hip import lib as L
from mojo_opset_amd.backends.hip.operators.gemm import dense_gemm
m, k, n = (int(v) for v in sys.argv[1:4])
cold = len(sys.argv) > 4 and sys.argv[4] == "cold"
dev = torch.device("cuda", 0)
copies = max(2, -(-768 * 2 ** 20 // (k * n * 2))) if cold else 1
ws = [torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(copies)]
x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
for i in range(20):
    dense_gemm(x, ws[i % copies], None, False)
torch.cuda.synchronize()
for i in range(400):
    dense_gemm(x, ws[i % copies], None, False)
torch.cuda.synchronize()
print(m, k, n, "cold" if cold else "warm", L.last_launch())
