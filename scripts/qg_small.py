import torch, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mojo_opset_amd as mo
from benchmarks.extras import _time, hip
dev = torch.device("cuda", 0)
res = {}
for qd, name in ((torch.int8, "int8"), (torch.float8_e4m3fn, "fp8")):
    for trans in (True, False):
        for m, k, n in ((1, 7168, 4096), (32, 7168, 4096), (128, 7168, 4096), (32, 18432, 7168), (32, 2048, 7168)):
            op = hip("MojoQuantGemm")(k, n, trans_weight=trans, quant_dtype=qd, weight_dtype=qd, device=dev)
            shape = (n, k) if trans else (k, n)
            if qd == torch.int8:
                op.weight.copy_(torch.randint(-127, 128, shape, dtype=torch.int8, device=dev)); x = torch.randint(-127, 128, (m, k), dtype=torch.int8, device=dev)
            else:
                op.weight.copy_(torch.randn(shape, device=dev).to(qd)); x = torch.randn(m, k, device=dev).to(qd)
            op.weight_scale.fill_(0.01); s = torch.rand(m, device=dev)
            t = _time(lambda: op(x, s), 30, 5)
            res[f"{name}_{'NK' if trans else 'KN'}_{m}x{k}x{n}"] = {"us": round(t*1e6, 1), "weight_GBps": round(k*n/t/1e9, 1)}
print(json.dumps(res, indent=1))
