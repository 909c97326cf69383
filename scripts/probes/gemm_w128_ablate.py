import os, sys, torch
sys.path.insert(0, ".")
from benchmarks import extras as X
dev = torch.device("cuda:0")
for arm in ("0", "1", "5", "1", "5"):
    os.environ["MOJO_HIP_GEMM_W128"] = arm
    r = X.group_gemm_case(dev, 16384, 4096, 28672, 8, True)
    print(arm, round(r["us"], 1), round(r["tflops"]), flush=True)
