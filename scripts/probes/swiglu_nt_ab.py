"""A/B in one process: SwiGLU with non-temporal loads / stores (MOJO_HIP_STREAM_NT bits: 1 loads, 2 stores; read per call)."""
import os, sys, torch
sys.path.insert(0, ".")
from benchmarks import extras as X
dev = torch.device("cuda:0")
for dt, rows in ((torch.bfloat16, 65536), (torch.bfloat16, 2048), (torch.float32, 2048), (torch.float32, 32768)):
    d = 4096
    x = torch.randn(rows, d, device=dev, dtype=dt); r = torch.randn(rows, d, device=dev, dtype=dt)
    op = X.hip("MojoSwiGLU")()
    res = {}
    for rnd in range(3):
        for nt in ("0", "1", "2", "3"):
            os.environ["MOJO_HIP_STREAM_NT"] = nt
            res.setdefault(nt, []).append(X._time(lambda: op(x, r), 20, 3))
    nb = 3 * rows * d * x.element_size()
    print(dt, rows, {k: (round(min(v) * 1e6, 1), round(nb / min(v) / 1e12, 2)) for k, v in res.items()}, flush=True)
