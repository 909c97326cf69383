import sys, torch
sys.path.insert(0, '.')
from benchmarks.extras import hip, _time
dev = torch.device("cuda", 0)
for rows, d in ((8192, 7168), (8192, 8192), (16384, 4096), (65536, 4096)):
    x = torch.randn(rows, d, device=dev, dtype=torch.bfloat16)
    dq = hip("MojoDynamicQuant")()
    t = _time(lambda: dq(x), 50, 10)
    print(rows, d, round(t * 1e6, 1), "us", round(rows * d * 3 / t / 1e9), "GB/s (3 B/elt)", round(rows * d * 2 / t / 1e9), "GB/s read only")
