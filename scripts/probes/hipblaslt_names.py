import sys, torch
dev = torch.device("cuda", 0)
for m, k, n in ((4096, 4096, 28672), (4096, 14336, 4096), (2048, 4096, 4096), (1024, 4096, 4096), (512, 4096, 4096), (2048, 4096, 28672), (16384, 4096, 4096)):
    x = torch.randn(m, k, device=dev, dtype=torch.bfloat16); w = torch.randn(n, k, device=dev, dtype=torch.bfloat16)
    for _ in range(3): torch.nn.functional.linear(x, w)
    torch.cuda.synchronize()
