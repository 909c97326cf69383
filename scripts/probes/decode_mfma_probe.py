"""Probe: which (token, 16-byte chunk) of K ends up in which token's score in the matrix-core decode kernel?
q = ones, K = 1/8 on one (token, chunk), V = one-hot rows: out[d] = P[d], log P shows who received the +1."""
import math, sys, torch
sys.path.insert(0, ".")
import mojo_opset_amd as mo
D, G, page, n = 128, 8, 16, 16
op = mo.MojoPagedDecodeGQA.get_backend_impl("hip", strict=True)()
q = torch.ones(1, G, D).bfloat16().cuda()
vc = torch.zeros(2, 1, page, D).bfloat16()
for t in range(n):
    vc[1, 0, t, t] = 1.0
vc = vc.cuda()
table = torch.tensor([[1]], dtype=torch.int32).cuda()
lens = torch.tensor([n], dtype=torch.int32).cuda()
for c0 in range(16):
    row = []
    for t0 in range(16):
        kc = torch.zeros(2, 1, page, D).bfloat16()
        kc[1, 0, t0, 8 * c0: 8 * c0 + 8] = 0.125
        out = op(q, kc.cuda(), vc, lens, table, softmax_scale=1.0).float().cpu()[0, 0, :n]
        s = torch.log(out); s = s - s.min()
        row.append("".join(str(int(round(x))) for x in s.tolist()))
    print("chunk", c0, " ".join(row))
