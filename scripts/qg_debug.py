import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mojo_opset_amd as mo, oracle
from oracle import quant_gemm_formula
dev = "cuda"
def q(x):
    s = x.abs().amax(-1).clamp_min(1e-8)/127.0
    return torch.clamp(torch.round(x/s[:,None]), -128, 127).to(torch.int8), s
torch.manual_seed(0)
m,k,n = 32,4096,11008
xq,xs = q(torch.randn(m,k)); wq,ws = q(torch.randn(n,k))
for trans in (False, True):
    op = mo.MojoQuantGemm.get_backend_impl("hip")(k, n, output_dtype=torch.float32, trans_weight=trans, device=dev)
    op.weight.copy_(wq if trans else wq.t()); op.weight_scale.copy_(ws.to(torch.bfloat16))
    exp = quant_gemm_formula(xq, wq.t(), xs, ws.to(torch.bfloat16), torch.float32)
    out = op(xq.to(dev), xs.to(dev)).cpu()
    bad = (out != exp)
    print("trans", trans, "bad", int(bad.sum()), "rows", bad.any(1).nonzero().flatten().tolist()[:10], "cols", sorted(set((bad.any(0).nonzero().flatten()//256).tolist()))[:20])
# odd number of K tiles in the bf16 grouped gemm
for kk in (192, 320, 448):
    x = torch.randint(-3,4,(300,kk)).to(torch.bfloat16); w = torch.randint(-3,4,(1,kk,512)).to(torch.bfloat16)
    got = mo.MojoGroupGemm.get_backend_impl("hip")(w.to(dev), False)(x.to(dev), torch.tensor([300],dtype=torch.int32,device=dev)).cpu().float()
    print("K", kk, "exact", torch.equal(got, x.float()@w[0].float()))
