import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import mojo_opset_amd as mo, oracle
from hip_utils import max_ulp_bf16ish
DEV="cuda"
g = torch.Generator().manual_seed(0)
x = torch.randn(512, 256, generator=g).to(torch.bfloat16)
w = (torch.randn(256, 384, generator=g) * 0.1).to(torch.bfloat16)
b = torch.randn(384).to(torch.bfloat16)
ref_op = mo.MojoGemmAllReduce.get_backend_impl("torch")(w, b, True)
w1 = torch.as_tensor(ref_op(x)); w2 = torch.as_tensor(ref_op(x))
print("cpu deterministic", torch.equal(w1, w2))
op = mo.MojoGemmAllReduce.get_backend_impl("hip")(w.to(DEV), b.to(DEV), True)
xd = x.to(DEV)
first = op(xd).cpu()
nd = 0
for i in range(300):
    o = op(xd).cpu()
    if not torch.equal(o, first): nd += 1
print("hip nondeterministic runs:", nd, "of 300; ulp vs cpu:", max_ulp_bf16ish(first, w1, atol=1e-3))
exact = (x.double() @ w.double())
e1 = (exact.to(torch.bfloat16).float() + b.float()).to(torch.bfloat16)
print("hip vs exact-two-round:", max_ulp_bf16ish(first, e1, atol=1e-3), " cpu vs exact-two-round:", max_ulp_bf16ish(w1, e1, atol=1e-3))
bad = (first.float()-w1.float()).abs()
i = bad.argmax(); print("worst idx", divmod(int(i), 384), first.flatten()[i].item(), w1.flatten()[i].item(), e1.flatten()[i].item(), exact.flatten()[i].item(), b[int(i)%384].item())
