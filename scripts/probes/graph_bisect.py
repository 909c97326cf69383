"""Capture growing prefixes of the MoE chain in a HIP graph and replay (diagnostic for tests/test_hip_graph.py)."""
import subprocess
import sys

BODY = r'''
import sys, torch
sys.path.insert(0, '.')
import mojo_opset_amd as mo
DEV = 'cuda'
def hip(n): return getattr(mo, n).get_backend_impl('hip', strict=True)
depth = int(sys.argv[1])
tokens, hidden, inter, experts, k = 64, 512, 1024, 8, 2
x = torch.rand(tokens, hidden, dtype=torch.bfloat16, device=DEV)
gating = hip("MojoMoEGating")(hidden_size=hidden, num_experts=experts, top_k=k).to(DEV)
ffn = hip("MojoExperts")(num_experts=experts, hidden_size=hidden, intermediate_size=inter).to(torch.bfloat16).to(DEV)
with torch.no_grad():
    gating.gate_weight.normal_(std=0.05); ffn.up_proj_weight.normal_(std=0.05); ffn.down_proj_weight.normal_(std=0.05)
dispatch, combine = hip("MojoMoEDispatch")(num_experts=experts), hip("MojoMoECombine")()
quant = hip("MojoDynamicQuant")()
qgemm = hip("MojoQuantGemm")(hidden, 256).to(DEV)
with torch.no_grad():
    qgemm.weight.copy_(torch.randint(-127, 128, (hidden, 256), dtype=torch.int8)); qgemm.weight_scale.copy_(torch.rand(256) * 0.01)
def step():
    idx, gates = gating(x)
    if depth == 1: return idx
    rows, counts, sg, tok = dispatch(x, gates, idx)
    if depth == 2: return rows
    e = ffn(rows, counts)
    if depth == 3: return e
    y = combine(x, e, sg, tok)
    if depth == 4: return y
    y_q, s = quant(y)
    if depth == 5: return y_q
    return qgemm(y_q, s.reshape(-1))
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    step(); step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = step()
for i in range(3):
    x.copy_(torch.rand(tokens, hidden))
    g.replay(); torch.cuda.synchronize()
    got = out.clone(); want = step(); torch.cuda.synchronize()
    assert torch.equal(got, want), "mismatch"
print("depth", depth, "replay ok")
'''
for depth in (1, 2, 3, 4, 5, 6):
    r = subprocess.run([sys.executable, "-c", BODY, str(depth)], capture_output=True, text=True, timeout=300)
    tail = (r.stdout.strip().splitlines() or [""])[-1]
    err = [l for l in r.stderr.splitlines() if "rror" in l or "Abort" in l or "HIP" in l or "ssert" in l][:4]
    print("depth", depth, "rc", r.returncode, "|", tail, "|", err)
