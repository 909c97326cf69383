"""Where does prefill_w64_kernel differ from prefill_kernel?  One sequence, no cache; error map by (64-position block, head, 32-d block)."""
import os, sys, torch
sys.path.insert(0, ".")
from benchmarks.extras import _paged, hip
dev = torch.device("cuda:0")
hq, hkv, d, page = 32, 8, 128, 16
n = int(os.environ.get("N", "1024"))
torch.manual_seed(0)
op = hip("MojoPagedPrefillGQA")()
k, v, table = _paged(dev, [n], hkv, d, page)
q = torch.randn(n, hq, d, device=dev, dtype=torch.bfloat16)
cu = torch.tensor([0, n], dtype=torch.int32, device=dev)
outs = {}
for m in ("0", "1"):
    os.environ["MOJO_HIP_PREFILL_W64"] = m
    outs[m] = op(q, k, v, cu, table, cu_total_seq_lens=cu, max_q_len=n, max_total_seq_len=n).float()
torch.cuda.synchronize()
err = (outs["1"] - outs["0"]).abs()
err = torch.nan_to_num(err, nan=1e9, posinf=1e9)
print("max err", err.max().item(), "bad frac", (err > 2e-2).float().mean().item())
blk = err.view(n // 64, 64, hq, 4, 32).amax(dim=(1, 4))          # [pos block, head, d block]
print("per position block (max over heads, d):", [f"{x:.2g}" for x in blk.amax(dim=(1, 2)).tolist()])
print("per d block:", [f"{x:.2g}" for x in blk.amax(dim=(0, 1)).tolist()])
print("per head:", [f"{x:.2g}" for x in blk.amax(dim=(0, 2)).tolist()])
rows = err.view(n // 64, 64, hq, d).amax(dim=(0, 2, 3))
print("per position inside its block (max):", [f"{x:.2g}" for x in rows.tolist()])
if os.environ.get("SHOW"):
    for t in (0, 1, 63, 64, n - 1):
        print("row", t, "head 0 base", outs["0"][t, 0, :6].tolist(), "w64", outs["1"][t, 0, :6].tolist())
if os.environ.get("NANMAP"):
    bad = torch.isnan(outs["1"]).any(-1) | torch.isinf(outs["1"]).any(-1) | ((outs["1"] - outs["0"]).abs().amax(-1) > 0.05)   # [n, hq]
    idx = bad.nonzero()
    print("bad (row, head) count", idx.shape[0], "rows", sorted(set(idx[:, 0].tolist()))[:80], "heads", sorted(set(idx[:, 1].tolist())))
    for r, hh in idx[:6].tolist():
        print("  row", r, "head", hh, "w64", outs["1"][r, hh, :4].tolist(), "base", outs["0"][r, hh, :4].tolist())
