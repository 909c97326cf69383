import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_hip_peer_virtual_ranks as T
orig_init = T.VirtualRanks.__init__
orig_view = T.VirtualRanks.view
def init(self, ws, cap, captured=False):
    orig_init(self, ws, cap, captured)
    s = torch.cuda.Stream(); self.streams = [s] * ws
T.VirtualRanks.__init__ = init
class W:
    def __init__(self, t): self.t = t
    def copy_(self, src): torch.add(src, 0, out=self.t); return self.t
T.VirtualRanks.view = lambda self, r, off, rows, n, dtype: W(orig_view(self, r, off, rows, n, dtype))
ws, dtype, n = 4, torch.float32, 256
for trial in range(6):
    v = T.VirtualRanks(ws, 4 << 20)
    m = 3
    parts = [T._pattern(r, m, n, dtype, salt=trial) for r in range(ws)]
    want = torch.stack(parts).sum(0)
    outs = T.all_reduce(v, parts, 1)
    torch.cuda.synchronize()
    for r in range(ws):
        if not torch.equal(outs[r], want):
            d = outs[r] - want
            rows = (d != 0).any(1).nonzero().flatten().tolist()
            # which partial sums explain the result?
            expl = []
            for row in rows:
                for mask in range(1 << ws):
                    s_ = sum(parts[p][row] for p in range(ws) if mask >> p & 1) if mask else torch.zeros(n, device="cuda")
                    if torch.equal(outs[r][row], s_): expl.append((row, bin(mask))); break
                else: expl.append((row, "unexplained", outs[r][row][:4].tolist(), want[row][:4].tolist()))
            print("trial", trial, "rank", r, "bad rows", rows, "= sum of partials", expl)
    # what do the buffers hold now?
    for p in range(ws):
        held = orig_view(v, p, v.cap, m, n, dtype)      # epoch 1 -> half 1
        print("   buffer", p, "holds own partial:", torch.equal(held, parts[p]), "reduced rows:", [torch.equal(held[i], want[i]) for i in range(m)])
    print("errors", v.errors())
    v.close()
