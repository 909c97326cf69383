"""Flakiness probe for tests/test_hip_peer_virtual_ranks.py: variants of the harness (one stream / producer = kernel instead of
copy_ / plain hipMalloc buffers), failures out of N runs each."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_hip_peer_virtual_ranks as T
from mojo_opset_amd.backends.hip import lib as L
modes = sys.argv[1].split("+")
orig_init = T.VirtualRanks.__init__
orig_view = T.VirtualRanks.view
if "plainmalloc" in modes:
    lib = L.load()
    real_alloc = lib.mojo_hip_peer_alloc
    class FakeLib:
        def __getattr__(self, n): return getattr(lib, n)
        def mojo_hip_peer_alloc(self, p, total, unc): return real_alloc(p, total, 0)
    fake = FakeLib()
    T.L = type("LL", (), {"load": staticmethod(lambda: fake), "check": staticmethod(L.check), "ptr": staticmethod(L.ptr), "dtype_code": staticmethod(L.dtype_code)})
def init(self, ws, cap, captured=False):
    orig_init(self, ws, cap, captured)
    if "onestream" in modes:
        s = torch.cuda.Stream()
        self.streams = [s] * ws
T.VirtualRanks.__init__ = init
if "kernelcopy" in modes:
    class W:
        def __init__(self, t): self.t = t
        def copy_(self, src): torch.add(src, 0, out=self.t); return self.t
    T.VirtualRanks.view = lambda self, r, off, rows, n, dtype: W(orig_view(self, r, off, rows, n, dtype))
fails, total = {}, 0
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
    for ws in (4, 8):
        for cap in (False, True):
            for dt in (torch.bfloat16, torch.float32):
                total += 1
                try:
                    T.test_direct_exchange_kernels_with_virtual_ranks(ws, cap, dt)
                except AssertionError as e:
                    k = (ws, cap, str(e).split(":")[0][:40])
                    fails[k] = fails.get(k, 0) + 1
print(sys.argv[1], "failures", sum(fails.values()), "of", total, fails)
