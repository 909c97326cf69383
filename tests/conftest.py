import functools
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def usable_cores():
    """Cores this process may actually use (affinity mask cut by the cgroup CPU quota).  torch sizes its pool from the cores
    it SEES; on the GPU box (a 16-core share of a 128-core host) an all-cores OpenMP team spinning on the quota turned the
    oracle's seconds into minutes, so the suite pins the pool to what is granted."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return max(1, n)


torch.set_num_threads(min(torch.get_num_threads(), usable_cores()))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class _SwitchAwareMonkeyPatch:
    """pytest's monkeypatch with one addition: after setenv / delenv of a ``MOJO_HIP_*`` variable the latched switches of the
    Python layer and of libmojo_hip.so are re-read (mojo_opset_amd.switches.reload), and once more when the test ends and the
    environment is restored.  The switches are latched at first use (include/mojo_hip.h "Run-time switches"): without the
    reload a test that flips a switch AFTER a first call in the same process silently re-runs the default path (the four
    vacuous A/B tests of round 4)."""

    def __init__(self, mp):
        self._mp = mp
        self.touched = False

    def __getattr__(self, name):
        return getattr(self._mp, name)

    def _reload(self, name):
        if str(name).startswith("MOJO_HIP_"):
            self.touched = True
            from mojo_opset_amd import switches

            switches.reload()

    def setenv(self, name, value, prepend=None):
        self._mp.setenv(name, value, prepend)
        self._reload(name)

    def delenv(self, name, raising=True):
        self._mp.delenv(name, raising)
        self._reload(name)


@pytest.fixture
def monkeypatch(monkeypatch):
    wrapped = _SwitchAwareMonkeyPatch(monkeypatch)
    yield wrapped
    if wrapped.touched:
        monkeypatch.undo()                       # restore the environment first, then latch it again
        from mojo_opset_amd import switches

        switches.reload()


def load_golden(name):
    return torch.load(os.path.join(GOLDEN, f"{name}.pt"), weights_only=False)["cases"]


def to_device(x, device):
    if isinstance(x, torch.Tensor):
        return x.to(device)
    if isinstance(x, (list, tuple)):
        return type(x)(to_device(v, device) for v in x)
    if isinstance(x, dict):
        return {k: to_device(v, device) for k, v in x.items()}
    return x


def clone_tree(x):
    if isinstance(x, torch.Tensor):
        return x.clone()
    if isinstance(x, (list, tuple)):
        return type(x)(clone_tree(v) for v in x)
    if isinstance(x, dict):
        return {k: clone_tree(v) for k, v in x.items()}
    return x


def build_op(cls, case, device="cpu"):
    """Instantiate ``cls`` the way the fixture generator did and load its recorded state."""
    ctor = case["ctor"]
    kwargs = dict(ctor.get("kwargs", {}))
    args = to_device(clone_tree(ctor.get("args", ())), device)
    op = cls(*args, **kwargs)
    if case.get("cast") is not None:
        op = op.to(case["cast"])
    op = op.to(device)
    with torch.no_grad():
        for k, v in case["state"].items():
            slot = functools.reduce(getattr, k.split("."), op)
            if k in case.get("keep_dtype", ()):      # e.g. the fp32 router weight inside a bf16 layer
                slot.data = v.to(device)
            else:
                slot.copy_(v.to(device))
    return op


def bit_equal(a, b):
    if isinstance(a, (tuple, list)):
        return len(a) == len(b) and all(bit_equal(x, y) for x, y in zip(a, b))
    a, b = a.detach().cpu(), b.detach().cpu()
    if a.dtype != b.dtype or a.shape != b.shape:
        return False
    if a.is_floating_point():
        return torch.equal(torch.nan_to_num(a.float(), nan=12345.0), torch.nan_to_num(b.float(), nan=12345.0))
    return torch.equal(a, b)
