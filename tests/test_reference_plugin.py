"""Drop-in check against the REAL reference (only where /root/reference exists, i.e. the authoring container;
never on the GPU box): `mojo_opset_amd.plugin.rebase_hip_backend` must register a `HIP<Op>` subclass of every
reference core op on the path, selectable with MOJO_BACKEND=hip, leaving the reference's torch backend intact.
CPU only: registration, dispatch and signatures — no compute."""
import os
import subprocess
import sys

import pytest

REF = os.environ.get("MOJO_REFERENCE_ROOT", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "mojo_opset")), reason="reference checkout not present")

SCRIPT = r'''
import inspect, os, sys
sys.path.insert(0, os.environ["MOJO_REFERENCE_ROOT"]); sys.path.insert(0, os.environ["REPO_ROOT"])
os.environ["MOJO_OPSET_PLUGIN_AUTOLOAD"] = "0"
import mojo_opset as ref
import mojo_opset.experimental
from mojo_opset.core import backend_registry as br
from mojo_opset.utils.platform import get_platform
plat = get_platform()                                   # "meta_device" on this host
# the maintainer-side patch of INTEGRATION.md §1, applied in memory: make "hip" a legal first-priority backend
br.BACKEND_PRIORITY_LIST.insert(0, "hip")
from mojo_opset_amd.plugin import rebase_hip_backend
made = rebase_hip_backend(ref, platforms=[plat])
import mojo_opset_amd as mine
expected = [n for n in mine.__all__ if n.startswith("Mojo") and n not in ("MojoOperator", "MojoBackendRegistry")]
assert sorted(made) == sorted(expected), (sorted(made), sorted(expected))
for name, cls in made.items():
    core = getattr(ref, name, None) or getattr(ref.experimental, name)
    assert issubclass(cls, core) and cls.__name__ == "HIP" + name[4:]
    assert core._registry.get("hip", strict=True) is cls
    assert core.get_registered_backends()[0] == "hip"
    torch_cls = core._registry.get("torch", strict=True)
    assert torch_cls.__name__ == "Torch" + name[4:] and torch_cls.forward is core.forward
    # same positional parameters as the golden forward (backends may only add optional ones)
    gold = [p for p in inspect.signature(core.forward).parameters.values()]
    ours = {p.name: p for p in inspect.signature(cls.forward).parameters.values()}
    for p in gold:
        assert p.name in ours, (name, p.name)
        assert ours[p.name].kind == p.kind, (name, p.name)
os.environ["MOJO_BACKEND"] = "hip"
op = ref.MojoPagedDecodeGQA(gqa_layout="ABAB")
assert type(op).__name__ == "HIPPagedDecodeGQA" and op.gqa_layout == "ABAB"
os.environ["MOJO_BACKEND"] = "torch"
assert type(ref.MojoPagedDecodeGQA()).__name__ == "TorchPagedDecodeGQA"
os.environ["MOJO_BACKEND"] = "hip"
q = ref.MojoQuantGemm(64, 32)
assert type(q).__name__ == "HIPQuantGemm" and set(q.state_dict()) == {"weight", "weight_scale"}
print("PLUGIN_OK", len(made))
'''


def test_hip_backend_registers_into_the_reference():
    env = dict(os.environ, MOJO_REFERENCE_ROOT=REF, REPO_ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env.pop("MOJO_BACKEND", None)
    res = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "PLUGIN_OK" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]
