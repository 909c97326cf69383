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


# `register()` itself — the entry point the reference's loader calls — against the UNMODIFIED reference: no priority
# list patched by hand, no `platforms=` argument.  The only thing faked is the probe "a ROCm GPU is visible".
REGISTER_SCRIPT = r'''
import os, sys
sys.path.insert(0, os.environ["MOJO_REFERENCE_ROOT"]); sys.path.insert(0, os.environ["REPO_ROOT"])
os.environ["MOJO_OPSET_PLUGIN_AUTOLOAD"] = "0"
import mojo_opset as ref
import mojo_opset.experimental
from mojo_opset.core import backend_registry as br
from mojo_opset.utils import platform as plat
import mojo_opset_amd.plugin as plugin
assert plat.get_platform() == "meta_device" and br.BACKEND_PRIORITY_LIST == ["torch"]

# 1. a host without a ROCm GPU is left alone, and the entry point does not raise
plugin.rocm_gpu_present = lambda: False
plugin.register()
assert plat.get_platform() == "meta_device" and br.BACKEND_PRIORITY_LIST == ["torch"]
assert ref.MojoPagedDecodeGQA.get_registered_backends() == ("torch",)
assert "rocm" not in ref.MojoOperator.supported_platforms_list

# 2. a ROCm host: platform, device / dist maps, priority and every HIP<Op> appear, with no edit of the reference
plugin.rocm_gpu_present = lambda: True
plugin.register()
assert plat.get_platform() == "rocm" and br.get_platform() == "rocm"
assert plat.get_torch_device() == "cuda" and plat.get_dist_backend() == "nccl"
assert br.BACKEND_PRIORITY_LIST == ["hip", "torch"] and br.PLATFORM_BACKEND_PRIORITY["rocm"] == ["hip", "torch"]
assert br.PLATFORM_BACKEND_PRIORITY["meta_device"] == ["torch"]
import mojo_opset_amd as mine
expected = [n for n in mine.__all__ if n.startswith("Mojo") and n not in ("MojoOperator", "MojoBackendRegistry")]
for name in expected:
    core = getattr(ref, name, None) or getattr(ref.experimental, name)
    assert core.get_registered_backends() == ("hip", "torch"), (name, core.get_registered_backends())
    assert core.get_backend_impl("hip", strict=True).__name__ == "HIP" + name[4:]
    assert core.get_backend_impl("torch", strict=True).forward is core.forward
os.environ["MOJO_BACKEND"] = "hip"
op = ref.MojoPagedDecodeGQA(gqa_layout="ABAB")
assert type(op).__name__ == "HIPPagedDecodeGQA" and op.gqa_layout == "ABAB"
os.environ["MOJO_BACKEND"] = "torch"
assert type(ref.MojoPagedDecodeGQA()).__name__ == "TorchPagedDecodeGQA"
os.environ.pop("MOJO_BACKEND")
assert type(ref.MojoSwiGLU()).__name__ == "HIPSwiGLU"            # first priority when nothing is asked for
# an operator defined AFTER registration still gets its torch fallback on the new platform
class MojoLateOp(ref.MojoOperator):
    def forward(self, x):
        return x
assert MojoLateOp.get_registered_backends() == ("torch",)
# calling the entry point twice is harmless
first = ref.MojoSwiGLU.get_backend_impl("hip", strict=True)
plugin.register()
assert ref.MojoSwiGLU.get_backend_impl("hip", strict=True) is first
print("REGISTER_OK", len(expected))
'''


def test_register_entry_point_works_on_the_unmodified_reference():
    env = dict(os.environ, MOJO_REFERENCE_ROOT=REF, REPO_ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env.pop("MOJO_BACKEND", None)
    res = subprocess.run([sys.executable, "-c", REGISTER_SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "REGISTER_OK" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]
