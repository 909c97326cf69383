"""Boundary behaviour pinned by the reference's `tests/base/test_backend_dispatch.py:16-74`
(default instance type == first priority; `Torch*` prefix; strict-missing raises; name
normalisation) plus the rules of `core/backend_registry.py:48-118`."""

import pytest
import torch

import mojo_opset_amd as mo
import oracle  # noqa: F401
from mojo_opset_amd.core import MojoOperator
from mojo_opset_amd.core.backend_registry import BACKEND_PRIORITY_LIST
from mojo_opset_amd.core.platform import get_dist_backend, get_platform, get_torch_device

ALL_OPS = [n for n in mo.__all__ if n.startswith("Mojo") and n not in ("MojoOperator", "MojoBackendRegistry")]


def test_platform_mapping():
    if torch.cuda.is_available():
        assert (get_platform(), get_torch_device(), get_dist_backend()) == ("rocm", "cuda", "nccl")
        assert BACKEND_PRIORITY_LIST == ["hip", "torch"]
    else:
        assert (get_platform(), get_torch_device(), get_dist_backend()) == ("cpu", "cpu", "gloo")
        assert BACKEND_PRIORITY_LIST == ["torch"]


@pytest.mark.parametrize("name", ALL_OPS)
def test_every_op_has_a_torch_backend_and_a_hip_class(name):
    core = getattr(mo, name)
    torch_cls = core.get_backend_impl("torch", strict=True)
    assert torch_cls.__name__ == "Torch" + name[4:]
    assert core.get_backend_impl(" Torch ") is torch_cls
    from mojo_opset_amd.backends import hip

    hip_cls = getattr(hip, "HIP" + name[4:])
    assert issubclass(hip_cls, core)
    if get_platform() == "rocm":
        assert core.get_registered_backends()[0] == "hip"
        assert core.get_backend_impl("hip", strict=True) is hip_cls
    else:
        assert core.get_registered_backends() == ("torch",)


def test_default_instance_is_first_priority(monkeypatch):
    monkeypatch.delenv("MOJO_BACKEND", raising=False)
    op = mo.MojoSwiGLU()
    first = mo.MojoSwiGLU.get_registered_backends()[0]
    assert type(op) is mo.MojoSwiGLU.get_backend_impl(first)


def test_env_selects_backend_at_each_construction(monkeypatch):
    monkeypatch.setenv("MOJO_BACKEND", "torch")
    assert type(mo.MojoSwiGLU()).__name__ == "TorchSwiGLU"
    monkeypatch.setenv("MOJO_BACKEND", "no_such_backend")   # silent fallback to first priority
    first = mo.MojoSwiGLU.get_registered_backends()[0]
    assert type(mo.MojoSwiGLU()) is mo.MojoSwiGLU.get_backend_impl(first)


def test_strict_missing_backend_raises():
    with pytest.raises(KeyError):
        mo.MojoSwiGLU.get_backend_impl("ttx", strict=True)


def test_bad_prefix_is_rejected():
    with pytest.raises(AssertionError):
        class BogusSwiGLU(mo.MojoSwiGLU):  # noqa: F841
            supported_platforms_list = ["cpu", "rocm"]

            def forward(self, a, b):
                return a
    with pytest.raises(NameError):
        class TorchyishSwiGLU(mo.MojoSwiGLU):  # noqa: F841
            supported_platforms_list = ["cpu", "rocm"]

            def forward(self, a, b):
                return a


def test_core_without_backend_fails_loudly():
    class MojoNothingRegistered(MojoOperator):
        pass

    with pytest.raises(NotImplementedError):
        MojoNothingRegistered()


def test_forward_diff_with_same_class_raises_not_implemented():
    a = mo.MojoSwiGLU.get_backend_impl("torch")()
    b = mo.MojoSwiGLU.get_backend_impl("torch")()
    with pytest.raises(NotImplementedError):
        a.forward_diff_with(b, torch.ones(2, 2), torch.ones(2, 2))


def test_ctor_error_conventions():
    with pytest.raises(ValueError):
        mo.MojoPagedDecodeGQA(gqa_layout="BABA")
    with pytest.raises(ValueError):
        mo.MojoPagedPrefillGQA(gqa_layout="x")
    with pytest.raises(ValueError):
        mo.MojoResidualAddRMSNorm(8, norm_pos="mid")
    with pytest.raises(TypeError):
        mo.MojoGemmAllReduce(torch.zeros(2, 2), trans_weight=1)
    with pytest.raises(AssertionError):
        mo.MojoApplyRoPE(interleaved=True)
    with pytest.raises(AssertionError):
        mo.MojoQuantGemm(8, 8, quant_dtype=torch.int32)
