"""Helpers for the `-m gpu` parity tests: run a HIP<Op> through the C ABI on cuda:0 and compare with
the oracle (CPU) or with the committed reference vectors."""
import torch

import mojo_opset_amd as mo
import oracle  # noqa: F401
from conftest import build_op, clone_tree, to_device

DEV = "cuda"


def skip_unless_experiments_build():
    """Kernels under csrc/experiments/ (measured slower, DESIGN Appendix A) are compiled only with
    MOJO_HIP_BUILD_EXPERIMENTS=1; their tests run against such a build and skip otherwise."""
    import pytest

    from mojo_opset_amd.backends.hip import lib

    if not lib.built_with_experiments():
        pytest.skip("library built without MOJO_HIP_BUILD_EXPERIMENTS=1")


def last_launch() -> str:
    """The kernel form the last HIP operator call of this thread launched (`mojo_hip_last_launch`): what an A/B test asserts
    on to prove that its two legs took two different forms."""
    from mojo_opset_amd.backends.hip import lib

    return lib.last_launch()


def launches_of(fn) -> str:
    """Run ``fn`` and return the '|'-separated kernel forms it launched (`mojo_hip_launch_history`)."""
    from mojo_opset_amd.backends.hip import lib

    lib.launch_history(clear=True)
    fn()
    return lib.launch_history()


class switch_env:
    """``with switch_env(MOJO_HIP_X="1", MOJO_HIP_Y=None): ...`` — set (None: unset) switches and make both layers re-read
    them (they are latched at first use); restores and reloads on exit.  For loops inside one test, where the `monkeypatch`
    fixture of conftest.py (which reloads too) is clumsy."""

    def __init__(self, **values):
        self.values, self.old = values, {}

    def __enter__(self):
        import os

        from mojo_opset_amd import switches

        for k, v in self.values.items():
            self.old[k] = os.environ.get(k)
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)
        switches.reload()
        return self

    def __exit__(self, *exc):
        import os

        from mojo_opset_amd import switches

        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        switches.reload()
        return False


def hip_cls(op_name):
    return getattr(mo, op_name).get_backend_impl("hip", strict=True)


def torch_cls(op_name):
    return getattr(mo, op_name).get_backend_impl("torch", strict=True)


def run_hip_case(case):
    op = build_op(hip_cls(case["op"]), case, device=DEV)
    args = to_device(clone_tree(case["args"]), DEV)
    kwargs = to_device(clone_tree(case["kwargs"]), DEV)
    out = op.forward(*args, **kwargs)
    torch.cuda.synchronize()
    return out


def to_cpu(x):
    if isinstance(x, (tuple, list)):
        return type(x)(to_cpu(v) for v in x)
    return x.detach().cpu()


def assert_close_tree(got, want, atol, rtol):
    if isinstance(want, (tuple, list)):
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert_close_tree(g, w, atol, rtol)
        return
    assert got.dtype == want.dtype and got.shape == want.shape, (got.dtype, want.dtype, got.shape, want.shape)
    torch.testing.assert_close(got.float(), want.float(), atol=atol, rtol=rtol)


def max_ulp_bf16ish(got, want, atol=0.0):
    """Largest |got - want| in units of the storage type's last place at |want| (16-bit float types).
    Differences not exceeding ``atol`` count as zero, so cancellation near zero is not mis-read as a
    many-ulp error."""
    if got.numel() == 0:
        return 0
    mant = {torch.bfloat16: 7, torch.float16: 10}[want.dtype]
    g, w = got.double(), want.double()
    diff = (g - w).abs()
    expo = torch.floor(torch.log2(w.abs().clamp_min(2.0 ** -24)))
    ulp = torch.pow(2.0, expo - mant)
    diff = torch.where(diff <= atol, torch.zeros_like(diff), diff)
    return float((diff / ulp).max())
