"""Out-of-tree registration of the hip backend into the *reference* package (XPU-Forces/mojo_opset).

The reference loads entry points of group ``mojo_opset.plugins`` at import time and calls them
(`mojo_opset/__init__.py:19-45`).  ``register`` is that callable; ``rebase_hip_backend`` does the work: for every
operator of the hot path it defines ``HIP<Op>(reference.Mojo<Op>)`` whose ``forward`` (and private helpers) are the
ones of this repository's ``HIP<Op>``, so the class statement registers it under backend name ``"hip"``
(`core/operator.py:22-36`, `core/backend_registry.py:49-54`).  See INTEGRATION.md for the two-line platform /
priority patch the reference needs so that a ROCm host resolves to platform ``"rocm"``.
"""
import importlib
import inspect
from typing import Dict, Optional, Sequence

_SKIP = {"__module__", "__qualname__", "__doc__", "__dict__", "__weakref__", "__init__", "__abstractmethods__",
         "_abc_impl", "supported_platforms_list", "_registry", "_backend", "__parameters__", "__orig_bases__"}


def _helpers_of(cls, stop_at) -> Dict[str, object]:
    """Functions / static methods defined by ``cls`` and its bases up to (excluding) ``stop_at``: the forward
    implementation plus the contract helpers it calls on ``self``."""
    out: Dict[str, object] = {}
    for klass in reversed(cls.__mro__):
        if klass is object or klass in stop_at:
            continue
        for name, value in vars(klass).items():
            if name in _SKIP or name.startswith("_init"):
                continue
            if inspect.isfunction(value) or isinstance(value, (staticmethod, classmethod)):
                out[name] = value
    return out


def rebase_hip_backend(reference, platforms: Optional[Sequence[str]] = None) -> Dict[str, type]:
    """Define and register ``HIP<Op>`` subclasses of ``reference.Mojo<Op>`` (also looked up in
    ``reference.experimental``).  Returns ``{op name: new class}``."""
    import torch

    import mojo_opset_amd as mine
    from mojo_opset_amd.backends import hip as hip_pkg
    from mojo_opset_amd.core.operator import MojoOperator as MyOperator

    platforms = list(platforms) if platforms is not None else ["rocm"]
    stop = {MyOperator, torch.nn.Module} | set(torch.nn.Module.__mro__) | set(MyOperator.__mro__)
    made: Dict[str, type] = {}
    for name in mine.__all__:
        if not name.startswith("Mojo") or name in ("MojoOperator", "MojoBackendRegistry"):
            continue
        hip_cls = getattr(hip_pkg, "HIP" + name[4:], None)
        ref_core = getattr(reference, name, None)
        if ref_core is None:
            try:
                exp = importlib.import_module(reference.__name__ + ".experimental")
            except ImportError:
                exp = None
            ref_core = getattr(exp, name, None) if exp is not None else None
        if hip_cls is None or ref_core is None:
            continue
        body = _helpers_of(hip_cls, stop)
        # the reference's own constructor / attributes win; only behaviour that is new comes from this repo
        for existing in list(body):
            if existing != "forward" and existing in vars(ref_core):
                del body[existing]
        body["supported_platforms_list"] = platforms
        body["__module__"] = __name__
        body["__doc__"] = f"hip backend of {name} (libmojo_hip.so), re-based on the reference's core class."
        made[name] = type("HIP" + name[4:], (ref_core,), body)
    return made


def register() -> None:  # entry point: mojo_opset.plugins
    import mojo_opset

    rebase_hip_backend(mojo_opset)
