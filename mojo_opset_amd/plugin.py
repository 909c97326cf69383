"""Out-of-tree registration of the hip backend into the *reference* package (XPU-Forces/mojo_opset).

The reference loads entry points of group ``mojo_opset.plugins`` at import time and calls them
(`mojo_opset/__init__.py:19-45`).  ``register`` is that callable; ``rebase_hip_backend`` does the work: for every
operator of the hot path it defines ``HIP<Op>(reference.Mojo<Op>)`` whose ``forward`` (and private helpers) are the
ones of this repository's ``HIP<Op>``, so the class statement registers it under backend name ``"hip"``
(`core/operator.py:22-36`, `core/backend_registry.py:49-54`).

The unmodified reference knows no ROCm platform (`utils/platform.py:16-41` falls through to ``"meta_device"``) and no
``"hip"`` backend name (`core/backend_registry.py:13-18`), so ``register`` first calls ``enable_rocm_platform``: on a
host where a ROCm GPU is visible it adds, in memory, the platform ``"rocm"`` (torch device ``cuda``, dist backend
``nccl``, backend priority ``["hip", "torch"]``) and points the reference's ``get_platform`` at it.  A host without a
ROCm GPU, or one that the reference already maps to one of its own platforms, is left exactly as it was.
"""
import functools
import importlib
import inspect
import sys
from typing import Dict, Optional, Sequence

ROCM_PLATFORM = "rocm"
ROCM_BACKEND_PRIORITY = ("hip", "torch")

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
        already = ref_core.get_registry()._registry.get("hip")
        if already is not None:                                  # the entry point ran before: keep what it made
            made[name] = already
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


def rocm_gpu_present() -> bool:
    """True on a PyTorch-ROCm build that sees at least one GPU (the probe ``register`` gates on)."""
    import torch

    return getattr(torch.version, "hip", None) is not None and torch.cuda.is_available()


def _all_subclasses(cls):
    seen, todo = set(), [cls]
    while todo:
        for sub in todo.pop().__subclasses__():
            if sub not in seen:
                seen.add(sub)
                todo.append(sub)
    return seen


def enable_rocm_platform(reference) -> bool:
    """Teach the imported, unmodified reference package the platform ``"rocm"``; returns whether that platform is
    active afterwards.  Restates in memory what INTEGRATION.md §1 shows as a source patch:

    * ``utils/platform.py:16-41``  ``get_platform()`` → ``"rocm"`` (every module that imported the name is re-pointed);
    * ``utils/platform.py:44-75``  torch device ``"cuda"``, dist backend ``"nccl"`` (= RCCL);
    * ``core/backend_registry.py:13-21``  priority ``["hip", "torch"]``, ``BACKEND_PRIORITY_LIST`` updated in place;
    * ``core/operator.py:19`` / ``core/function.py:13``  every ``supported_platforms_list`` that names the generic
      ``"meta_device"`` fallback also names ``"rocm"``, so classes defined later still register their torch backend.

    Only acts when the reference resolved this host to ``"meta_device"`` (no accelerator of its own) and
    ``rocm_gpu_present()``; otherwise nothing is touched."""
    root = reference.__name__
    plat_mod = importlib.import_module(root + ".utils.platform")
    current = plat_mod.get_platform()
    if current == ROCM_PLATFORM:
        return True
    if current != "meta_device" or not rocm_gpu_present():
        return False
    registry_mod = importlib.import_module(root + ".core.backend_registry")
    operator_mod = importlib.import_module(root + ".core.operator")
    function_mod = importlib.import_module(root + ".core.function")

    original = plat_mod.get_platform

    @functools.lru_cache
    def get_platform() -> str:
        return ROCM_PLATFORM

    get_platform.__doc__ = original.__doc__
    get_platform.reference_get_platform = original
    for name, mod in list(sys.modules.items()):
        if mod is None or not (name == root or name.startswith(root + ".")):
            continue
        if vars(mod).get("get_platform") is original:
            setattr(mod, "get_platform", get_platform)
        if vars(mod).get("platform") == "meta_device":          # module-level snapshots (backends/__init__.py:8)
            setattr(mod, "platform", ROCM_PLATFORM)
    plat_mod._PLATFORM_TO_TORCH_DEVICE[ROCM_PLATFORM] = "cuda"
    plat_mod._PLATFORM_TO_DIST_BACKEND[ROCM_PLATFORM] = "nccl"
    plat_mod.get_torch_device.cache_clear()
    plat_mod.get_dist_backend.cache_clear()

    registry_mod.PLATFORM_BACKEND_PRIORITY[ROCM_PLATFORM] = list(ROCM_BACKEND_PRIORITY)
    # the list object is shared by reference (registration validation and sort()): change it in place
    shared = registry_mod.BACKEND_PRIORITY_LIST
    for plat, prio in list(registry_mod.PLATFORM_BACKEND_PRIORITY.items()):
        if prio is shared:                                       # keep the dict entry of the old platform intact
            registry_mod.PLATFORM_BACKEND_PRIORITY[plat] = list(prio)
    shared[:] = list(ROCM_BACKEND_PRIORITY)

    for base in (operator_mod.MojoOperator, function_mod.MojoFunction):
        for klass in {base} | _all_subclasses(base):
            own = vars(klass).get("supported_platforms_list")
            if isinstance(own, list) and "meta_device" in own and ROCM_PLATFORM not in own:
                own.append(ROCM_PLATFORM)
    return True


def register() -> None:  # entry point: mojo_opset.plugins
    """Entry point of group ``mojo_opset.plugins`` (`mojo_opset/__init__.py:19-45`).  Works against the unmodified
    reference: on a ROCm host it adds the platform and registers every ``HIP<Op>``; anywhere else it does nothing
    (the reference's loader logs and skips a plugin that raises, so a CPU-only host must not raise either)."""
    import mojo_opset

    if not enable_rocm_platform(mojo_opset):
        return
    rebase_hip_backend(mojo_opset)
