"""Per-operator backend registry.

Behavioural restatement of `mojo_opset/core/backend_registry.py` for the two platforms this
build knows (see `platform.py`):

* backend name = lower-cased class-name prefix in front of the operator name
  (reference :49-54), so ``HIPPagedDecodeGQA`` registers as ``"hip"`` and the golden
  ``TorchPagedDecodeGQA`` as ``"torch"``;
* a prefix that is not in the platform's priority list is a programming error
  (`NameError` for a near miss, `AssertionError` otherwise — reference :65-75);
* a class whose ``supported_platforms_list`` lacks the current platform is skipped
  (reference :77,:90-91);
* ``get(name)`` normalises ``strip().lower()`` (:28-33) and silently falls back to the
  highest-priority registered class unless ``strict=True`` (:93-118).

New relative to the reference: platform ``"rocm"`` with priority ``["hip", "torch"]`` and
platform ``"cpu"`` with priority ``["torch"]``.
"""
from typing import Dict, Optional

from .platform import get_platform

PLATFORM_BACKEND_PRIORITY = {
    "rocm": ["hip", "torch"],
    "cpu": ["torch"],
}

# every prefix that is a legal backend name on *some* platform; used to tell a typo from
# a backend that simply does not apply to this host.
KNOWN_BACKENDS = ("hip", "torch")

BACKEND_PRIORITY_LIST = PLATFORM_BACKEND_PRIORITY[get_platform()]


def _normalize_backend_name(name: Optional[str]) -> Optional[str]:
    if name is None:
        return None
    return name.strip().lower()


class MojoBackendRegistry:
    def __init__(self, core_op_cls):
        assert core_op_cls.__name__.startswith("Mojo"), (
            f"Operator {core_op_cls.__name__} who is a subclass of MojoOperator, class name must start with Mojo."
        )
        self._core_op_cls = core_op_cls
        self._operator_name = core_op_cls.__name__[len("Mojo"):]
        self._registry: Dict[str, type] = {}

    def get_core_op_cls(self):
        return self._core_op_cls

    # -- registration -------------------------------------------------------------------
    def register(self, cls) -> None:
        cut = cls.__name__.find(self._operator_name)
        assert cut != -1, (
            f"Operator {cls.__name__} who be a subclass of {self._core_op_cls.__name__} must "
            f"contain {self._operator_name} in its name."
        )
        backend = _normalize_backend_name(cls.__name__[:cut])
        assert backend != "mojo", "should not register base backend"

        platform = get_platform()
        if backend not in BACKEND_PRIORITY_LIST:
            if backend in KNOWN_BACKENDS:
                # e.g. HIP* classes imported on a CPU-only host: legal, just not usable here.
                return
            for known in KNOWN_BACKENDS:
                if backend.startswith(known):
                    raise NameError(
                        f"Operator {cls.__name__} backend[{backend}] is not supported, "
                        f"are you wish to named {known.upper()}{self._operator_name} ?"
                    )
            raise AssertionError(
                f"Operator {cls.__name__} backend[{backend}] is not supported for platform[{platform}], "
                f"please choose from {BACKEND_PRIORITY_LIST}."
            )

        if platform not in getattr(cls, "supported_platforms_list", ()):
            return
        if backend in self._registry:
            raise ValueError(f"Operator {self._core_op_cls.__name__} backend[{backend}] has been registered")

        self._registry[backend] = cls
        cls._backend = backend
        self.sort()

    # -- lookup -------------------------------------------------------------------------
    def get(self, backend_name: Optional[str] = None, *, strict: bool = False):
        backend_name = _normalize_backend_name(backend_name)
        if backend_name is not None and backend_name in self._registry:
            return self._registry[backend_name]
        if strict and backend_name is not None:
            raise KeyError(
                f"{self._operator_name} backend {backend_name!r} is not registered; "
                f"available: {list(self._registry)}"
            )
        assert len(self._registry) > 0, f"{self._operator_name} does not implement any backend."
        return next(iter(self._registry.values()))

    def registered_backends(self):
        return tuple(self._registry)

    def sort(self) -> None:
        order = {name: i for i, name in enumerate(BACKEND_PRIORITY_LIST)}
        self._registry = dict(sorted(self._registry.items(), key=lambda kv: order.get(kv[0], len(order))))
