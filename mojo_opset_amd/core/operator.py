"""`MojoOperator`: the drop-in boundary.

Restates the dispatch of `mojo_opset/core/operator.py`:

* a class that lists `MojoOperator` directly in its bases is a *core op* (`Mojo<Name>`);
  defining it attaches a fresh `MojoBackendRegistry` (reference :22-36);
* any deeper subclass is a backend implementation and registers itself as a side effect of
  the ``class`` statement (:36);
* instantiating the core op reads ``MOJO_BACKEND`` **at every construction** and builds the
  selected backend class instead (:38-51);
* `forward_diff_with` is the parity helper used by every accuracy test (:81-129).

Deliberate difference: the reference's core classes carry the torch-native golden `forward`
and auto-generate a ``Torch<Name>`` subclass (:34).  Here the core classes are API-only
(constructor, contracts, abstract `forward`) and ``Torch<Name>`` lives in the repo-level
``oracle/`` package, which is test infrastructure: the product never computes on the CPU,
and constructing an op with no usable backend fails loudly.
"""
import os
from abc import ABC, abstractmethod
from typing import Optional

import torch

from .acc import check_tol_diff

_FACTORY_KEYS = ("device", "dtype", "layout", "requires_grad", "pin_memory", "memory_format")


def get_tensor_factory_kwargs(**kwargs):
    """`mojo_opset/utils/misc.py:27-32`: keep only the `torch.empty` keywords that are set."""
    return {k: v for k, v in kwargs.items() if v is not None and k in _FACTORY_KEYS}


class MojoOperator(ABC, torch.nn.Module):
    supported_platforms_list = ["rocm", "cpu"]
    _backend = None

    def __init_subclass__(cls, **kwargs):
        kwargs.pop("default_priority", None)
        super().__init_subclass__(**kwargs)
        if MojoOperator in cls.__bases__:
            from .backend_registry import MojoBackendRegistry

            cls._registry = MojoBackendRegistry(cls)
        else:
            cls._registry.register(cls)

    def __new__(cls, *args, **kwargs):
        if MojoOperator not in cls.__bases__:
            return super().__new__(cls)
        registry = getattr(cls, "_registry", None)
        if registry is None or not registry.registered_backends():
            raise NotImplementedError(
                f"No {cls.__name__} implementation found, please register at least one "
                f"(the HIP backend registers on a ROCm host; the torch golden backend is "
                f"provided by the repo-level `oracle` package for tests only)."
            )
        target = registry.get(os.environ.get("MOJO_BACKEND"))
        return target.__new__(target, *args, **kwargs)

    # -- introspection (reference :53-70) -------------------------------------------------
    @classmethod
    def get_registry(cls):
        registry = getattr(cls, "_registry", None)
        if registry is None:
            raise NotImplementedError(f"No {cls.__name__} implementation found, please register at least one.")
        return registry

    @classmethod
    def get_backend_impl(cls, backend_name: Optional[str] = None, *, strict: bool = False):
        return cls.get_registry().get(backend_name, strict=strict)

    @classmethod
    def get_registered_backends(cls):
        return cls.get_registry().registered_backends()

    def __init__(self, **kwargs):
        torch.nn.Module.__init__(self)
        self.tensor_factory_kwargs = get_tensor_factory_kwargs(**kwargs)

    @abstractmethod
    def forward(self, *args, **kwargs):
        raise NotImplementedError

    # -- parity helper (reference :81-129) -------------------------------------------------
    def forward_diff_with(
        self,
        other_op: "MojoOperator",
        *args,
        atol=1e-2,
        rtol=1e-2,
        ptol=1.0,
        random_seed: int = 42,
        mixed_tol=False,
        ref_device: Optional[str] = None,
        **kwargs,
    ):
        """Run ``self`` and ``other_op`` on clones of the same inputs and compare.

        ``ref_device`` (an addition to the reference signature): when set, tensor arguments
        are moved to that device for ``other_op`` only, so a CPU oracle can check a GPU op.
        """
        if type(self) is type(other_op):
            raise NotImplementedError(
                f"No dedicated backend for {type(self).__name__}; "
                f"both operands resolve to the same implementation, skipping comparison."
            )

        def _fresh(v):
            return v.clone() if isinstance(v, torch.Tensor) else v

        os.environ["PYTHONHASHSEED"] = str(random_seed)
        torch.manual_seed(random_seed)
        mine = self.forward(*[_fresh(a) for a in args], **{k: _fresh(v) for k, v in kwargs.items()})
        torch.manual_seed(random_seed)

        def _ref(v):
            v = _fresh(v)
            return v.to(ref_device) if (ref_device is not None and isinstance(v, torch.Tensor)) else v

        theirs = other_op.forward(*[_ref(a) for a in args], **{k: _ref(v) for k, v in kwargs.items()})

        assert mine is not None, "forward should return a non-None value."
        assert theirs is not None, "comparison operator should return a non-None value."

        def _host(t):
            if isinstance(t, (tuple, list)):
                return type(t)(_host(x) for x in t)
            return t.detach().cpu() if isinstance(t, torch.Tensor) else t

        check_tol_diff(_host(mine), _host(theirs), atol, rtol, ptol, mixed_tol)
        return mine

    def extra_repr(self) -> str:
        return ""
