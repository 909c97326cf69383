"""Run-time switches of the hip backend (``MOJO_HIP_*`` environment variables) — ONE semantics in both layers.

The reference reads its one switch where it uses it (``MOJO_BACKEND`` in ``MojoOperator.__new__``,
`mojo_opset/core/operator.py:45-47`; this package does the same for ``MOJO_BACKEND``).  The ``MOJO_HIP_*`` switches select kernel
forms and exchange algorithms on paths that run thousands of times per second, so a value is taken from the environment the
FIRST time it is needed and then LATCHED — by this module for the switches the Python layer reads, by ``libmojo_hip.so`` for
the ones its launchers read (`include/mojo_hip.h`, "Run-time switches").  ``reload()`` drops every latched value of both
layers; the next use reads the environment again.  Call it after changing a variable, with no operator call in flight.

The table of switches is INTEGRATION.md section 5; `tests/test_c_abi.py` checks it against the names in the binary and in
this package.
"""
import os
from typing import Dict, Optional

_LATCHED: Dict[str, Optional[str]] = {}


def get(name: str, default: Optional[str] = None) -> Optional[str]:
    """The latched value of ``name`` (``default`` when unset or empty)."""
    try:
        v = _LATCHED[name]
    except KeyError:
        v = os.environ.get(name) or None
        _LATCHED[name] = v
    return default if v is None else v


def get_int(name: str, default: int) -> int:
    v = get(name)
    try:
        return default if v is None else int(v)
    except ValueError:
        return default


def reload() -> None:
    """Forget every latched value: this module's and, when the library is loaded, its own (``mojo_hip_reload_env``)."""
    _LATCHED.clear()
    from .backends.hip import lib

    if lib._lib is not None:
        lib._lib.mojo_hip_reload_env()


def in_effect() -> Dict[str, object]:
    """Every switch read so far with the value in effect (None = unset): the Python layer's and the library's."""
    out: Dict[str, object] = dict(_LATCHED)
    from .backends.hip import lib

    if lib._lib is not None:
        out.update(lib.switches())
    return out
