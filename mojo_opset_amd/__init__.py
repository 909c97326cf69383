"""mojo_opset_amd — an MI355X-native ``hip`` backend behind the unchanged ``Mojo*`` operator API.

``import mojo_opset_amd`` exposes the `Mojo<Op>` classes of the hot path and
registers the ``HIP<Op>`` backend classes.  Backend selection follows the reference:
``MOJO_BACKEND`` is read at every construction; on a ROCm host the priority is
``["hip", "torch"]``.  The package contains no CPU compute path.
"""
from .core import *  # noqa: F401,F403
from .core import __all__ as _core_all
from . import backends  # noqa: F401  (registers HIP<Op> classes)
from .paged_cache import PagedDummyCache  # noqa: E402  device-side block allocator

__all__ = list(_core_all) + ["PagedDummyCache"]
__version__ = "0.1.0"
