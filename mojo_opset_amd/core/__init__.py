from .acc import check_tol_diff
from .backend_registry import MojoBackendRegistry
from .operator import MojoOperator
from .operators import *  # noqa: F401,F403
from .operators import __all__ as _ops_all
from .platform import get_dist_backend, get_platform, get_torch_device

__all__ = ["MojoOperator", "MojoBackendRegistry", "check_tol_diff", "get_platform", "get_torch_device",
           "get_dist_backend", *_ops_all]
