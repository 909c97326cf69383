"""The ``hip`` backend: `HIP<Op>` classes (one per `Mojo<Op>`), registered by definition.

On a host without a ROCm GPU the classes still import (so their signatures can be inspected) but
the registry ignores them; on a ROCm host they are the first-priority backend.
"""
from .operators.attention import *  # noqa: F401,F403
from .operators.streaming import *  # noqa: F401,F403
from .operators.gemm import HIPGemm, HIPGroupGemm, HIPQuantGemm, HIPSwiGLUMLP  # noqa: F401
from .operators.mla import HIPPagedDecodeMLA, HIPPagedPrefillMLA  # noqa: F401
from .operators.compute_with_comm import (HIPAllGatherGemm, HIPGemmAll2All, HIPGemmAllReduce,  # noqa: F401
                                          HIPGemmReduceScatter)
from .operators.moe import HIPExperts, HIPMoE, HIPMoECombine, HIPMoEDispatch, HIPMoEGating  # noqa: F401
from .operators.quantize import HIPDynamicQuant, HIPResidualAddRMSNormQuant  # noqa: F401
