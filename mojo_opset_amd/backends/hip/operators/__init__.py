from .attention import *  # noqa: F401,F403
from .gemm import HIPGemm, HIPGroupGemm, HIPQuantGemm, HIPSwiGLUMLP  # noqa: F401
from .streaming import *  # noqa: F401,F403
from .mla import *  # noqa: F401,F403
from .compute_with_comm import *  # noqa: F401,F403
from .moe import *  # noqa: F401,F403
from .quantize import *  # noqa: F401,F403
