from .attention import *  # noqa: F401,F403
from .streaming import *  # noqa: F401,F403
