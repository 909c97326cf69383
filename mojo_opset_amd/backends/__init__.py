from . import hip  # noqa: F401
