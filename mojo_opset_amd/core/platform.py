"""Platform detection for the MI355X build.

The reference (`mojo_opset/utils/platform.py:16-41`) knows `npu | mlu | ilu | meta_device`
and maps a host without an accelerator to the torch device "meta" (:44-49), which makes
its own tests vacuous on CPU-only hosts.  This build knows exactly two platforms:

* ``"rocm"`` — `torch.cuda.is_available()` on a HIP build of torch; torch device
  ``"cuda"``; `torch.distributed` backend ``"nccl"`` (= RCCL on ROCm).
* ``"cpu"``  — everything else; torch device ``"cpu"``; dist backend ``"gloo"``.
"""
import functools

import torch


@functools.lru_cache(maxsize=None)
def get_platform() -> str:
    try:
        if torch.version.hip is not None and torch.cuda.is_available():
            return "rocm"
    except Exception:  # pragma: no cover - defensive, a broken driver must not kill import
        pass
    return "cpu"


def get_torch_device() -> str:
    return "cuda" if get_platform() == "rocm" else "cpu"


def get_dist_backend() -> str:
    return "nccl" if get_platform() == "rocm" else "gloo"
