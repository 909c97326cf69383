"""ctypes binding of libmojo_hip.so (the C ABI declared in include/mojo_hip.h).

The library is located in-tree (``mojo_opset_amd/lib/libmojo_hip.so``) or through
``MOJO_HIP_LIB``.  There is no fallback of any kind: if it is missing, or a call returns a
non-zero status, the caller gets an exception.
"""
import ctypes
import os
import threading
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p

import torch

_PKG = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DEFAULT_LIB = os.path.join(_PKG, "lib", "libmojo_hip.so")

MOJO_F32, MOJO_F16, MOJO_BF16, MOJO_I8, MOJO_F8E4M3 = 0, 1, 2, 3, 4
_DTYPE_CODE = {
    torch.float32: MOJO_F32, torch.float16: MOJO_F16, torch.bfloat16: MOJO_BF16,
    torch.int8: MOJO_I8, torch.float8_e4m3fn: MOJO_F8E4M3,
}

MOJO_EINVAL, MOJO_EUNSUPPORTED, MOJO_ELAUNCH, MOJO_EWORKSPACE = -1, -2, -3, -4

_P = c_void_p
_I = c_int64
_I64x3 = c_int64 * 3

# name -> (restype, argtypes); must mirror include/mojo_hip.h exactly (tests/test_c_abi.py checks
# that every declared symbol is exported and listed here).
SIGNATURES = {
    "mojo_hip_version": (c_char_p, []),
    "mojo_hip_last_error": (c_char_p, []),
    "mojo_hip_reload_env": (None, []),
    "mojo_hip_switches": (c_int64, [c_char_p, _I]),
    "mojo_hip_last_launch": (c_char_p, []),
    "mojo_hip_launch_history": (c_char_p, [c_int]),
    "mojo_hip_store_paged_kv_plan": (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "mojo_hip_store_paged_kv_layout": (c_int, [_P, _P, _P, _P, _P, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I,
                                               _I, _I, _I, _P]),
    "mojo_hip_swiglu": (c_int, [_P, _P, _P, _I, c_int, c_float, _P]),
    "mojo_hip_swiglu_rows": (c_int, [_P, _P, _P, _I, _I, _I, _I, _I, c_int, c_float, _P]),
    "mojo_hip_store_paged_mla_kv": (c_int, [_P, _P, _P, _P, _P, _I, _I, _P, _P] + [_I] * 13 + [_P]),
    "mojo_hip_dynamic_quant": (c_int, [_P, _P, _P, _P, _I, _I, c_int, _P]),
    "mojo_hip_residual_add_rmsnorm_quant": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, c_int, c_int, c_float, c_float, _P]),
    "mojo_hip_moe_gating_workspace_bytes": (c_int64, [_I, _I, _I, c_int]),
    "mojo_hip_moe_gating": (c_int, [_P, _P, _P, _P, _I, _I, _I, _I, c_int, _P, _I, _P]),
    "mojo_hip_moe_dispatch_workspace_bytes": (c_int64, [_I, _I]),
    "mojo_hip_moe_dispatch": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, c_int, _P, _I, _P]),
    "mojo_hip_moe_combine_workspace_bytes": (c_int64, [_I, _I]),
    "mojo_hip_moe_combine": (c_int, [_P, _P, _P, _P, _I, _I, _I, c_int, _P, _I, _P]),
    "mojo_hip_residual_add_rmsnorm": (c_int, [_P, _P, _P, _P, _P, _I, _I, c_int, c_float, _P]),
    "mojo_hip_apply_rope": (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I64x3, _I64x3, _I64x3, _I64x3,
                                    _I, _I, c_int, _P]),
    "mojo_hip_rotary_embedding": (c_int, [_P, _P, _I, _I, c_int, _P, _P, _P, _I, _P, _P, _I, _P, c_float, _P]),
    "mojo_hip_page_pool_extend": (c_int, [_P, _I, _I, _P, _P, _I, _P, _P, _P, _I, _I, _I, _P]),
    "mojo_hip_page_pool_advance": (c_int, [_P, _P, _I, _P, _I, _P]),
    "mojo_hip_paged_decode_gqa_workspace_bytes": (c_int64, [_I, _I, _I, _I, _I, _I, _I]),
    "mojo_hip_paged_decode_gqa": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I,
                                          c_float, c_int, c_int, c_int, _P]),
    "mojo_hip_paged_prefill_gqa_workspace_bytes": (c_int64, [_I, _I, _I, _I, _I, _I, _I, _I, _I]),
    "mojo_hip_paged_prefill_gqa": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I,
                                           c_float, c_int, c_int, _P, _I, _P]),
    "mojo_hip_group_gemm_workspace_bytes": (c_int64, [_I]),
    "mojo_hip_group_gemm": (c_int, [_P, _P, _P, _P, c_int, _I, _I, _I, _I, c_int, c_int, _P, _I, _P]),
    "mojo_hip_group_gemm_swiglu": (c_int, [_P, _P, _P, _P, c_int, _I, _I, _I, _I, c_int, c_int, _P, _I, _P]),
    "mojo_hip_group_gemm_strided": (c_int, [_P, _P, _P, _P, c_int, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, c_int, _P, _I, _P]),
    "mojo_hip_mla_latent_attn_workspace_bytes": (c_int64, [_I, _I, _I, _I]),
    "mojo_hip_mla_latent_attn": (c_int, [_P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I,
                                         _I, _I, _I, _I, _I, _I, c_float, c_int, _P]),
    "mojo_hip_gemm_workspace_bytes": (c_int64, [_I, _I, _I]),
    "mojo_hip_gemm": (c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, c_int, _P, _I, _P]),
    "mojo_hip_gemm_rowmap": (c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, c_int, _P, _I, _P]),
    "mojo_hip_gemm_swiglu_workspace_bytes": (c_int64, [_I, _I, _I]),
    "mojo_hip_gemm_swiglu": (c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, c_int, _P, _I, _P]),
    "mojo_hip_gemm_residual_rmsnorm_workspace_bytes": (c_int64, [_I, _I, _I]),
    "mojo_hip_gemm_residual_rmsnorm": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, c_int, c_float, _P, _I, _P]),
    "mojo_hip_qkv_rope_store_workspace_bytes": (c_int64, [_I, _I, _I]),
    "mojo_hip_qkv_rope_store": (c_int, [_P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _P] + [_I] * 12 + [c_int, _P, _I, _P]),
    "mojo_hip_mla_prefill_supported": (c_int, [_I, _I, _I, c_int]),
    "mojo_hip_mla_unpage": (c_int, [_P, _P, _P, _P, _P, _P, _P] + [_I] * 13 + [_P, _P]),
    "mojo_hip_mla_prefill_attn": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, c_float, c_int, c_int, c_int, _P]),
    "mojo_hip_peer_ctrl_bytes": (c_int64, []),
    "mojo_hip_peer_max_ranks": (c_int64, []),
    "mojo_hip_peer_max_chunks": (c_int64, []),
    "mojo_hip_peer_handle_bytes": (c_int64, []),
    "mojo_hip_peer_set_timeout_ms": (c_int64, [_I]),
    "mojo_hip_peer_alloc": (c_int, [_P, _I, c_int]),
    "mojo_hip_peer_free": (c_int, [_P]),
    "mojo_hip_peer_export": (c_int, [_P, _P]),
    "mojo_hip_peer_open": (c_int, [_P, _P]),
    "mojo_hip_peer_close": (c_int, [_P]),
    "mojo_hip_peer_error": (c_int, [_P, c_int, _P]),
    "mojo_hip_peer_peek": (c_int, [_P, _P, _I]),
    "mojo_hip_peer_begin": (c_int, [_P, _P, _I, _I, _P]),
    "mojo_hip_peer_signal": (c_int, [_P, _P, _I, _I, c_int, _I, ctypes.c_uint32, _P]),
    "mojo_hip_peer_reduce": (c_int, [_P, _P, _I, _I, _I, ctypes.c_uint32, _I, _I, _I, _P, _I, c_int, c_int, _P]),
    "mojo_hip_peer_gather": (c_int, [_P, _P, _I, _I, _I, ctypes.c_uint32, _I, _I, _I, _P, _I, c_int, _P]),
    "mojo_hip_peer_pull": (c_int, [_P, _P, _I, _I, c_int, _I, ctypes.c_uint32, _I, _I, _P, _I, c_int, _P]),
    "mojo_hip_quant_gemm_workspace_bytes": (c_int64, [_I, _I, _I]),
    "mojo_hip_quant_gemm": (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, c_int, c_int, c_int, _P, _I, _P]),
}

_lock = threading.Lock()
_lib = None


class MojoHipError(RuntimeError):
    pass


def lib_path() -> str:
    return os.environ.get("MOJO_HIP_LIB", DEFAULT_LIB)


def load():
    """Load (once) and return the ctypes handle; raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = lib_path()
        if not os.path.exists(path):
            raise MojoHipError(
                f"libmojo_hip.so not found at {path}. Build it with `python -m mojo_opset_amd.csrc.build` "
                f"(or __graft_entry__.build()). The hip backend has no CPU fallback."
            )
        handle = ctypes.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise MojoHipError(f"{path} does not export {name}; rebuild the library") from e
            fn.restype = res
            fn.argtypes = args
        _check_source_hash(handle, path)
        _lib = handle
    return _lib


def reload_env() -> None:
    """Make the library re-read its MOJO_HIP_* switches (they are latched at first use: include/mojo_hip.h)."""
    load().mojo_hip_reload_env()


def last_launch() -> str:
    """The kernel form this thread's last operator call launched (debug / A-B tests)."""
    return load().mojo_hip_last_launch().decode()


def launch_history(clear: bool = False) -> str:
    """'|'-separated kernel forms of this thread's launches since the last clear (debug / A-B tests)."""
    return load().mojo_hip_launch_history(1 if clear else 0).decode()


def switches() -> dict:
    """{name: value or None} of every switch the library has read so far."""
    buf = ctypes.create_string_buffer(4096)
    load().mojo_hip_switches(buf, len(buf))
    out = {}
    for item in buf.value.decode().split():
        name, _, val = item.partition("=")
        out[name] = int(val) if val else None
    return out


def built_with_experiments() -> bool:
    """True when the loaded library was built with MOJO_HIP_BUILD_EXPERIMENTS=1 (csrc/experiments/ kernels present)."""
    return "+experiments" in load().mojo_hip_version().decode()


def _check_source_hash(handle, path):
    """The prebuilt library travels with the tree (gpurun snapshot): refuse one built from OTHER sources than the
    ``csrc/`` + ``include/`` beside it.  ``mojo_hip_version()`` ends in ``src=<hash>`` (csrc/build.py::source_hash).
    Skipped when the sources are not there (a library deployed on its own) or with MOJO_HIP_ALLOW_STALE=1."""
    if os.environ.get("MOJO_HIP_ALLOW_STALE", "") == "1":
        return
    if os.path.abspath(path) != os.path.abspath(DEFAULT_LIB):
        return
    try:
        from mojo_opset_amd.csrc.build import hashed_files, source_hash

        if not hashed_files():
            return
        want = source_hash()
    except (ImportError, OSError):
        return
    version = handle.mojo_hip_version().decode()
    have = version.rsplit("src=", 1)[-1] if "src=" in version else "unstamped"
    if have != want:
        raise MojoHipError(
            f"{path} was built from other sources than this tree (library src={have}, tree src={want}). Rebuild it with "
            f"`python -m mojo_opset_amd.csrc.build` (or set MOJO_HIP_ALLOW_STALE=1 to load it anyway).")


c_void_p = c_void_p  # re-export for callers that build offset pointers


def dtype_code(dtype: torch.dtype) -> int:
    try:
        return _DTYPE_CODE[dtype]
    except KeyError:
        raise NotImplementedError(f"hip backend: dtype {dtype} is not supported") from None


def ptr(t):
    """Device pointer of a tensor (or NULL for None)."""
    return None if t is None else c_void_p(t.data_ptr())


def stream_of(t: torch.Tensor):
    return c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def strides3(*vals):
    return _I64x3(*vals)


def ints4(*vals):
    return (c_int64 * 4)(*vals)


def check(status: int, what: str):
    """Map a C status to the exception type the reference's operators raise."""
    if status == 0:
        return
    msg = load().mojo_hip_last_error().decode("utf-8", "replace")
    text = f"{what}: {msg}"
    if status == MOJO_EINVAL:
        raise ValueError(text)
    if status == MOJO_EUNSUPPORTED:
        raise NotImplementedError(text)
    raise MojoHipError(f"{text} (status {status})")


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise MojoHipError(
                "hip backend called with a CPU tensor; it has no CPU path (select MOJO_BACKEND=torch from the "
                "oracle package for host-side reference runs)."
            )
