"""HIP<Op> classes of the paged attention family."""
import math
from typing import Optional

import torch

from ....core.operators.attention import (MojoPagedDecodeGQA, MojoPagedPrefillGQA, assert_paged_decode_contract,
                                          assert_paged_prefill_contract)
from .... import switches
from .. import lib as L

_ROCM = ["rocm"]


def _validate_tables() -> bool:
    """Opt-in host check that reproduces the golden's ValueError for a row whose first page id is
    negative although its length is positive (`core/operators/attention.py:186-187`).  Off by default
    because it costs a device->host sync per call and cannot run under graph capture; without it such
    a row is computed over zero K/V (the kernel's treatment of every negative page id).  The same switch checks the
    ``max_total_seq_len`` / ``max_q_len`` hints against the device-side lengths (a length above its hint is truncated
    to the hint by the kernels)."""
    return switches.get("MOJO_HIP_VALIDATE", "0") == "1"


def _capturing(t: torch.Tensor) -> bool:
    return t.is_cuda and torch.cuda.is_current_stream_capturing()


def _check_cache_layout(key_cache, value_cache, what):
    if key_cache.stride() != value_cache.stride() or key_cache.stride(-1) != 1:
        raise NotImplementedError(f"{what}: key/value caches must share strides and be dense in head_dim")


class HIPPagedDecodeGQA(MojoPagedDecodeGQA):
    supported_platforms_list = _ROCM

    def forward(self, query, key_cache, value_cache, total_seq_lens, block_tables,
                softmax_scale: Optional[float] = None, mask: Optional[torch.Tensor] = None, *,
                max_total_seq_len: Optional[int] = None, leave_empty_rows: Optional[bool] = None):
        assert_paged_decode_contract(block_tables, total_seq_lens)
        if not self.is_causal or mask is not None:
            raise NotImplementedError("HIPPagedDecodeGQA supports causal attention without an explicit mask only")
        L.require_cuda(query, key_cache, value_cache, total_seq_lens, block_tables)
        batch, hq, dim = query.shape
        n_blocks, hkv, page, dim_c = key_cache.shape
        assert dim_c == dim and value_cache.shape == key_cache.shape and hq % hkv == 0
        assert query.dtype == key_cache.dtype == value_cache.dtype
        _check_cache_layout(key_cache, value_cache, "HIPPagedDecodeGQA")
        if _validate_tables() and batch > 0 and block_tables.shape[1] > 0:
            if bool(((total_seq_lens > 0) & (block_tables[:, 0] < 0)).any()):
                raise ValueError("Paged decode requires a valid block table for rows with kv lens > 0.")
            if max_total_seq_len is not None and int(total_seq_lens.max()) > int(max_total_seq_len):
                raise ValueError("HIPPagedDecodeGQA: a total_seq_lens entry exceeds max_total_seq_len")
        q = query if query.is_contiguous() else query.contiguous()
        tables = block_tables if block_tables.stride(1) == 1 else block_tables.contiguous()
        lens = total_seq_lens if total_seq_lens.is_contiguous() else total_seq_lens.contiguous()
        scale = 1.0 / math.sqrt(dim) if softmax_scale is None else float(softmax_scale)
        hint = int(max_total_seq_len) if max_total_seq_len is not None else 0
        out = torch.empty_like(q)
        lib = L.load()
        ws_bytes = lib.mojo_hip_paged_decode_gqa_workspace_bytes(batch, hq, hkv, dim, page, tables.shape[1], hint)
        ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=q.device)
        L.check(lib.mojo_hip_paged_decode_gqa(
            L.ptr(q), L.ptr(key_cache), L.ptr(value_cache), L.ptr(lens), L.ptr(tables), L.ptr(out), L.ptr(ws),
            ws.numel(), batch, hq, hkv, dim, page, tables.shape[1], tables.stride(0), key_cache.stride(0),
            key_cache.stride(1), key_cache.stride(2), hint, scale, 1 if self.gqa_layout == "ABAB" else 0,
            # replay contract of padded rows (seq_len <= 0): untouched while a graph is being captured, zeros eagerly
            1 if (_capturing(q) if leave_empty_rows is None else leave_empty_rows) else 0,
            L.dtype_code(q.dtype), L.stream_of(q)), "HIPPagedDecodeGQA")
        return out


class HIPPagedPrefillGQA(MojoPagedPrefillGQA):
    supported_platforms_list = _ROCM

    def forward(self, query, key_cache, value_cache, cu_q_lens, block_tables, softmax_scale: Optional[float] = None,
                cu_total_seq_lens: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None,
                max_q_len: Optional[int] = None, max_total_seq_len: Optional[int] = None):
        assert_paged_prefill_contract(cu_q_lens, block_tables, cu_total_seq_lens)
        if not self.is_causal or mask is not None:
            raise NotImplementedError("HIPPagedPrefillGQA supports causal attention without an explicit mask only")
        L.require_cuda(query, key_cache, value_cache, cu_q_lens, block_tables, cu_total_seq_lens)
        tokens, hq, dim = query.shape
        n_blocks, hkv, page, dim_c = key_cache.shape
        assert dim_c == dim and value_cache.shape == key_cache.shape and hq % hkv == 0
        assert query.dtype == key_cache.dtype == value_cache.dtype
        _check_cache_layout(key_cache, value_cache, "HIPPagedPrefillGQA")
        batch = cu_q_lens.shape[0] - 1
        if _validate_tables() and batch > 0 and block_tables.shape[1] > 0:
            q_lens = cu_q_lens[1:] - cu_q_lens[:-1]
            kv_lens = q_lens if cu_total_seq_lens is None else cu_total_seq_lens[1:] - cu_total_seq_lens[:-1]
            if bool(((q_lens > 0) & (kv_lens > 0) & (block_tables[:, 0] < 0)).any()):
                raise ValueError("Paged prefill requires a valid block table for rows with kv lens > 0.")
        q = query if query.is_contiguous() else query.contiguous()
        tables = block_tables if block_tables.stride(1) == 1 else block_tables.contiguous()
        cu_q = cu_q_lens.contiguous()
        cu_kv = None if cu_total_seq_lens is None else cu_total_seq_lens.contiguous()
        scale = 1.0 / math.sqrt(dim) if softmax_scale is None else float(softmax_scale)
        out = torch.empty_like(q)
        lib = L.load()
        hint_q = int(max_q_len) if max_q_len else 0
        hint_kv = int(max_total_seq_len) if max_total_seq_len else 0
        # few, long blocks (a chunked prefill against a long cache) are cut along the keys: fp32 partials + a merge launch
        ws_bytes = lib.mojo_hip_paged_prefill_gqa_workspace_bytes(tokens, batch, hq, hkv, dim, page, tables.shape[1], hint_q, hint_kv)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=q.device) if ws_bytes > 0 else None
        L.check(lib.mojo_hip_paged_prefill_gqa(
            L.ptr(q), L.ptr(key_cache), L.ptr(value_cache), L.ptr(cu_q), L.ptr(cu_kv), L.ptr(tables), L.ptr(out),
            tokens, batch, hq, hkv, dim, page, tables.shape[1], tables.stride(0), key_cache.stride(0),
            key_cache.stride(1), key_cache.stride(2), hint_q, hint_kv, scale,
            1 if self.gqa_layout == "ABAB" else 0, L.dtype_code(q.dtype), L.ptr(ws), ws_bytes, L.stream_of(q)), "HIPPagedPrefillGQA")
        return out
