"""HIP<Op> classes of the HBM-streaming operators: SwiGLU, (ResidualAdd)RMSNorm, RoPE,
RotaryEmbedding, StorePagedKVCache.  Each `forward` validates the reference's contract, hands raw
device pointers to the C ABI and returns torch tensors; there is no torch compute in here."""
from typing import Optional

import torch

from ....core.operators.activation import MojoSwiGLU
from ....core.operators.kv_cache import (MojoStorePagedKVCache, MojoStorePagedMLAKVCache,
                                           assert_paged_kv_layout_contract)
from ....core.operators.normalization import MojoResidualAddRMSNorm, MojoRMSNorm, MojoRMSNormInplace
from ....core.operators.position_embedding import MojoApplyRoPE, MojoRotaryEmbedding
from .. import lib as L

_ROCM = ["rocm"]


def _dense(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


def _row_view(t: torch.Tensor):
    """(rows, row stride in elements) when ``t`` is a stack of rows with one uniform row stride and a dense last dimension
    (a column slice of a dense [.., N] tensor); None otherwise."""
    if t.dim() == 0 or t.stride(-1) != 1 or t.numel() == 0:
        return None
    cols = t.shape[-1]
    if t.dim() == 1:
        return 1, cols
    ld = t.stride(-2)
    rows, expect = 1, ld
    for size, stride in zip(reversed(t.shape[:-1]), reversed(t.stride()[:-1])):
        if size != 1 and stride != expect:
            return None
        expect *= size
        rows *= size
    return rows, ld


class HIPSwiGLU(MojoSwiGLU):
    supported_platforms_list = _ROCM

    def forward(self, gate_out: torch.Tensor, up_out: torch.Tensor) -> torch.Tensor:
        L.require_cuda(gate_out, up_out)
        if gate_out.shape != up_out.shape or gate_out.dtype != up_out.dtype:
            raise NotImplementedError("HIPSwiGLU: gate_out and up_out must have identical shape and dtype")
        if gate_out.is_contiguous() and up_out.is_contiguous():
            g, u = gate_out, up_out
            out = torch.empty_like(g)
            L.check(L.load().mojo_hip_swiglu(L.ptr(g), L.ptr(u), L.ptr(out), g.numel(), L.dtype_code(g.dtype),
                                             float(self.swiglu_limit), L.stream_of(g)), "HIPSwiGLU")
            return out
        # Row-strided views — the two halves of a fused [.., 2 * inter] projection (`gate, up = gu.chunk(2, -1)`) — are read in
        # place: no copy of either half.  Anything else is made dense first.
        rows_g, rows_u = _row_view(gate_out), _row_view(up_out)
        if rows_g is None or rows_u is None or (rows_g[1] * gate_out.element_size()) % 16 or (rows_u[1] * up_out.element_size()) % 16 \
                or gate_out.data_ptr() % 16 or up_out.data_ptr() % 16:
            return self.forward(_dense(gate_out), _dense(up_out))
        cols = gate_out.shape[-1]
        out = torch.empty(gate_out.shape, dtype=gate_out.dtype, device=gate_out.device)
        L.check(L.load().mojo_hip_swiglu_rows(L.ptr(gate_out), L.ptr(up_out), L.ptr(out), rows_g[0], cols, rows_g[1], rows_u[1], cols,
                                              L.dtype_code(gate_out.dtype), float(self.swiglu_limit), L.stream_of(gate_out)), "HIPSwiGLU")
        return out


def _rmsnorm(hidden, residual, weight, eps, want_sum, out=None):
    L.require_cuda(hidden, residual, weight)
    if weight.dtype != hidden.dtype or (residual is not None and residual.dtype != hidden.dtype):
        raise NotImplementedError("hip rmsnorm: hidden, residual and weight must share one dtype")
    if residual is not None and residual.shape != hidden.shape:
        raise NotImplementedError("hip rmsnorm: hidden and residual must have identical shapes")
    dim = hidden.shape[-1]
    assert weight.shape == (dim,), f"weight shape {tuple(weight.shape)} does not match hidden size {dim}"
    h = _dense(hidden)
    r = None if residual is None else _dense(residual)
    normed = torch.empty_like(h) if out is None else out
    summed = torch.empty_like(h) if want_sum else None
    rows = h.numel() // dim if dim else 0
    L.check(L.load().mojo_hip_residual_add_rmsnorm(L.ptr(h), L.ptr(r), L.ptr(_dense(weight.detach())), L.ptr(normed),
                                                   L.ptr(summed), rows, dim, L.dtype_code(h.dtype), float(eps),
                                                   L.stream_of(h)), "hip rmsnorm")
    return normed, summed


class HIPRMSNorm(MojoRMSNorm):
    supported_platforms_list = _ROCM

    def forward(self, hidden_state: torch.Tensor) -> torch.Tensor:
        return _rmsnorm(hidden_state, None, self.weight, self.variance_epsilon, False)[0]


class HIPRMSNormInplace(MojoRMSNormInplace):
    """``inplace=True``: the kernel's output pointer IS the input (each thread reads its elements of a row before the
    row's reduction and writes the same elements after it, so no element is read after it was overwritten)."""

    supported_platforms_list = _ROCM

    def forward(self, hidden_state: torch.Tensor) -> torch.Tensor:
        if not self.inplace:
            return _rmsnorm(hidden_state, None, self.weight, self.variance_epsilon, False)[0]
        if hidden_state.is_contiguous():
            _rmsnorm(hidden_state, None, self.weight, self.variance_epsilon, False, out=hidden_state)
        else:                                # a strided view (e.g. the q slice of a fused qkv): normalise a dense copy, write back
            hidden_state.copy_(_rmsnorm(hidden_state, None, self.weight, self.variance_epsilon, False)[0])
        return hidden_state


class HIPResidualAddRMSNorm(MojoResidualAddRMSNorm):
    supported_platforms_list = _ROCM

    def forward(self, hidden_state: torch.Tensor, residual: torch.Tensor):
        pre = self.norm_pos == "pre"
        normed, summed = _rmsnorm(hidden_state, residual, self.weight, self.variance_epsilon, pre)
        return (normed, summed) if pre else (normed, normed)


class HIPApplyRoPE(MojoApplyRoPE):
    supported_platforms_list = _ROCM

    @staticmethod
    def _btn_strides(t: torch.Tensor, head_first: bool):
        """(batch, token, head) strides of a 3-D/4-D q or k in the caller's layout."""
        s = t.stride()
        if t.ndim == 3:
            return (0, s[1], s[0]) if head_first else (0, s[0], s[1])
        return (s[0], s[2], s[1]) if head_first else (s[0], s[1], s[2])

    def forward(self, q, k, cos, sin, head_first: bool = True):
        self.check_shape_contract(q, k, cos, sin)
        L.require_cuda(q, k, cos, sin)
        if q.dtype != k.dtype:
            raise NotImplementedError("HIPApplyRoPE: q and k must share one dtype")
        if q.stride(-1) != 1:
            q = q.contiguous()
        if k.stride(-1) != 1:
            k = k.contiguous()
        cos = _dense(cos.float())
        sin = _dense(sin.float())
        head_axis, tok_axis = (-3, -2) if head_first else (-2, -3)
        batch = q.shape[0] if q.ndim == 4 else 1
        tokens = q.shape[tok_axis]
        assert k.shape[tok_axis] == tokens and k.shape[-1] == q.shape[-1] and (k.ndim == 3 or k.shape[0] == batch)
        rope_dim = cos.shape[-1]
        assert cos.shape[-2] == tokens, "cos/sin rows must match the token dimension of q/k"
        if cos.ndim == 3:
            assert q.ndim == 4 and cos.shape[0] == batch
            cos_b, cos_t = cos.stride(0), cos.stride(1)
        else:
            cos_b, cos_t = 0, cos.stride(0)
        q_out = torch.empty(q.shape, dtype=q.dtype, device=q.device)
        k_out = torch.empty(k.shape, dtype=k.dtype, device=k.device)
        L.check(L.load().mojo_hip_apply_rope(
            L.ptr(q), L.ptr(k), L.ptr(q_out), L.ptr(k_out), L.ptr(cos), L.ptr(sin),
            batch, tokens, q.shape[head_axis], k.shape[head_axis], q.shape[-1], rope_dim,
            L.strides3(*self._btn_strides(q, head_first)), L.strides3(*self._btn_strides(k, head_first)),
            L.strides3(*self._btn_strides(q_out, head_first)), L.strides3(*self._btn_strides(k_out, head_first)),
            cos_b, cos_t, L.dtype_code(q.dtype), L.stream_of(q)), "HIPApplyRoPE")
        return q_out, k_out


class HIPRotaryEmbedding(MojoRotaryEmbedding):
    supported_platforms_list = _ROCM

    def forward(self, x, cu_q_lens=None, total_seq_lens=None, position_ids=None):
        self.check_index_contract(x, cu_q_lens, total_seq_lens, position_ids)
        L.require_cuda(x, cu_q_lens, total_seq_lens, position_ids)
        dev = x.device
        d = 2 * self.inv_freq.shape[0]            # rope_dim (the reference's class keeps only inv_freq)
        batch = 0
        if cu_q_lens is not None:
            mode, lead = 2, (x.shape[0],)
            batch = cu_q_lens.shape[0] - 1
            cu_q_lens = _dense(cu_q_lens)
            total_seq_lens = None if total_seq_lens is None else _dense(total_seq_lens)
        elif position_ids is not None:
            mode, lead = 0, tuple(position_ids.shape)
            position_ids = _dense(position_ids)
        else:
            mode, lead = 1, (x.shape[1],)
        n_pos = 1
        for s in lead:
            n_pos *= s
        cos = torch.empty(*lead, d, dtype=torch.float32, device=dev)
        sin = torch.empty_like(cos)
        if mode == 2 and batch == 0:
            mode, position_ids = 0, torch.full(lead, -1, dtype=torch.int32, device=dev)
        cached = self.init_max_length is not None
        inv_freq = _dense(self.inv_freq.to(dev))
        L.check(L.load().mojo_hip_rotary_embedding(
            L.ptr(cos), L.ptr(sin), n_pos, d, mode, L.ptr(position_ids), L.ptr(cu_q_lens), L.ptr(total_seq_lens),
            batch, L.ptr(self.cos.to(dev)) if cached else None, L.ptr(self.sin.to(dev)) if cached else None,
            self.init_max_length if cached else 0, L.ptr(inv_freq), float(self.attention_scaling),
            L.stream_of(cos)), "HIPRotaryEmbedding")
        return cos, sin


class HIPStorePagedKVCache(MojoStorePagedKVCache):
    supported_platforms_list = _ROCM

    def forward(self, key_states, value_states, key_cache, value_cache, block_table: Optional[torch.Tensor] = None,
                cu_q_lens: Optional[torch.Tensor] = None, context_kv_lens: Optional[torch.Tensor] = None, *,
                chunk_metadata: Optional[torch.Tensor] = None):
        self.check_call_contract(key_states, value_states, block_table, cu_q_lens, context_kv_lens, chunk_metadata)
        L.require_cuda(key_states, value_states, key_cache, value_cache, block_table, cu_q_lens, context_kv_lens,
                       chunk_metadata)
        assert key_cache.shape == value_cache.shape and key_cache.dim() == 4
        assert key_cache.dtype == key_states.dtype == value_states.dtype == value_cache.dtype
        n_blocks, heads, page, dim = key_cache.shape
        tokens = key_states.shape[0]
        assert key_states.shape[1:] == (heads, dim), "key/value states do not match the cache's (heads, head_dim)"
        if key_cache.stride(-1) != 1 or value_cache.stride(-1) != 1 or key_cache.stride() != value_cache.stride():
            raise NotImplementedError("HIPStorePagedKVCache: caches must share strides and be dense in head_dim")
        if key_states.stride(-1) != 1 or key_states.stride() != value_states.stride():
            key_states, value_states = key_states.contiguous(), value_states.contiguous()
        eb = key_cache.element_size()
        common = (tokens, heads, dim, n_blocks, page, eb, key_states.stride(0), key_states.stride(1),
                  key_cache.stride(0), key_cache.stride(1), key_cache.stride(2), L.stream_of(key_cache))
        lib = L.load()
        if chunk_metadata is not None:
            plan = _dense(chunk_metadata)
            L.check(lib.mojo_hip_store_paged_kv_plan(L.ptr(key_states), L.ptr(value_states), L.ptr(key_cache),
                                                     L.ptr(value_cache), L.ptr(plan), plan.shape[0], *common),
                    "HIPStorePagedKVCache")
        else:
            assert_paged_kv_layout_contract(block_table, cu_q_lens, context_kv_lens)
            batch = context_kv_lens.shape[0]
            if cu_q_lens is not None:
                assert cu_q_lens.shape[0] == batch + 1
            table = block_table if block_table.stride(1) == 1 else block_table.contiguous()
            L.check(lib.mojo_hip_store_paged_kv_layout(
                L.ptr(key_states), L.ptr(value_states), L.ptr(key_cache), L.ptr(value_cache), L.ptr(table),
                table.stride(0), table.shape[1], L.ptr(None if cu_q_lens is None else _dense(cu_q_lens)),
                L.ptr(_dense(context_kv_lens)), batch, *common), "HIPStorePagedKVCache")
        return key_cache, value_cache


class HIPStorePagedMLAKVCache(MojoStorePagedMLAKVCache):
    supported_platforms_list = _ROCM

    def forward(self, compressed_kv_states, k_pe_states, compressed_kv_cache, k_pe_cache, block_table, cu_q_lens,
                context_kv_lens):
        assert_paged_kv_layout_contract(block_table, cu_q_lens, context_kv_lens)
        L.require_cuda(compressed_kv_states, k_pe_states, compressed_kv_cache, k_pe_cache, block_table, cu_q_lens,
                       context_kv_lens)
        if context_kv_lens is None:
            return compressed_kv_cache, k_pe_cache
        assert compressed_kv_cache.dim() == 4 and k_pe_cache.dim() == 4 and compressed_kv_cache.shape[1] == 1
        assert compressed_kv_cache.shape[:3] == k_pe_cache.shape[:3]
        n_blocks, _, page, r = compressed_kv_cache.shape
        rope = k_pe_cache.shape[3]
        assert compressed_kv_states.dim() == 2 and compressed_kv_states.shape[1] == r
        assert k_pe_states.dim() == 2 and k_pe_states.shape[1] == rope
        assert compressed_kv_states.shape[0] == k_pe_states.shape[0]
        dt = compressed_kv_cache.dtype
        if not (dt == k_pe_cache.dtype == compressed_kv_states.dtype == k_pe_states.dtype):
            raise NotImplementedError("HIPStorePagedMLAKVCache: states and caches must share one dtype")
        if compressed_kv_cache.stride(3) != 1 or k_pe_cache.stride(3) != 1:
            raise NotImplementedError("HIPStorePagedMLAKVCache: caches must be dense in their last dimension")
        ckv = compressed_kv_states if compressed_kv_states.stride(1) == 1 else compressed_kv_states.contiguous()
        kpe = k_pe_states if k_pe_states.stride(1) == 1 else k_pe_states.contiguous()
        batch = context_kv_lens.shape[0]
        if cu_q_lens is not None:
            assert cu_q_lens.shape[0] == batch + 1
        table = block_table if block_table.stride(1) == 1 else block_table.contiguous()
        L.check(L.load().mojo_hip_store_paged_mla_kv(
            L.ptr(ckv), L.ptr(kpe), L.ptr(compressed_kv_cache), L.ptr(k_pe_cache), L.ptr(table), table.stride(0),
            table.shape[1], L.ptr(None if cu_q_lens is None else _dense(cu_q_lens)), L.ptr(_dense(context_kv_lens)), batch,
            ckv.shape[0], r, rope, n_blocks, page, dt.itemsize, ckv.stride(0), kpe.stride(0),
            compressed_kv_cache.stride(0), compressed_kv_cache.stride(2), k_pe_cache.stride(0), k_pe_cache.stride(2),
            L.stream_of(compressed_kv_cache)), "HIPStorePagedMLAKVCache")
        return compressed_kv_cache, k_pe_cache
