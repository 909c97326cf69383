"""HIP<Op> classes of the paged MLA pair (weight-absorbed formulation, see csrc/mla_attn.hip)."""
import math
from typing import Optional

import torch

from ....core.operators.attention import assert_paged_decode_contract, assert_paged_prefill_contract
from ....core.operators.mla import MojoPagedDecodeMLA, MojoPagedPrefillMLA
from .. import lib as L

_ROCM = ["rocm"]


def _per_head_gemm(x_hm: torch.Tensor, proj: torch.Tensor, heads: int, rows: int, k: int, n: int, w_off: int,
                   w_group: int, w_k: int, w_n: int) -> torch.Tensor:
    """``out[h, i, :] = x_hm[h, i, :] @ W[h]`` for W[h] a strided sub-block of ``proj`` (one GEMM group per head)."""
    out = torch.empty(heads * rows, n, dtype=x_hm.dtype, device=x_hm.device)
    counts = torch.full((heads,), rows, dtype=torch.int32, device=x_hm.device)
    lib = L.load()
    ws = torch.empty(lib.mojo_hip_group_gemm_workspace_bytes(heads), dtype=torch.uint8, device=x_hm.device)
    w_ptr = L.c_void_p(proj.data_ptr() + w_off * proj.element_size())
    L.check(lib.mojo_hip_group_gemm_strided(L.ptr(x_hm), w_ptr, L.ptr(out), L.ptr(counts), 0, heads * rows, k, n, heads,
                                            k, n, w_group, w_k, w_n, L.dtype_code(x_hm.dtype), L.ptr(ws), ws.numel(),
                                            L.stream_of(x_hm)), "hip mla projection")
    return out.view(heads, rows, n)


def _mla_forward(op, query, ckv_cache, kpe_cache, block_tables, softmax_scale, *, total_seq_lens=None, cu_q_lens=None,
                 cu_total_seq_lens=None):
    L.require_cuda(query, ckv_cache, kpe_cache, block_tables, total_seq_lens, cu_q_lens, cu_total_seq_lens, op.kv_b_proj)
    tq, heads, qk = query.shape
    nope, rope, vdim, r = op.qk_nope_head_dim, op.qk_rope_head_dim, op.v_head_dim, op.kv_lora_rank
    assert heads == op.num_heads and qk == nope + rope
    proj = op.kv_b_proj.detach()
    if proj.dtype != query.dtype or ckv_cache.dtype != query.dtype or kpe_cache.dtype != query.dtype:
        # the golden's `c_kv @ kv_b_proj.T` raises the same way when the module was not cast to the cache dtype
        raise RuntimeError("expected m1 and m2 to have the same dtype: cast the module (kv_b_proj) to the cache dtype")
    assert ckv_cache.dim() == 4 and ckv_cache.shape[1] == 1 and ckv_cache.shape[3] == r
    assert kpe_cache.dim() == 4 and kpe_cache.shape[1] == 1 and kpe_cache.shape[3] == rope
    assert ckv_cache.shape[:3] == kpe_cache.shape[:3]
    if ckv_cache.stride(3) != 1 or kpe_cache.stride(3) != 1:
        raise NotImplementedError("hip mla: caches must be dense in their last dimension")
    proj = proj if proj.is_contiguous() else proj.contiguous()
    page = ckv_cache.shape[2]
    scale = 1.0 / math.sqrt(nope + rope) if softmax_scale is None else float(softmax_scale)
    dev = query.device
    if tq == 0:
        return torch.zeros(0, heads, vdim, dtype=query.dtype, device=dev)

    # 1) absorb W_kn into the query:  q_lat = [q_nope @ W_kn[h] | q_rope]
    q_hm = query[..., :nope].transpose(0, 1).contiguous()                          # [H, Tq, nope]
    q_abs = _per_head_gemm(q_hm, proj, heads, tq, nope, r, 0, (nope + vdim) * r, r, 1)   # W_kn[h]: [nope, r] (k rows)
    q_lat = torch.cat([q_abs.transpose(0, 1), query[..., nope:]], dim=-1).contiguous()   # [Tq, H, r + rope]

    # 2) attention over the compressed cache
    tables = block_tables if block_tables.stride(1) == 1 else block_tables.contiguous()
    o_lat = torch.empty(tq, heads, r, dtype=query.dtype, device=dev)
    lib = L.load()
    batch = tables.shape[0]
    max_len = page * tables.shape[1]
    ws = torch.empty(max(lib.mojo_hip_mla_latent_attn_workspace_bytes(tq, heads, r, max_len), 64), dtype=torch.uint8, device=dev)
    sink = getattr(op, "attn_sink", None)
    sink = None if sink is None else sink.detach().to(torch.float32).contiguous()
    L.check(lib.mojo_hip_mla_latent_attn(
        L.ptr(q_lat), L.ptr(ckv_cache), L.ptr(kpe_cache),
        L.ptr(None if total_seq_lens is None else total_seq_lens.contiguous()),
        L.ptr(None if cu_q_lens is None else cu_q_lens.contiguous()),
        L.ptr(None if cu_total_seq_lens is None else cu_total_seq_lens.contiguous()),
        L.ptr(tables), L.ptr(sink), L.ptr(o_lat), L.ptr(ws), ws.numel(), tq, batch, heads, r, rope, page, tables.shape[1],
        tables.stride(0), ckv_cache.stride(0), ckv_cache.stride(2), kpe_cache.stride(0), kpe_cache.stride(2), max_len,
        scale, L.dtype_code(query.dtype), L.stream_of(query)), "hip mla attention")

    # 3) out[h] = o_lat[h] @ W_v[h]^T      (W_v[h]: rows nope.. of head h's block, stored [v, r] = [N, K])
    o_hm = o_lat.transpose(0, 1).contiguous()                                      # [H, Tq, r]
    out = _per_head_gemm(o_hm, proj, heads, tq, r, vdim, nope * r, (nope + vdim) * r, 1, r)
    return out.transpose(0, 1).contiguous()


class HIPPagedDecodeMLA(MojoPagedDecodeMLA):
    supported_platforms_list = _ROCM

    def forward(self, query, compressed_kv_cache, k_pe_cache, total_seq_lens, block_tables,
                softmax_scale: Optional[float] = None):
        assert_paged_decode_contract(block_tables, total_seq_lens)
        return _mla_forward(self, query, compressed_kv_cache, k_pe_cache, block_tables, softmax_scale,
                            total_seq_lens=total_seq_lens)


class HIPPagedPrefillMLA(MojoPagedPrefillMLA):
    supported_platforms_list = _ROCM

    def forward(self, query, compressed_kv_cache, k_pe_cache, cu_q_lens, block_tables,
                softmax_scale: Optional[float] = None, cu_total_seq_lens: Optional[torch.Tensor] = None):
        assert_paged_prefill_contract(cu_q_lens, block_tables, cu_total_seq_lens)
        if not self.is_causal:
            raise NotImplementedError("HIPPagedPrefillMLA supports causal attention only")
        return _mla_forward(self, query, compressed_kv_cache, k_pe_cache, block_tables, softmax_scale,
                            cu_q_lens=cu_q_lens, cu_total_seq_lens=cu_total_seq_lens)
