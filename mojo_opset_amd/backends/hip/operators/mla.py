"""HIP<Op> classes of the paged MLA pair (weight-absorbed formulation, see csrc/mla_attn.hip)."""
import math
from typing import Optional

import torch

from ....core.operators.attention import assert_paged_decode_contract, assert_paged_prefill_contract
from ....core.operators.mla import MojoPagedDecodeMLA, MojoPagedPrefillMLA
from .... import switches
from .. import lib as L

_ROCM = ["rocm"]


def _per_head_gemm(x: torch.Tensor, lda: int, out: torch.Tensor, proj: torch.Tensor, heads: int, rows: int, k: int,
                   n: int, w_off: int, w_group: int, w_k: int, w_n: int) -> None:
    """``out[t, h, :n] = x[t, h, :k] @ W[h]`` for token-major ``x`` / ``out`` and W[h] a strided sub-block of
    ``proj``: one GEMM group per head, whose logical row h*rows + t is mapped onto storage row t*heads + h, so neither
    operand is transposed in memory."""
    lib = L.load()
    ws = torch.empty(lib.mojo_hip_group_gemm_workspace_bytes(heads), dtype=torch.uint8, device=x.device)
    w_ptr = L.c_void_p(proj.data_ptr() + w_off * proj.element_size())
    row_map = L.ints4(rows, 1, 0, heads)
    L.check(lib.mojo_hip_group_gemm_strided(L.ptr(x), w_ptr, L.ptr(out), None, 0, heads * rows, k, n, heads,
                                            lda, n, w_group, w_k, w_n, row_map, row_map, L.dtype_code(x.dtype),
                                            L.ptr(ws), ws.numel(), L.stream_of(x)), "hip mla projection")


def _absorbed_k_major(op, proj: torch.Tensor, heads: int, nope: int, vdim: int, r: int) -> torch.Tensor:
    """``W_kn[h]^T`` for every head, contiguous ``[H, r, nope]`` (weight repacking, not per-call compute).

    The copy is keyed on the parameter's storage, version counter, dtype and device, and dropped by everything that goes
    through ``nn.Module`` machinery (``.to()/.cuda()/.half()`` -> ``_apply``, ``load_state_dict``).  A write that bypasses
    the version counter (``param.data.copy_(w)``, a raw pointer write) is invisible to those: call
    ``op.refresh_weights()`` after such an update.  ``MOJO_HIP_VALIDATE=1`` compares the copy with the live parameter
    on every call (one device->host sync) and raises if it is stale."""
    src = op.kv_b_proj                      # (``proj`` may be a per-call contiguous copy of a strided parameter)
    key = (src.data_ptr(), src._version, src.dtype, str(src.device), tuple(src.stride()))
    cached = getattr(op, "_hip_w_kn_t", None)
    if cached is None or cached[0] != key:
        w = proj.view(heads, nope + vdim, r)[:, :nope, :].transpose(1, 2).contiguous()
        cached = (key, w)
        op._hip_w_kn_t = cached
    elif switches.get("MOJO_HIP_VALIDATE", "0") == "1":
        live = proj.view(heads, nope + vdim, r)[:, :nope, :].transpose(1, 2)
        if not torch.equal(cached[1], live):
            raise RuntimeError("HIP MLA: kv_b_proj changed without a version bump (e.g. through .data); call "
                               "op.refresh_weights() after such an update")
    return cached[1]


class _AbsorbedWeightCache:
    """Mixin of the two HIP MLA classes: lifetime of the repacked ``W_kn`` copy (see `_absorbed_k_major`)."""

    def refresh_weights(self) -> None:
        """Drop the K-major copy of the absorbed key projection; the next decode-sized call rebuilds it."""
        self._hip_w_kn_t = None

    # (explicit base calls, not zero-argument super(): mojo_opset_amd.plugin re-bases these functions onto the
    # reference's classes, where the defining class is no longer in the instance's MRO)
    def _apply(self, fn, *args, **kwargs):
        self._hip_w_kn_t = None
        return torch.nn.Module._apply(self, fn, *args, **kwargs)

    def _load_from_state_dict(self, *args, **kwargs):
        self._hip_w_kn_t = None
        return torch.nn.Module._load_from_state_dict(self, *args, **kwargs)


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
    if not proj.is_contiguous():           # (a strided parameter is copied per call; the repack cache is keyed on the original)
        proj = proj.contiguous()
    page = ckv_cache.shape[2]
    scale = 1.0 / math.sqrt(nope + rope) if softmax_scale is None else float(softmax_scale)
    dev = query.device
    if tq == 0:
        return torch.zeros(0, heads, vdim, dtype=query.dtype, device=dev)

    # 1) absorb W_kn into the query:  q_abs[t, h] = q_nope[t, h] @ W_kn[h]   (W_kn[h]: [nope, r], k rows)
    if query.stride(2) != 1 or query.stride(1) != qk or query.stride(0) != heads * qk:
        query = query.contiguous()
    q_abs = torch.empty(tq, heads, r, dtype=query.dtype, device=dev)
    if tq <= 128 and nope % 128 == 0 and r % 64 == 0:
        # decode-sized: the weight-streaming grouped GEMM wants K-major weights; W_kn is stored [nope, r] per head, so a
        # K-major copy [H, r, nope] is kept next to the parameter (a one-time repack, redone when the parameter changes)
        w_t = _absorbed_k_major(op, proj, heads, nope, vdim, r)
        _per_head_gemm(query, qk, q_abs, w_t, heads, tq, nope, r, 0, r * nope, 1, nope)
    else:
        _per_head_gemm(query, qk, q_abs, proj, heads, tq, nope, r, 0, (nope + vdim) * r, r, 1)
    q_rope, q_rope_ld = query[..., nope:], qk                                      # read in place by the kernel
    if nope % 8 or qk % 8:
        q_rope, q_rope_ld = q_rope.contiguous(), rope

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
        L.ptr(q_abs), r, L.ptr(q_rope), q_rope_ld, L.ptr(ckv_cache), L.ptr(kpe_cache),
        L.ptr(None if total_seq_lens is None else total_seq_lens.contiguous()),
        L.ptr(None if cu_q_lens is None else cu_q_lens.contiguous()),
        L.ptr(None if cu_total_seq_lens is None else cu_total_seq_lens.contiguous()),
        L.ptr(tables), L.ptr(sink), L.ptr(o_lat), L.ptr(ws), ws.numel(), tq, batch, heads, r, rope, page, tables.shape[1],
        tables.stride(0), ckv_cache.stride(0), ckv_cache.stride(2), kpe_cache.stride(0), kpe_cache.stride(2), max_len,
        scale, L.dtype_code(query.dtype), L.stream_of(query)), "hip mla attention")

    # 3) out[t, h] = o_lat[t, h] @ W_v[h]^T      (W_v[h]: rows nope.. of head h's block, stored [v, r] = [N, K])
    out = torch.empty(tq, heads, vdim, dtype=query.dtype, device=dev)
    _per_head_gemm(o_lat, r, out, proj, heads, tq, r, vdim, nope * r, (nope + vdim) * r, 1, r)
    return out


def _prefill_decompressed(op, query, ckv_cache, kpe_cache, cu_q_lens, block_tables, softmax_scale, cu_total_seq_lens,
                          max_total_seq_len=None, round_scaled_scores=False, max_q_len=None):
    """Prefill in the golden's own formulation (experimental/operators/attention.py:405-447): un-page the latent,
    decompress K_nope / V for every key with ONE GEMM (rounded to the storage type, like `c_kv @ kv_b_proj.T`), then flash
    attention per head with D_qk = nope + rope, D_v = v (csrc/mla_prefill.hip).  3.4x fewer FLOPs than running the
    absorbed decode kernel per query token at DeepSeek-V3 dimensions, and the golden's rounding points.

    The decompressed image needs `keys * H * (nope + v)` elements (64 KiB per key at DeepSeek-V3 dimensions).  Lengths stay
    on the device, so the host sizes it from what it knows: the table width (`max_total_seq_len` tightens it — pass it), and
    walks the batch in slices of sequences whose image fits `MOJO_HIP_MLA_PREFILL_BYTES` (default 1 GiB: on a serving box
    most of HBM is KV cache, and the block stays in torch's caching allocator).  The per-sequence bound and the image's row capacity
    go to both kernels: a sequence longer than the host's bound loses its OWN tail (cut per sequence, as the paged GQA ops
    truncate theirs; nothing is written or read past the capacity); `MOJO_HIP_VALIDATE=1` raises instead (one device sync).  Returns None when this route does not
    apply (dimensions without an instantiation, or ONE sequence's capacity alone exceeds the budget) — the caller then
    takes the absorbed route."""
    lib = L.load()
    tq, heads, qk = query.shape
    nope, rope, vdim, r = op.qk_nope_head_dim, op.qk_rope_head_dim, op.v_head_dim, op.kv_lora_rank
    if not lib.mojo_hip_mla_prefill_supported(nope, rope, vdim, L.dtype_code(query.dtype)):
        return None
    if switches.get("MOJO_HIP_MLA_PREFILL", "decompress") == "absorbed":
        return None
    dev, dt = query.device, query.dtype
    batch, width = block_tables.shape
    page = ckv_cache.shape[2]
    es = query.element_size()
    kv_cols = heads * (nope + vdim)
    budget = switches.get_int("MOJO_HIP_MLA_PREFILL_BYTES", 1 << 30)
    # host-side bound of one sequence's keys (lengths stay on the device: no sync)
    per_seq = width * page
    if max_total_seq_len is not None:
        per_seq = min(per_seq, int(max_total_seq_len))
    if cu_total_seq_lens is None:
        per_seq = min(per_seq, tq)
    if batch == 0 or per_seq <= 0 or per_seq * kv_cols * es > budget:
        return None
    seqs_per_slice = max(1, min(batch, budget // (per_seq * kv_cols * es)))
    if cu_total_seq_lens is None and tq * kv_cols * es <= budget:
        seqs_per_slice = batch                              # kv = q lengths: the whole batch has exactly tq keys
    proj = op.kv_b_proj.detach()
    proj = proj if proj.is_contiguous() else proj.contiguous()
    if query.stride(2) != 1 or query.stride(1) != qk or query.stride(0) != heads * qk:
        query = query.contiguous()
    tables = block_tables if block_tables.stride(1) == 1 else block_tables.contiguous()
    cu_q = cu_q_lens.contiguous()
    cu_kv = None if cu_total_seq_lens is None else cu_total_seq_lens.contiguous()
    stream = L.stream_of(query)
    out = torch.empty(tq, heads, vdim, dtype=dt, device=dev)
    sink = getattr(op, "attn_sink", None)
    sink = None if sink is None else sink.detach().to(torch.float32).contiguous()
    scale = 1.0 / math.sqrt(nope + rope) if softmax_scale is None else float(softmax_scale)
    # (no clamp to the cache's own token count: sequences that share prefix pages sum to more than the cache holds)
    cap = seqs_per_slice * per_seq
    if cu_total_seq_lens is None:
        cap = min(cap, tq)
    if switches.get("MOJO_HIP_VALIDATE", "0") == "1" and not torch.cuda.is_current_stream_capturing():
        cu = (cu_q if cu_kv is None else cu_kv).to(torch.int64)
        longest = int((cu[1:] - cu[:-1]).max()) if batch > 0 else 0
        if longest > per_seq:
            raise ValueError(f"HIPPagedPrefillMLA: a sequence holds {longest} keys, above the bound {per_seq} the call was "
                             "sized for (block table width x page, or max_total_seq_len)")
    ckv_flat = torch.empty(cap, r, dtype=dt, device=dev)
    kpe_flat = torch.empty(cap, rope, dtype=dt, device=dev)
    kv = torch.empty(cap, kv_cols, dtype=dt, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    # Head groups (opt-in, `op.prefill_head_groups`; measured slower by default, see _head_groups): the decompression of
    # group g + 1 runs on a side stream beside the attention of group g.  Each group's GEMM fills its own
    # columns of the image (`group_gemm_strided`, output row stride = the whole row), the attention launch covers the
    # group's heads only.
    groups = _head_groups(heads, getattr(op, "prefill_head_groups", None))
    cols_g = (heads // groups) * (nope + vdim)
    wss = [torch.empty(lib.mojo_hip_group_gemm_workspace_bytes(1), dtype=torch.uint8, device=dev) for _ in range(groups)]
    main = torch.cuda.current_stream(dev)
    side = _side_stream(dev) if groups > 1 else None
    identity = None
    no_tail = L.ints4(0, 0, 0, 1)                            # an (identity) output row map: the GEMM entry point then leaves the rows past
    #                                                          the slice's key count alone (nobody reads them; zeroing them cost 223 ms
    #                                                          for a 920 MB tail before round 5, and would still be a wasted GB of writes)

    def decompress(g, on):
        # kv[t, g*cols_g + j] = sum_k ckv[t, k] * kv_b_proj[g*cols_g + j, k]: one group whose row count is the slice's
        # device-side number of keys, so rows past it are never computed
        L.check(lib.mojo_hip_group_gemm_strided(
            L.ptr(ckv_flat), L.c_void_p(proj.data_ptr() + g * cols_g * r * es), L.c_void_p(kv.data_ptr() + g * cols_g * es),
            L.ptr(count), 0, cap, r, cols_g, 1, r, kv_cols, 0, 1, r, identity, no_tail, L.dtype_code(dt),
            L.ptr(wss[g]), wss[g].numel(), L.c_void_p(on.cuda_stream)), "hip mla decompression")

    for b0 in range(0, batch, seqs_per_slice):
        nb = min(seqs_per_slice, batch - b0)
        cq = L.c_void_p(cu_q.data_ptr() + 4 * b0)
        ck = None if cu_kv is None else L.c_void_p(cu_kv.data_ptr() + 4 * b0)
        tb = L.c_void_p(tables.data_ptr() + 4 * b0 * tables.stride(0))
        if side is not None and b0 > 0:
            main.wait_stream(side)                          # (the image is rewritten: the previous slice's side work is done)
        L.check(lib.mojo_hip_mla_unpage(L.ptr(ckv_cache), L.ptr(kpe_cache), L.ptr(ckv_flat), L.ptr(kpe_flat), cq, ck, tb,
                                        tables.stride(0), width, nb, r, rope, page, es, ckv_cache.stride(0),
                                        ckv_cache.stride(2), kpe_cache.stride(0), kpe_cache.stride(2), per_seq, cap, L.ptr(count),
                                        stream), "hip mla un-page")
        last = b0 + nb >= batch
        events = []
        if side is not None:
            side.wait_stream(main)                          # the un-paged latent and the key count are ready
            for g in range(1, groups):
                decompress(g, side)
                ev = torch.cuda.Event()
                ev.record(side)
                events.append(ev)
        decompress(0, main)
        for g in range(groups):
            if g > 0:
                main.wait_event(events[g - 1])
            L.check(lib.mojo_hip_mla_prefill_attn(L.ptr(query), L.ptr(kv), L.ptr(kpe_flat), L.ptr(sink), L.ptr(out), cq, ck, tq,
                                                  nb, heads, g * (heads // groups), heads // groups, nope, rope, vdim,
                                                  min(tq, per_seq) if max_q_len is None else max_q_len, per_seq, cap, scale,
                                                  1 if round_scaled_scores else 0, 1 if (last and g == 0) else 0,
                                                  L.dtype_code(dt), stream), "hip mla prefill attention")
    return out


_SIDE_STREAMS = {}


def _side_stream(dev):
    """One side stream per device for the decompression GEMMs that run beside the attention (created once, outside any
    capture when possible; every use is joined back into the caller's stream by an event before the call returns)."""
    key = torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device()
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return st


def _head_groups(heads: int, wanted) -> int:
    """How many head groups the decompression / attention pipeline uses.  Default ONE (a single GEMM, then a single attention
    launch): measured on an MI355X at DeepSeek-V3 dims (scripts/probes/mla_prefill_groups_ab.py, round 4; same bits for every
    group count): 4 x 512 211 us as one group, 237 us as two, 235-246 us as four; + 2048 cached 1 000 / 971 / 1 007 us.  The
    two kernels do not share the chip — the GEMM's 256 x 256 tiles occupy every CU, a half-width GEMM runs at the same
    tiles per second — so the side stream buys at most 3 % on the long case and costs 12 % on the short one.
    ``op.prefill_head_groups = <n>`` selects the pipeline (an attribute, not an environment switch: an experiment kept for
    its test, not a tuning knob)."""
    if wanted:
        g = max(1, int(wanted))
        while g > 1 and heads % g:
            g -= 1
        return g
    return 1


def _decode_decompressed(op, query, ckv_cache, kpe_cache, total_seq_lens, block_tables, softmax_scale, max_total_seq_len):
    """Decode in the golden's own formulation (experimental/operators/attention.py:184-220): every sequence is a one-token
    "prefill" over its cached keys — un-page, decompress K_nope / V with the GEMM (rounded to the storage type like
    `c_kv @ kv_b_proj.T`), scores rounded to the storage type before the scale, probabilities rounded to the storage type.
    It spends 64 KiB of HBM traffic per key where the absorbed kernel spends 1.1 KiB, so it is the selectable PARITY route
    (`MOJO_HIP_MLA_DECODE=golden` or `op.decode_route = "golden"`), not the default: it reproduces the golden's rounding
    points and holds the reference test's atol = rtol = 1e-2 against the golden, which the (more accurate) absorbed
    form cannot where the golden's own rounding error exceeds that bound.  Returns None when the route does not apply."""
    batch = query.shape[0]
    dev = query.device
    lens = total_seq_lens.to(torch.int32).clamp(min=0)          # (out of place: `.to` returns the caller's tensor when it is int32 already)
    cu_kv = torch.zeros(batch + 1, dtype=torch.int32, device=dev)
    torch.cumsum(lens, 0, out=cu_kv[1:])
    cu_q = torch.arange(batch + 1, dtype=torch.int32, device=dev)
    return _prefill_decompressed(op, query, ckv_cache, kpe_cache, cu_q, block_tables, softmax_scale, cu_kv, max_total_seq_len,
                                 round_scaled_scores=True, max_q_len=1)


class HIPPagedDecodeMLA(_AbsorbedWeightCache, MojoPagedDecodeMLA):
    supported_platforms_list = _ROCM

    def forward(self, query, compressed_kv_cache, k_pe_cache, total_seq_lens, block_tables,
                softmax_scale: Optional[float] = None, *, max_total_seq_len: Optional[int] = None):
        """``max_total_seq_len`` (extension, kw-only host int): upper bound of any sequence's length; only the
        golden-rounding route uses it (it sizes the decompressed K/V image)."""
        assert_paged_decode_contract(block_tables, total_seq_lens)
        route = getattr(self, "decode_route", None) or switches.get("MOJO_HIP_MLA_DECODE", "absorbed")
        if route == "golden" and query.shape[0] > 0 and query.is_cuda \
                and self.kv_b_proj.dtype == query.dtype == compressed_kv_cache.dtype == k_pe_cache.dtype \
                and compressed_kv_cache.stride(3) == 1 and k_pe_cache.stride(3) == 1:
            L.require_cuda(query, compressed_kv_cache, k_pe_cache, total_seq_lens, block_tables, self.kv_b_proj)
            if switches.get("MOJO_HIP_VALIDATE", "0") == "1" and block_tables.shape[1] > 0 \
                    and bool(((total_seq_lens > 0) & (block_tables[:, 0] < 0)).any()):
                raise ValueError("Paged decode requires a valid block table for rows with kv lens > 0.")
            out = _decode_decompressed(self, query, compressed_kv_cache, k_pe_cache, total_seq_lens, block_tables,
                                       softmax_scale, max_total_seq_len)
            if out is not None:
                return out
        return _mla_forward(self, query, compressed_kv_cache, k_pe_cache, block_tables, softmax_scale,
                            total_seq_lens=total_seq_lens)


class HIPPagedPrefillMLA(_AbsorbedWeightCache, MojoPagedPrefillMLA):
    supported_platforms_list = _ROCM

    def forward(self, query, compressed_kv_cache, k_pe_cache, cu_q_lens, block_tables,
                softmax_scale: Optional[float] = None, cu_total_seq_lens: Optional[torch.Tensor] = None, *,
                max_total_seq_len: Optional[int] = None):
        """``max_total_seq_len`` (extension, kw-only host int like the paged GQA ops take): an upper bound of any sequence's
        total length; without it the block table's width bounds the size of the decompressed K/V image."""
        assert_paged_prefill_contract(cu_q_lens, block_tables, cu_total_seq_lens)
        if not self.is_causal:
            raise NotImplementedError("HIPPagedPrefillMLA supports causal attention only")
        if query.shape[0] > 0 and query.is_cuda and self.kv_b_proj.dtype == query.dtype == compressed_kv_cache.dtype == k_pe_cache.dtype \
                and compressed_kv_cache.stride(3) == 1 and k_pe_cache.stride(3) == 1 and query.shape[0] >= 16:
            L.require_cuda(query, compressed_kv_cache, k_pe_cache, cu_q_lens, block_tables, cu_total_seq_lens, self.kv_b_proj)
            out = _prefill_decompressed(self, query, compressed_kv_cache, k_pe_cache, cu_q_lens, block_tables, softmax_scale,
                                        cu_total_seq_lens, max_total_seq_len)
            if out is not None:
                return out
        return _mla_forward(self, query, compressed_kv_cache, k_pe_cache, block_tables, softmax_scale,
                            cu_q_lens=cu_q_lens, cu_total_seq_lens=cu_total_seq_lens)
