"""GEMM + collective pipelines for the four ComputeComm operators.

Each op is a local GEMM plus exactly one exchange step over the tensor-parallel group.  The golden
(`mojo_opset/core/operators/compute_with_comm.py:96-111, :160-176, :234-253, :316-332`) runs them back to
back: one GEMM, one blocking collective, no overlap.  Here the GEMM is cut into row chunks and the chunk's
collective is enqueued asynchronously as soon as the chunk is computed — `torch.distributed` (backend
"nccl" = RCCL) runs it on the process group's own HIP stream, ordered after the GEMM by an event — so the
xGMI transfer of chunk c overlaps the matrix work of chunk c+1:

    all-reduce      : chunk c of the output rows   -> all_reduce(out[rows_c])            (in place)
    reduce-scatter  : sub-chunk c of EVERY rank's row block, produced contiguously by the GEMM's A-row map
                      -> reduce_scatter_tensor(out[rows_c], buf_c)
    all-gather      : sub-chunk c of every rank's input rows -> all_gather_into_tensor(buf_c, x[rows_c]);
                      the GEMM on buf_c writes its rows to their final place through the C-row map
    all-to-all      : (scatter along rows) sub-chunk c of every destination's row block, produced contiguously by the
                      A-row map -> all_to_all of that chunk, received in place; other scatter dims: one exchange

The pipelines only orchestrate; the matrix product is delegated to a `GemmEngine` (the hip backend passes
the C-ABI GEMM; the multi-process CPU tests pass a torch engine over gloo).
"""
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .. import switches

CHIP_TILE_SLOTS = 256          # workgroups of the 256 x 256 GEMM tile kernel resident at once on an MI355X (one per CU)
MAX_CHUNKS = 8


class GemmEngine:
    """``out[c_map(m)] = x[a_map(m)] @ W (+ bias)`` for logical rows ``m in [0, rows)``.

    ``a_map`` / ``c_map`` are ``(rc, ml, off)`` triples meaning ``row = (m // rc) * ml + off + m % rc``
    (``None`` = identity).  ``weight`` is ``[K, N]`` when ``trans_weight`` else ``[N, K]``.
    """

    def __call__(self, x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], trans_weight: bool, *,
                 out: Optional[torch.Tensor] = None, rows: Optional[int] = None,
                 a_map: Optional[Tuple[int, int, int]] = None,
                 c_map: Optional[Tuple[int, int, int]] = None) -> torch.Tensor:
        raise NotImplementedError

    @staticmethod
    def out_features(weight: torch.Tensor, trans_weight: bool) -> int:
        return weight.shape[1] if trans_weight else weight.shape[0]


def chunk_count(rows: int, n_cols: int, world: int = 1, gemm_rows_per_row: int = 1, elt_bytes: int = 2,
                link_cols: Optional[int] = None) -> int:
    """How many row chunks a pipeline cuts ``rows`` into — a function of the payload and the world size (round 5; it was
    the constant 4).  Two floors on the rows of a chunk, then as many chunks as fit (at most ``MAX_CHUNKS``):

    * MATRIX WORK: a chunk's GEMM is its own launch, and a launch of the 256 x 256 tile kernel takes one round of the chip
      whether it has 64 tiles or 256 (one workgroup per CU).  Four chunks of 1024 rows at N = 8192 are four rounds of 128
      tiles where two chunks of 2048 rows are two full rounds: chunking finer than one full round DOUBLES the matrix time it
      is meant to hide the exchange behind.  ``gemm_rows_per_row`` = GEMM rows per planned row (reduce-scatter, all-gather
      and all-to-all plan over the M / ws rows one rank keeps, and a chunk's GEMM covers that sub-chunk of EVERY rank).
    * EXCHANGE: the message one rank moves to / from ONE peer for a chunk (the ring's step, the direct exchange's pull) is
      the chunk's payload / ws; below ~512 KiB a transfer is latency-, not bandwidth-bound on xGMI.  ``link_cols`` = columns
      of the tensor that travels when that is not the GEMM's output (all-gather moves the [rows, K] input).
    """
    if rows <= 0:
        return 0
    tiles_per_256_rows = max(1, gemm_rows_per_row) * max(1, -(-n_cols // 256))
    rows_round = -(-CHIP_TILE_SLOTS // tiles_per_256_rows) * 256
    rows_link = -(-(512 << 10) * max(1, world) // max(1, gemm_rows_per_row * (link_cols or n_cols) * elt_bytes))
    floor_rows = max(256, rows_round, -(-rows_link // 256) * 256)
    return max(1, min(MAX_CHUNKS, rows // floor_rows))


def plan_row_chunks(rows: int, n_cols: int = 0, world: int = 1, gemm_rows_per_row: int = 1, elt_bytes: int = 2,
                    link_cols: Optional[int] = None) -> List[Tuple[int, int]]:
    """Split ``rows`` into contiguous ranges of whole GEMM tiles (multiples of 256 rows); the count is ``chunk_count`` of the
    payload and world size, ``MOJO_HIP_COMM_CHUNKS=<n>`` forces it (still at least 512 rows per chunk unless n = 1)."""
    if rows <= 0:
        return []
    forced = switches.get_int("MOJO_HIP_COMM_CHUNKS", 0)
    if forced > 0:
        want = max(1, min(forced, rows // 512 if rows >= 512 else 1))
    else:
        want = chunk_count(rows, n_cols if n_cols > 0 else 8192, world, gemm_rows_per_row, elt_bytes, link_cols)
    step = -(-rows // want)
    step = -(-step // 256) * 256 if rows >= 256 else step
    out, lo = [], 0
    while lo < rows:
        hi = min(rows, lo + step)
        out.append((lo, hi))
        lo = hi
    return out


def _group_info(group):
    return dist.get_world_size(group), dist.get_rank(group)


class _Done:
    """Work handle of a collective that already completed (the synchronous host-staged gloo route below)."""

    def wait(self):
        return True


class _Staged:
    """Work handle of an ASYNCHRONOUS gloo collective on a host copy of a device tensor: ``wait()`` waits for the
    collective and then copies the host result to its device destination (``finish`` does the copy)."""

    def __init__(self, work, finish):
        self._work, self._finish = work, finish

    def wait(self):
        self._work.wait()
        self._finish()
        return True


def _host_staged(group, t: torch.Tensor) -> bool:
    """gloo moves device tensors through the host anyway, and its asynchronous DEVICE-tensor collectives proved unreliable
    when several are in flight (an `all_gather_into_tensor` of a 33 MB device tensor hung both ranks of the two-rank GPU
    test once in three runs).  On a gloo group the exchange therefore runs on explicit HOST copies.  Only the single-GPU
    multi-rank TESTS use gloo with device tensors; on a node the group is RCCL and this is never taken."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


STAGED_ASYNC = True            # (module attribute, not an environment switch: only the single-GPU multi-rank tests take this route)


def _staged_async() -> bool:
    """On the host-staged route the collective itself is still asynchronous by default (gloo's own threads reduce chunk c
    while the GEMM of chunk c + 1 runs on the device, several chunks in flight, waited for in order at the end — the same
    issue / wait ordering the RCCL route has).  ``pipelines.STAGED_ASYNC = False`` makes it synchronous."""
    return STAGED_ASYNC


def _all_reduce(t, group):
    if _host_staged(group, t):
        h = t.cpu()                                                     # (waits for this chunk's GEMM only)
        if _staged_async():
            return _Staged(dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group, async_op=True), lambda: t.copy_(h))
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
        return _Done()
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True)


def _reduce_scatter_tensor(out, buf, group):
    if _host_staged(group, buf):
        ws, rank = _group_info(group)
        h = buf.cpu()
        rows = h.shape[0] // ws

        def finish():
            out.copy_(h[rank * rows:(rank + 1) * rows])

        if _staged_async():                                             # gloo has no reduce_scatter for every dtype
            return _Staged(dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group, async_op=True), finish)
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        finish()
        return _Done()
    return dist.reduce_scatter_tensor(out, buf, op=dist.ReduceOp.SUM, group=group, async_op=True)


def _all_gather_into_tensor(buf, x, group):
    if _host_staged(group, x):
        h = torch.empty(buf.shape, dtype=buf.dtype)
        src = x.cpu().contiguous()
        if _staged_async():
            return _Staged(dist.all_gather_into_tensor(h, src, group=group, async_op=True), lambda: buf.copy_(h))
        dist.all_gather_into_tensor(h, src, group=group)
        buf.copy_(h)
        return _Done()
    return dist.all_gather_into_tensor(buf, x.contiguous(), group=group, async_op=True)


def _flatten(x: torch.Tensor) -> torch.Tensor:
    x2 = x.reshape(-1, x.shape[-1])
    return x2 if x2.stride(-1) == 1 else x2.contiguous()


# ---------------------------------------------------------------------------------------------------------
def gemm_all_reduce(engine: GemmEngine, x, weight, bias, trans_weight, group) -> torch.Tensor:
    """allreduce_sum(x @ W (+ bias)); the bias joins every rank's partial, as in the golden (:106-110)."""
    n = engine.out_features(weight, trans_weight)
    x2 = _flatten(x)
    m = x2.shape[0]
    out = torch.empty(m, n, dtype=x.dtype, device=x.device)
    if group is None:
        engine(x2, weight, bias, trans_weight, out=out)
        return out.reshape(*x.shape[:-1], n)
    works = []
    ws, _ = _group_info(group)
    for lo, hi in plan_row_chunks(m, n, ws, 1, x.element_size()):
        engine(x2[lo:hi], weight, bias, trans_weight, out=out[lo:hi])
        works.append(_all_reduce(out[lo:hi], group))
    for w in works:
        w.wait()
    return out.reshape(*x.shape[:-1], n)


def gemm_reduce_scatter(engine: GemmEngine, x, weight, bias, trans_weight, group, scatter_dim: int) -> torch.Tensor:
    """This rank's ``scatter_dim`` chunk of sum_ranks(x @ W (+ bias))."""
    n = engine.out_features(weight, trans_weight)
    if group is None:
        return engine(_flatten(x), weight, bias, trans_weight).reshape(*x.shape[:-1], n)
    ws, rank = _group_info(group)
    out_shape = list(x.shape[:-1]) + [n]
    sd = scatter_dim % len(out_shape)
    if out_shape[sd] % ws != 0:
        raise ValueError(f"reduce_scatter: dimension {sd} of size {out_shape[sd]} is not divisible by world size {ws}")
    if sd != 0:
        # not row-blocked in the flattened product: compute it whole, exchange contiguous copies of the chunks
        y = engine(_flatten(x), weight, bias, trans_weight).reshape(out_shape)
        chunks = [c.contiguous() for c in y.chunk(ws, dim=sd)]
        mine = torch.empty_like(chunks[rank])
        dist.reduce_scatter(mine, chunks, op=dist.ReduceOp.SUM, group=group)
        return mine
    x2 = _flatten(x)
    m = x2.shape[0]
    ml = m // ws                                   # rows every rank keeps
    out = torch.empty(ml, n, dtype=x.dtype, device=x.device)
    works, keep = [], []
    for lo, hi in plan_row_chunks(ml, n, ws, ws, x.element_size()):
        rc = hi - lo
        buf = torch.empty(ws * rc, n, dtype=x.dtype, device=x.device)   # [dest rank][rc rows]
        engine(x2, weight, bias, trans_weight, out=buf, rows=ws * rc, a_map=(rc, ml, lo))
        works.append(_reduce_scatter_tensor(out[lo:hi], buf, group))
        keep.append(buf)
    for w in works:
        w.wait()
    out_shape[0] //= ws
    return out.reshape(out_shape)


def all_gather_gemm(engine: GemmEngine, x, weight, bias, trans_weight, group, gather_dim: int) -> torch.Tensor:
    """allgather(x, gather_dim) @ W (+ bias)."""
    n = engine.out_features(weight, trans_weight)
    if group is None:
        return engine(_flatten(x), weight, bias, trans_weight).reshape(*x.shape[:-1], n)
    ws, _ = _group_info(group)
    gd = gather_dim % x.dim()
    if gd != 0 or gd == x.dim() - 1:
        parts = [torch.empty_like(x) for _ in range(ws)]
        dist.all_gather(parts, x.contiguous(), group=group)
        full = torch.cat(parts, dim=gd)
        return engine(_flatten(full), weight, bias, trans_weight).reshape(*full.shape[:-1], n)
    x2 = _flatten(x)
    ml, k = x2.shape
    out = torch.empty(ws * ml, n, dtype=x.dtype, device=x.device)
    chunks = plan_row_chunks(ml, n, ws, ws, x.element_size(), link_cols=k)
    stages = []
    for lo, hi in chunks:                          # enqueue every gather first: they run back to back on the comm stream
        buf = torch.empty(ws * (hi - lo), k, dtype=x.dtype, device=x.device)
        stages.append((buf, _all_gather_into_tensor(buf, x2[lo:hi], group)))
    for (lo, hi), (buf, work) in zip(chunks, stages):
        work.wait()                                # the GEMM of chunk c overlaps the gathers of chunks c+1..
        engine(buf, weight, bias, trans_weight, out=out, rows=buf.shape[0], c_map=(hi - lo, ml, lo))
    shape = list(x.shape[:-1]) + [n]
    shape[0] *= ws
    return out.reshape(shape)


def _all_to_all_rows(recv_views, send_views, group, ws, rank):
    """Asynchronous all-to-all of row blocks: ``send_views[d]`` goes to rank d, ``recv_views[s]`` receives rank s's block
    for this rank (both lists of ``ws`` contiguous row ranges).  RCCL: one grouped send/recv, asynchronous.  gloo (the
    single-GPU / CPU tests; no all-to-all there): an all-gather of every rank's send buffer on the host, this rank's blocks
    picked out — asynchronous as well when the staged route is (``_staged_async``)."""
    if dist.get_backend(group) != "gloo":
        return dist.all_to_all(list(recv_views), list(send_views), group=group, async_op=True)
    mine = torch.stack([v.cpu() if v.is_cuda else v for v in send_views])           # [ws, rows, n] on the host
    everyone = [torch.empty_like(mine) for _ in range(ws)]

    def finish():
        for src in range(ws):
            recv_views[src].copy_(everyone[src][rank])

    if _staged_async():
        return _Staged(dist.all_gather(everyone, mine, group=group, async_op=True), finish)
    dist.all_gather(everyone, mine, group=group)
    finish()
    return _Done()


def gemm_all2all(engine: GemmEngine, x, weight, bias, trans_weight, group, scatter_dim: int, gather_dim: int) -> torch.Tensor:
    """cat(all_to_all(chunk(x @ W (+ bias), ws, scatter_dim)), gather_dim)  (golden: compute_with_comm.py:234-253).

    Scattering along dimension 0 (the Ulysses switch from sequence- to head-sharding) is ROW-blocked in the flattened
    product, so it gets the same row-chunk pipeline as the other three operators: for sub-chunk c of every destination's
    row block the GEMM's A-row map produces ``[dest rank][rc rows]`` contiguously, that chunk's all-to-all is enqueued
    asynchronously and overlaps the matrix work of chunk c + 1; every source's rows land directly in their final place of
    the ``[source rank][rows]`` receive buffer (no copy when ``gather_dim == 0``; one concatenation otherwise).  Any other
    scatter dimension computes the product whole and exchanges contiguous copies, as the golden does."""
    n = engine.out_features(weight, trans_weight)
    if group is None:
        return engine(_flatten(x), weight, bias, trans_weight).reshape(*x.shape[:-1], n)
    ws, rank = _group_info(group)
    out_shape = list(x.shape[:-1]) + [n]
    sd, gd = scatter_dim % len(out_shape), gather_dim % len(out_shape)
    if out_shape[sd] % ws != 0:
        raise ValueError(f"all_to_all: dimension {sd} of size {out_shape[sd]} is not divisible by world size {ws}")
    if sd != 0 or len(out_shape) < 2:
        y = engine(_flatten(x), weight, bias, trans_weight).reshape(out_shape)
        send = [c.contiguous() for c in y.chunk(ws, dim=sd)]
        pieces = [torch.empty_like(c) for c in send]
        _all_to_all_list(list(pieces), send, group, ws, rank)
        return pieces[0] if ws == 1 else torch.cat(list(pieces), dim=gd)
    x2 = _flatten(x)
    m = x2.shape[0]
    ml = m // ws                                   # rows of the flattened product every destination receives from this rank
    recv = torch.empty(m, n, dtype=x.dtype, device=x.device)          # [source rank][ml rows]
    works, keep = [], []
    for lo, hi in plan_row_chunks(ml, n, ws, ws, x.element_size()):
        rc = hi - lo
        buf = torch.empty(ws * rc, n, dtype=x.dtype, device=x.device)  # [dest rank][rc rows]
        engine(x2, weight, bias, trans_weight, out=buf, rows=ws * rc, a_map=(rc, ml, lo))
        works.append(_all_to_all_rows([recv[s * ml + lo: s * ml + hi] for s in range(ws)],
                                      [buf[d * rc: (d + 1) * rc] for d in range(ws)], group, ws, rank))
        keep.append(buf)
    for w in works:
        w.wait()
    piece_shape = list(out_shape)
    piece_shape[0] //= ws
    if gd == 0:
        piece_shape[0] *= ws
        return recv.reshape(piece_shape)
    return torch.cat([recv[s * ml: (s + 1) * ml].reshape(piece_shape) for s in range(ws)], dim=gd)


def _all_to_all_list(recv, send, group, ws, rank):
    try:
        dist.all_to_all(recv, send, group=group)
    except RuntimeError:
        stacked = torch.stack(send)
        everyone = [torch.empty_like(stacked) for _ in range(ws)]
        dist.all_gather(everyone, stacked, group=group)
        for src in range(ws):
            recv[src].copy_(everyone[src][rank])
