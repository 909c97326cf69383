"""GPU tests of the GEMM + collective operators through the C-ABI GEMM.

A single MI355X is available to the test box, so the exchange step is exercised (a) with torch.distributed
uninitialised (identity, as the golden specifies) and (b) inside a world_size = 1 RCCL group, which runs
the real chunked pipeline — async collectives on the process group's stream, row-mapped GEMM launches —
with results that must equal the plain product.  The world_size = 2 behaviour of the same pipelines is
covered on CPU over gloo (tests/test_comm_gloo.py)."""
import os

import pytest
import torch
import torch.distributed as dist

from hip_utils import DEV, hip_cls, max_ulp_bf16ish, switch_env, to_cpu, torch_cls

pytestmark = pytest.mark.gpu


def _mk(m, k, n, trans, dtype, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(m, k, generator=g).to(dtype)
    w = (torch.randn(k, n, generator=g) * 0.1).to(dtype) if trans else (torch.randn(n, k, generator=g) * 0.1).to(dtype)
    return x, w


CASES = [(4096, 4096, 4096), (2048, 8192, 4096), (8192, 4096, 2048), (300, 1024, 520), (32, 64, 128)]


def _run_all(dtype, tol):
    for m, k, n in CASES:
        for trans in (True, False):
            x, w = _mk(m, k, n, trans, dtype, seed=m)
            want = (x.double() @ (w.double() if trans else w.double().t())).to(dtype)
            xd, wd = x.to(DEV), w.to(DEV)
            outs = {
                "allreduce": hip_cls("MojoGemmAllReduce")(wd, None, trans)(xd),
                "reducescatter": hip_cls("MojoGemmReduceScatter")(wd, None, trans, scatter_dim=0)(xd),
                "allgather": hip_cls("MojoAllGatherGemm")(wd, None, trans, gather_dim=0)(xd),
                "all2all": hip_cls("MojoGemmAll2All")(wd, None, trans, scatter_dim=0, gather_dim=1)(xd),
            }
            for name, got in outs.items():
                got = to_cpu(got)
                assert got.shape == want.shape, (name, got.shape, want.shape)
                if dtype == torch.float32:
                    torch.testing.assert_close(got, want, atol=tol, rtol=tol, msg=lambda s: f"{name} {m}x{k}x{n}: {s}")
                else:
                    assert max_ulp_bf16ish(got, want, atol=1e-3) <= 1, (name, m, k, n, trans)


def test_comm_ops_identity_when_dist_is_not_initialised():
    assert not dist.is_initialized()
    _run_all(torch.bfloat16, None)


def test_comm_ops_bias_is_added_like_the_golden():
    x, w = _mk(512, 256, 384, True, torch.bfloat16)
    b = torch.randn(384, generator=torch.Generator().manual_seed(7)).to(torch.bfloat16)
    want = torch.as_tensor(torch_cls("MojoGemmAllReduce")(w, b, True)(x))
    got = to_cpu(hip_cls("MojoGemmAllReduce")(w.to(DEV), b.to(DEV), True)(x.to(DEV)))
    # product rounded to bf16, THEN the bias (two roundings, like the golden): one ulp of the product may survive,
    # and where the bias cancels the product that ulp is large relative to the result -> absolute bound of one
    # bf16 ulp at the product's magnitude (|x@w| < 8 here), not a relative one.
    torch.testing.assert_close(got.float(), want.float(), atol=2.0 ** -5, rtol=2.0 ** -7)
    assert (got != want).float().mean() < 0.02
    with pytest.raises(TypeError):
        hip_cls("MojoGemmAllReduce")(w.to(DEV), None, trans_weight="yes")


def test_comm_ops_inside_a_single_rank_rccl_group():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV, 0))
    try:
        for chunks in ("1", "4"):
            with switch_env(MOJO_HIP_COMM_CHUNKS=chunks):
                _run_all(torch.bfloat16, None)
        _run_all(torch.float32, 2e-3)
        # fp32 reference shape of the all-to-all test (test_compute_with_comm.py:215-247)
        x, w = _mk(32, 64, 128, True, torch.float32, seed=5)
        got = hip_cls("MojoGemmAll2All")(w.to(DEV), None, True, scatter_dim=0, gather_dim=0)(x.to(DEV))
        torch.testing.assert_close(to_cpu(got), x @ w, atol=1e-4, rtol=1e-4)
    finally:
        dist.destroy_process_group()
