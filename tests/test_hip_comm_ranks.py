"""GPU test of the four GEMM + collective operators at world_size = 2 (the reference's own shape:
`tests/accuracy/operators/test_compute_with_comm.py:47-70` starts its ranks as a child `torch.distributed.run`).

Two child processes share the box's one MI355X (RCCL refuses two ranks on a device, so the process group is gloo; with
`direct` modes the reduce exchange is the HIP-IPC pull-and-add path, which needs no RCCL at all).  Each rank runs
tests/comm_rank_worker.py: per-rank reference vectors + the oracle at the reference's shapes, chunk counts 1 and 4.
The ranks are CHILDREN of this process (never an exec of a process that touched the GPU)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_ranks(modes, ws=2, timeout=900):
    port = _free_port()
    procs = []
    for rank in range(ws):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(ws), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), MOJO_TEST_COMM_MODES=modes, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "comm_rank_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=timeout))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            outs.append(p.communicate())
    for rank, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} exited {p.returncode}\n--- stdout\n{so[-3000:]}\n--- stderr\n{se[-6000:]}"
    return [json.loads(ln) for ln in outs[0][0].splitlines() if ln.startswith("{")]


def test_hip_compute_comm_two_ranks_rccl_pipeline_layout():
    recs = run_ranks("chunks1,chunks4")
    checks = [r for r in recs if "check" in r]
    names = {r["check"].split(":")[1] for r in checks}
    assert {"MojoGemmAllReduce", "MojoGemmReduceScatter", "MojoAllGatherGemm", "MojoGemmAll2All"} <= names, names
    assert len(checks) == 2 * (5 + 8 + 4 + 1), len(checks)
    print("\n".join(json.dumps(r) for r in recs))


def test_hip_compute_comm_two_ranks_direct_peer_exchange():
    recs = run_ranks("direct1,direct4")
    assert any(r.get("check", "").startswith("oracle:MojoGemmAllReduce") for r in recs)
    print("\n".join(json.dumps(r) for r in recs))
