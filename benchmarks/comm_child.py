"""One rank of the GEMM + collective benchmark, run as a CHILD of a `bench.py` rank (benchmarks/extras.py::_comm_in_children).

Why a child: the direct peer exchange has never run on real xGMI.  A fault inside it (a bad peer mapping aborts the process
from the HIP runtime) must cost the `compute_comm_bf16` block, not the rank that still has to print the result line; a hang is
ended by the parent's deadline.  The children form their own process group on a port the parents agreed on.

    python benchmarks/comm_child.py <result.json>      (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT in the env)
"""
import faulthandler
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    out_path = sys.argv[1]
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    faulthandler.dump_traceback_later(int(os.environ.get("MOJO_BENCH_COMM_DUMP_AFTER_S", "240")), exit=False)
    n_dev = max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local % n_dev)
    device = torch.device("cuda", local % n_dev)
    backend = os.environ.get("MOJO_BENCH_DIST_BACKEND", "nccl")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=device)
    else:
        dist.init_process_group(backend)
    from benchmarks.extras import bench_compute_comm

    res = bench_compute_comm(device, world, rank)
    torch.cuda.synchronize()
    if rank == 0:
        with open(out_path + ".tmp", "w") as f:
            json.dump(res, f)
        os.replace(out_path + ".tmp", out_path)
    try:
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        pass
    sys.stdout.flush()
    os._exit(0)                                  # no interpreter teardown: a peer that died must not leave this rank in a collective


if __name__ == "__main__":
    main()
