"""bench.py — headline benchmark: MojoPagedDecodeGQA, Llama-3-8B shape, on MI355X.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A *step* is one pass of the hot path over one batch: paged decode attention for B=64 sequences
(32 q heads / 8 kv heads, head_dim 128, page 16, bf16, ctx 4096 each, block tables a random
permutation) with all inputs resident in HBM.  ``value`` = tokens/s = B * steps / time, whole job.
With N > 1 every rank runs its own batch (the op has no exchange step: replicas, weak scaling).

Extra objects on the JSON line:
  roofline     — HBM roofline of the decode op from HIP-event time of the timed region.
  cpu_baseline — the torch-native oracle (a port of the reference's golden backend) timed on the
                 host cores on a bounded sample of the same workload.
  extras       — other hot-path ops measured in the same run (absolute rate + roofline fraction).
"""
import argparse
import json
import threading
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)
MFMA_BF16_PEAK_TFLOPS = 2500.0

B, HQ, HKV, D, PAGE, CTX = 64, 32, 8, 128, 16, 4096


def launch_ranks(gpus, argv):
    """``python bench.py --gpus N`` without a launcher: start ``torch.distributed.run`` with N ranks as a CHILD process
    (this parent never touches the GPU — it has not even imported torch — so nothing that initialised HIP is replaced
    or forked), relay the child's output and exit with its status."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=sys.stderr, text=True)
    got_line = False
    for ln in proc.stdout:
        if ln.startswith("{") and '"metric"' in ln:
            got_line = True
        sys.stdout.write(ln)
        sys.stdout.flush()
    rc = proc.wait()
    if rc == 0 and not got_line:
        print("bench.py: the ranks exited without printing a result line", file=sys.stderr)
        rc = 4
    sys.exit(rc)


def _early_launch():
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("--gpus", type=int, default=1)
    ns, _ = ap.parse_known_args()
    if ns.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(ns.gpus, sys.argv[1:])


if __name__ == "__main__":
    _early_launch()                # before torch is even imported: the parent of the ranks never initialises HIP

import torch  # noqa: E402


def build_decode_inputs(device, ctx=CTX, batch=B, seed=20260716, sets=2):
    g = torch.Generator().manual_seed(seed)
    pages_per_seq = ctx // PAGE
    n_pages = batch * pages_per_seq + 10
    out = []
    for s in range(sets):
        q = torch.randn(batch, HQ, D, generator=g).to(torch.bfloat16).to(device)
        k = torch.randn(n_pages, HKV, PAGE, D, device=device, dtype=torch.bfloat16)
        v = torch.randn(n_pages, HKV, PAGE, D, device=device, dtype=torch.bfloat16)
        table = torch.randperm(n_pages, generator=g, dtype=torch.int32)[: batch * pages_per_seq].view(batch, -1)
        lens = torch.full((batch,), ctx, dtype=torch.int32)
        out.append((q, k, v, lens.to(device), table.to(device)))
    return out


def decode_algorithmic_bytes(lens, max_blocks):
    kv = int(lens.sum()) * HKV * D * 2 * 2
    qo = 2 * lens.numel() * HQ * D * 2
    idx = 4 * lens.numel() * (max_blocks + 1)
    return kv + qo + idx


def _profiled(name, key):
    """A value from a committed rocprofv3 summary under profiles/ (None when absent)."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name))).get(key)
    except Exception:
        return None


def usable_cores():
    """Cores this process may actually use: the affinity mask, cut by the cgroup CPU quota when one is set.  torch sizes
    its pool from the cores it SEES; on a box that grants 16 of 128 an all-cores OpenMP team spins on itself."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_baseline(budget_s=12.0):
    """Oracle (port of the reference's torch golden) on the host cores: 4 sequences of the same shape."""
    import mojo_opset_amd as mo
    import oracle  # noqa: F401

    torch.set_num_threads(usable_cores())

    sample_b = 4
    (q, k, v, lens, table), = build_decode_inputs("cpu", batch=sample_b, sets=1)
    ref = mo.MojoPagedDecodeGQA.get_backend_impl("torch", strict=True)()
    ref(q, k, v, lens, table)  # warm
    n, t0 = 0, time.perf_counter()
    while True:
        ref(q, k, v, lens, table)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 50:
            break
    return {"value": sample_b * n / el, "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} passes of B={sample_b} sequences at ctx={CTX} (same head/page shape), oracle.TorchPagedDecodeGQA"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle-ms", type=float, default=300.0, help="untimed device settle time before the warmup steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--extras-deadline", type=float, default=600.0, help="seconds before the extras are abandoned")
    ns = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if ns.gpus != world:
        sys.exit(f"bench.py: --gpus {ns.gpus} but the launcher started WORLD_SIZE={world} ranks")

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1
    n_dev = max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank % n_dev)
    device = torch.device("cuda", local_rank % n_dev)
    if dist_on:
        import torch.distributed as dist

        # "nccl" is RCCL on ROCm.  MOJO_BENCH_DIST_BACKEND=gloo exists only to dry-run the multi-process control
        # flow on a single-GPU box (several ranks on one device, which RCCL refuses).
        backend = os.environ.get("MOJO_BENCH_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    import mojo_opset_amd as mo

    op = mo.MojoPagedDecodeGQA.get_backend_impl("hip", strict=True)(is_causal=True, gqa_layout="AABB")
    sets = build_decode_inputs(device)
    scale = 1.0 / math.sqrt(D)

    def step(i):
        q, k, v, lens, table = sets[i % len(sets)]
        return op(q, k, v, lens, table, softmax_scale=scale, max_total_seq_len=CTX)

    # Power-state settle (untimed, not counted as warmup): the first ~50 launches after an idle period run through a
    # clock / power transient (kernel-trace durations 176 -> 234 -> 195 us, profiles/r1_decode_kernel_stats.csv); a
    # short run would otherwise time the transient instead of the kernel.
    t_settle = time.perf_counter()
    i_settle = 0
    while (time.perf_counter() - t_settle) * 1e3 < ns.settle_ms:
        for _ in range(16):
            step(i_settle)
            i_settle += 1
        torch.cuda.synchronize()
    for i in range(ns.warmup):
        step(i)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(ns.steps):
        step(i)
    ev1.record()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if dist_on:
        t = torch.tensor([wall], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    lens_cpu = sets[0][3].cpu()
    alg_bytes = decode_algorithmic_bytes(lens_cpu, sets[0][4].shape[1])
    kernel_s = dev_ms / 1e3 / ns.steps
    achieved = alg_bytes / kernel_s / 1e9
    # HBM bytes per launch from the PMC counters cannot be collected inside this process (rocprofv3 owns the counters and
    # must wrap the program): they come from the committed summary of scripts/profile_r2.sh + scripts/summarize_r2.py
    # (separate --pmc FETCH_SIZE / WRITE_SIZE passes over THIS command line; date and box are in the file)
    traffic, traffic_src = None, None
    tp = os.path.join(ROOT, "profiles", "decode_gqa_traffic.json")
    if os.path.exists(tp):
        try:
            tj = json.load(open(tp))
            traffic = tj.get("hbm_bytes_per_launch")
            traffic_src = f"profiles/decode_gqa_traffic.json ({tj.get('collected', 'round 1')})"
        except Exception:
            traffic = None

    line = {
        "metric": "MojoPagedDecodeGQA tokens/s (Llama-3-8B shape, bf16, 1/8 MI355X)",
        "value": world * B * ns.steps / wall,
        "unit": "tokens/s",
        "n_gpus": world,
        "steps": ns.steps,
        "warmup": ns.warmup,
        "ms_per_step": wall * 1e3 / ns.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic",
        "config": {"workload": "MojoPagedDecodeGQA bf16, B=64, 32q/8kv, head_dim=128, page=16, ctx=4096 uniform, "
                               "AABB, random block tables (BASELINE configs[1])",
                   "per_gpu_batch": B, "parallelism": f"replicas x{world} (no collective on this path)"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": alg_bytes, "device_us_per_launch": kernel_s * 1e6,
                     "kernel": ("mojo::decode_split_kernel<bf16,4,nt,paired> (one launch per op call: 8-wave workgroups own a length-ranked pair "
                                "of sequences, their chunk partials are merged in LDS)"
                                if os.environ.get("MOJO_HIP_DECODE_FUSE", "1") != "0" and os.environ.get("MOJO_HIP_DECODE_PAIR", "1") != "0" else
                                "mojo::decode_split_kernel<bf16,4,nt,fused> (MOJO_HIP_DECODE_PAIR=0)"
                                if os.environ.get("MOJO_HIP_DECODE_FUSE", "1") != "0" else
                                "mojo::decode_split_kernel<bf16,4,nt> + mojo::decode_merge_kernel (MOJO_HIP_DECODE_FUSE=0)")},
    }
    hung = False
    if not ns.no_extras:            # every rank takes part: the GEMM + collective cases contain collectives
        # Extras run under a deadline in a worker thread: a collective that one rank never enters (an exception on a
        # peer, a fabric problem) must not take the headline line down with it.
        box = {}

        def _extras():
            try:
                torch.cuda.set_device(device)
                from benchmarks.extras import run_extras

                box["extras"] = run_extras(device, world, rank)
            except Exception as e:      # extras must never break the headline line
                box["extras"] = {"error": repr(e)}

        worker = threading.Thread(target=_extras, daemon=True)
        worker.start()
        worker.join(ns.extras_deadline)
        hung = worker.is_alive()
        line["extras"] = {"error": f"extras did not finish within {ns.extras_deadline:.0f} s"} if hung else box.get("extras")
        # second headline (BASELINE metric: "MojoGroupGemm TFLOPs vs roofline"): its own roofline block on the line
        gg = (line["extras"] or {}).get("MojoGroupGemm_bf16", {}) if isinstance(line["extras"], dict) else {}
        head = gg.get("mixtral_up_16384x4096x28672_G8_KN") if isinstance(gg, dict) else None
        if isinstance(head, dict) and "tflops" in head:
            line["roofline_group_gemm"] = {
                "bound": "mfma", "achieved": head["tflops"], "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": head["tflops"] / MFMA_BF16_PEAK_TFLOPS,
                "traffic": (lambda r, w: None if r is None or w is None else int(r + w))(
                    _profiled("r2_group_gemm_traffic.json", "memory_side_read_bytes"),
                    _profiled("r2_group_gemm_traffic.json", "memory_side_write_bytes")),
                "traffic_source": "profiles/r2_group_gemm_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)",
                "algorithmic_bytes_per_launch": 2 * (16384 * 4096 + 8 * 4096 * 28672 + 16384 * 28672),
                "flops_per_launch": 2.0 * 16384 * 4096 * 28672, "device_us_per_launch": head["us"],
                "sustained_clock_mhz": _profiled("r2_group_gemm_counters.json", "sustained_clock_mhz"),
                "mfma_busy_frac_profiled": _profiled("r2_group_gemm_counters.json", "mfma_busy_frac"),
                "workload": "MojoGroupGemm bf16, Mixtral up-projection: 16384 rows over 8 experts (balanced), K=4096, N=28672, "
                            "weights [G,K,N], random data (BASELINE configs[2])",
                "kernel": "mojo::g256::gemm256_kernel<bf16>"}
    if rank == 0 and world == 1 and not ns.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline()
    if rank == 0:                   # the line goes out before any further collective can get in its way
        print(json.dumps(line), flush=True)
    if dist_on and not hung:
        closer = threading.Thread(target=lambda: (dist.barrier(), dist.destroy_process_group()), daemon=True)
        closer.start()
        closer.join(60.0)
        hung = closer.is_alive()
    if hung:                        # a stuck collective would also block interpreter shutdown: leave, and say so
        print(f"bench.py: rank {rank} abandoned a hung collective / extras thread", file=sys.stderr, flush=True)
        sys.stdout.flush()
        os._exit(3)


if __name__ == "__main__":
    main()
