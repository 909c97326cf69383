"""bench.py — headline benchmark: MojoPagedDecodeGQA, Llama-3-8B shape, on MI355X.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A *step* is one pass of the hot path over one batch: paged decode attention for B=64 sequences
(32 q heads / 8 kv heads, head_dim 128, page 16, bf16, ctx 4096 each, block tables a random
permutation) with all inputs resident in HBM.  ``value`` = tokens/s = B * steps / time, whole job.
With N > 1 every rank runs its own batch (the op has no exchange step: replicas, weak scaling).

Output (benchmarks/result_line.py):
  LAST stdout line — compact JSON (<= 4 KB, asserted): the contract fields + ``roofline`` (HBM roofline of the
                 decode op from HIP-event time of the timed region), ``cpu_baseline`` (the torch-native oracle, a port
                 of the reference's golden backend, timed on the host cores on a bounded sample of the same
                 workload) and a six-key ``roofline_group_gemm``.
  an EARLIER line ``EXTRAS {...}`` and ``bench_extras.json`` — everything else measured in the same run: the other
                 hot-path ops (absolute rate + roofline fraction), per-op CPU baselines, GEMM + collective cases.
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


def _headline_kernel():
    """The kernel form the headline op call launched, as the LIBRARY reports it (`mojo_hip_last_launch`, called right after
    the timed region) — not a guess from the environment: the switches are latched inside the library."""
    from mojo_opset_amd.backends.hip import lib

    form = lib.last_launch()                      # e.g. "decode_mfma:paired:nt"
    kernel, _, rest = form.partition(":")
    names = {"decode_mfma": "mojo::decode_mfma_kernel<bf16, head_dim 128>", "decode_valu": "mojo::decode_split_kernel<bf16, 4 heads>"}
    shape = {"paired": "one launch per op call: 8-wave workgroups own a length-ranked pair of sequences, their chunk partials are merged in LDS",
             "fused": "one launch per op call: a (sequence, kv-head)'s chunks are the waves of one workgroup, merged in LDS",
             "grouped+merge": "eight-wave workgroups leave one partial each + mojo::decode_merge_kernel",
             "split+merge": "one partial per chunk + mojo::decode_merge_kernel"}
    what = rest.split(":")[0]
    return f"{names.get(kernel, kernel)} [{form}: {shape.get(what, what)}]"


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


def _cpu_loop(fn, budget_s, max_n=50):
    fn()  # warm (thread pool, allocator)
    n, t0 = 0, time.perf_counter()
    while True:
        fn()
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= max_n:
            return n, el / n


def cpu_baselines_per_op(budget_s=2.0):
    """The oracle (a port of the reference's MOJO_BACKEND=torch path) timed on this node's host cores for every extras group,
    on bounded samples (north_star: "next to the MOJO_BACKEND=torch CPU path timed on the node's own host cores in the same
    run").  configs[0] — ResidualAddRMSNorm + SwiGLU, fp32 [2048, 4096] — runs at full size; the others at stated reduced
    sizes.  Reported baselines, not targets."""
    import mojo_opset_amd as mo
    import oracle  # noqa: F401

    cores = usable_cores()
    torch.set_num_threads(cores)
    ref = lambda name: getattr(mo, name).get_backend_impl("torch", strict=True)  # noqa: E731
    out = {}
    g = torch.Generator().manual_seed(20260716)

    def rec(key, fn, unit, per_call, sample):
        try:
            n, sec = _cpu_loop(fn, budget_s)
            out[key] = {"value": per_call / sec, "unit": unit, "us_per_call": sec * 1e6, "cores": cores, "kind": "port",
                        "sample": f"{n} passes; {sample}"}
        except Exception as e:
            out[key] = {"error": repr(e)}

    # configs[0] at full size
    rows, d = 2048, 4096
    x, r = torch.randn(rows, d, generator=g), torch.randn(rows, d, generator=g)
    norm = ref("MojoResidualAddRMSNorm")(d, 1e-5, "pre", dtype=torch.float32)
    with torch.no_grad():
        norm.weight.copy_(torch.randn(d, generator=g))
    rec("streaming_ops/residual_add_rmsnorm_fp32_2048x4096", lambda: norm(x, r), "GB/s", (4 * rows * d * 4 + d * 4) / 1e9,
        "BASELINE configs[0] at full size: oracle TorchResidualAddRMSNorm fp32 [2048, 4096]")
    act = ref("MojoSwiGLU")()
    rec("streaming_ops/swiglu_fp32_2048x4096", lambda: act(x, r), "GB/s", 3 * rows * d * 4 / 1e9,
        "BASELINE configs[0] at full size: oracle TorchSwiGLU fp32 [2048, 4096]")
    # GroupGemm, 1/8 of the Mixtral rows and 1/7 of its N
    m, k, n, groups = 2048, 4096, 4096, 8
    xg = torch.randn(m, k, generator=g).to(torch.bfloat16)
    wg = torch.randn(groups, k, n, generator=g).to(torch.bfloat16)
    counts = torch.full((groups,), m // groups, dtype=torch.int32)
    gg = ref("MojoGroupGemm")(wg, False)
    rec("MojoGroupGemm_bf16", lambda: gg(xg, counts), "TFLOP/s", 2.0 * m * k * n / 1e12,
        f"oracle TorchGroupGemm bf16, {m} rows over {groups} experts, K={k}, N={n} (bench case: 16384 x 4096 x 28672)")
    del xg, wg
    # QuantGemm int8 at the decode-sized DeepSeek shape
    m, k, n = 16, 7168, 4096
    qg = ref("MojoQuantGemm")(k, n, trans_weight=True)
    with torch.no_grad():
        qg.weight.copy_(torch.randint(-127, 128, (n, k), dtype=torch.int8, generator=g))
        qg.weight_scale.fill_(0.01)
    xq, sq = torch.randint(-127, 128, (m, k), dtype=torch.int8, generator=g), torch.rand(m, generator=g)
    rec("MojoQuantGemm", lambda: qg(xq, sq), "TOP/s", 2.0 * m * k * n / 1e12,
        f"oracle TorchQuantGemm int8 {m} x {k} x {n} (bench cases: M up to 4096)")
    # paged prefill GQA: one sequence of 512 tokens, Llama-3-8B heads
    hq, hkv, dd, page, t = 32, 8, 128, 16, 512
    pages = t // page
    kc = torch.randn(pages + 2, hkv, page, dd, generator=g).to(torch.bfloat16)
    vc = torch.randn(pages + 2, hkv, page, dd, generator=g).to(torch.bfloat16)
    tb = torch.randperm(pages, generator=g, dtype=torch.int32).view(1, pages)
    qp = torch.randn(t, hq, dd, generator=g).to(torch.bfloat16)
    cu = torch.tensor([0, t], dtype=torch.int32)
    pf = ref("MojoPagedPrefillGQA")()
    rec("MojoPagedPrefillGQA_bf16", lambda: pf(qp, kc, vc, cu, tb), "TFLOP/s", 4.0 * hq * dd * (t * t / 2.0) / 1e12,
        f"oracle TorchPagedPrefillGQA bf16, 1 x {t} tokens, no cache (bench cases: 4 x 2048 ... 16384)")
    # paged MLA decode / prefill at DeepSeek-V3 dims, 2 sequences of 1024 cached tokens / one sequence of 128 new tokens
    h, nope, rope, vd, rr, ctx = 128, 128, 64, 128, 512, 1024
    b = 2
    pages = ctx // page
    ckv = torch.randn(b * pages + 2, 1, page, rr, generator=g).to(torch.bfloat16)
    kpe = torch.randn(b * pages + 2, 1, page, rope, generator=g).to(torch.bfloat16)
    tbm = torch.randperm(b * pages, generator=g, dtype=torch.int32).view(b, pages)
    md = ref("MojoPagedDecodeMLA")(h, nope, rope, vd, rr).to(torch.bfloat16)
    with torch.no_grad():
        md.kv_b_proj.copy_(torch.randn(md.kv_b_proj.shape, generator=g) * 0.02)
    qm = torch.randn(b, h, nope + rope, generator=g).to(torch.bfloat16)
    lens = torch.full((b,), ctx, dtype=torch.int32)
    rec("MojoPagedDecodeMLA_bf16", lambda: md(qm, ckv, kpe, lens, tbm), "tokens/s", float(b),
        f"oracle TorchPagedDecodeMLA bf16, B={b}, H=128, ctx={ctx} (bench case: B=64, ctx=4096)")
    tq = 128
    mp = ref("MojoPagedPrefillMLA")(h, nope, rope, vd, rr).to(torch.bfloat16)
    with torch.no_grad():
        mp.kv_b_proj.copy_(md.kv_b_proj)
    qpm = torch.randn(tq, h, nope + rope, generator=g).to(torch.bfloat16)
    cum = torch.tensor([0, tq], dtype=torch.int32)
    rec("MojoPagedPrefillMLA_bf16", lambda: mp(qpm, ckv, kpe, cum, tbm[:1]), "TFLOP/s",
        (2.0 * tq * rr * h * (nope + vd) + 2.0 * h * (tq * tq / 2.0) * (nope + rope + vd)) / 1e12,
        f"oracle TorchPagedPrefillMLA bf16, 1 x {tq} tokens (bench cases: 4 x 512 with up to 2048 cached)")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle-ms", type=float, default=300.0, help="untimed device settle time before the warmup steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--extras-deadline", type=float, default=900.0, help="seconds before the extras are abandoned")
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
    # must wrap the program): they come from the committed summary of scripts/profile_r3.sh + scripts/summarize_r3.py
    # (separate --pmc FETCH_SIZE / WRITE_SIZE passes over THIS command line; date and box are in the file)
    traffic, traffic_src = None, None
    tp = os.path.join(ROOT, "profiles", "decode_gqa_traffic.json")
    if os.path.exists(tp):
        try:
            tj = json.load(open(tp))
            traffic = tj.get("hbm_bytes_per_launch")
            traffic_src = f"profiled: profiles/decode_gqa_traffic.json ({tj.get('collected', 'round 1')})"
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
                     "kernel": _headline_kernel()},
    }
    if dist_on:                     # what the collective library itself reports: a SCALE line must describe its own fabric
        try:
            names = [None] * world
            dist.all_gather_object(names, f"{torch.cuda.get_device_name(device)} (device {device.index})")
            line["comm"] = {"backend": dist.get_backend(), "world_observed": dist.get_world_size(), "ranks": names,
                            "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if dist.get_backend() == "nccl" else None,
                            "xgmi_link_peak_GBps": 153.0,
                            "per_case": "extras.compute_comm_bf16: payload_MB_per_rank, link_MB, link_GBps_over_exposed_time, "
                                        "bare_collective_us, overlap_frac (1 - exposed exchange / bare collective), speedup_vs_tp1"}
        except Exception as e:
            line["comm"] = {"error": repr(e)}
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
                "traffic": (lambda o: None if not isinstance(o, dict) else int(
                    o.get("read_bytes_beyond_L2 (FETCH_SIZE KiB x 1024 x 2)", 0) + o.get("write_bytes", 0)))(
                    _profiled("r5_group_gemm_counters.json", "orders_measured_0")),
                "traffic_source": "profiled: profiles/r5_group_gemm_counters.json orders_measured_0 (round-5 binary; rocprofv3 --pmc "
                                  "FETCH_SIZE x2 + WRITE_SIZE, separate passes: scripts/profile_r5_gemm.sh)",
                "algorithmic_bytes_per_launch": 2 * (16384 * 4096 + 8 * 4096 * 28672 + 16384 * 28672),
                "flops_per_launch": 2.0 * 16384 * 4096 * 28672, "device_us_per_launch": head["us"],
                "sustained_clock_mhz": (_profiled("r5_group_gemm_counters.json", "orders_measured_0") or {}).get("sustained_clock_mhz"),
                "mfma_busy_frac_profiled": (_profiled("r5_group_gemm_counters.json", "orders_measured_0") or {}).get("mfma_busy_frac"),
                "calibration_this_run": gg.get("mixtral_up_16384x4096x28672_G8_KN_calibration"),
                "bare_mfma_loop_tflops_profiled": "profiles/r5_mfma_shape_power_probe.txt: a loop of NOTHING but 16x16x32 bf16 MFMAs on random "
                                                  "register operands holds 2 060 TFLOP/s (0.82 of nominal) on this chip, 1 820 with its fragments "
                                                  "re-read from LDS (0.73); the 32x32x16 form 1 830 / 1 630",
                "workload": "MojoGroupGemm bf16, Mixtral up-projection: 16384 rows over 8 experts (balanced), K=4096, N=28672, "
                            "weights [G,K,N], random data (BASELINE configs[2])",
                "kernel": "mojo::g256::gemm256_kernel<bf16>"}
    # roofline blocks of the other BASELINE configs measured in the extras (same layout as `roofline`; `traffic` from the
    # committed PMC summary profiles/r3_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 x2
    # correction on the read side — scripts/profile_r3_traffic.sh)
    ex = line.get("extras") if isinstance(line.get("extras"), dict) else {}

    def _case(group, case):
        rec = ex.get(group, {})
        rec = rec.get(case) if isinstance(rec, dict) else None
        return rec if isinstance(rec, dict) and "us" in rec else None

    def _traffic(tag):
        t = (_profiled("r5_traffic.json", tag) or _profiled("r4_traffic.json", tag) or _profiled("r3_traffic.json", tag))
        if not isinstance(t, dict):
            return None, None
        return t.get("hbm_bytes_per_op"), f"profiled: profiles/r5_traffic.json (r4 / r3 file when absent):{tag} ({t.get('collected', '')}; kernels: {', '.join(t.get('kernels', []))})"

    rec = _case("MojoPagedDecodeMLA_bf16", "B64_H128_ctx4096_page16")
    if rec:
        alg = 64 * 4096 * (512 + 64) * 2 + 128 * 256 * 512 * 2 + 2 * 64 * 128 * (192 + 128) * 2
        tr, src = _traffic("mla_decode")
        line["roofline_mla_decode"] = {
            "bound": "hbm", "achieved": rec["GB/s"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rec["GB/s"] / HBM_PEAK_GBS,
            "traffic": tr, "traffic_source": src, "algorithmic_bytes_per_launch": alg, "device_us_per_launch": rec["us"],
            "us_min": rec.get("us_min"), "us_max": rec.get("us_max"), "tflops": rec.get("tflops"),
            "workload": "MojoPagedDecodeMLA bf16, DeepSeek-V3 dims (H=128, nope 128 / rope 64 / v 128, r 512), B=64, ctx=4096, page 16 "
                        "(BASELINE configs[4]); one op call = absorb projection + latent attention + split merge + output projection",
            "kernel": "mojo::mla512_ps_kernel<bf16> (+ gemm_skinny_kernel x2, mla_merge_kernel)"}
    rec = _case("MojoPagedPrefillGQA_bf16", "4x2048_nocache")
    if rec:
        tr, src = _traffic("prefill_gqa_4x2048")
        line["roofline_prefill_gqa"] = {
            "bound": "mfma", "achieved": rec["tflops"], "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": rec["tflops"] / MFMA_BF16_PEAK_TFLOPS, "traffic": tr, "traffic_source": src,
            "algorithmic_bytes_per_launch": 2 * 4 * 2048 * 32 * 128 * 2 + 2 * 4 * 2048 * 8 * 128 * 2,
            "flops_per_launch": 4 * 4.0 * 32 * 128 * (2048 * 2048 / 2.0), "device_us_per_launch": rec["us"],
            "us_min": rec.get("us_min"), "us_max": rec.get("us_max"),
            "workload": "MojoPagedPrefillGQA bf16, 4 sequences x 2048 new tokens, 32q/8kv heads, head_dim 128, page 16, causal "
                        "(BASELINE configs[2], Mixtral attention shape)",
            "kernel": "mojo::prefill_kernel<bf16>"}
    rec = _case("MojoQuantGemm", "fp8_e4m3_4096x7168x36864_NK")
    if rec:
        tr, src = _traffic("quant_gemm_fp8_4096x7168x36864")
        q5 = _profiled("r5_quant_gemm_counters.json", "fp8")
        if isinstance(q5, dict) and q5.get("hbm_bytes_per_launch"):
            tr, src = int(q5["hbm_bytes_per_launch"]), "profiled: profiles/r5_quant_gemm_counters.json fp8 (round-5 binary; scripts/profile_r5_gemm.sh)"
        line["roofline_quant_gemm"] = {
            "bound": "mfma", "achieved": rec["tflops"], "peak": 5000.0, "unit": "TFLOP/s", "frac": rec["tflops"] / 5000.0,
            "traffic": tr, "traffic_source": src, "algorithmic_bytes_per_launch": 4096 * 7168 + 7168 * 36864 + 4096 * 36864 * 2,
            "flops_per_launch": 2.0 * 4096 * 7168 * 36864, "device_us_per_launch": rec["us"],
            "us_min": rec.get("us_min"), "us_max": rec.get("us_max"),
            "workload": "MojoQuantGemm fp8 e4m3 (v_mfma_f32_16x16x128_f8f6f4), DeepSeek-V3 shape M=4096, K=7168, N=36864, weight [N,K] "
                        "(BASELINE configs[4]; fp8 parity is unpinned: the reference implements int8 only)",
            "kernel": "mojo::g256::gemm256_kernel<PolF8, EpilogueDequant>",
            "sustained_clock_mhz": (q5 or {}).get("sustained_clock_mhz") if isinstance(q5, dict) else None,
            "mfma_busy_frac_profiled": (q5 or {}).get("mfma_busy_frac") if isinstance(q5, dict) else None,
            "lds_active_frac_of_cu_cycles_profiled": (q5 or {}).get("lds_active_frac_of_cu_cycles") if isinstance(q5, dict) else None,
            "zero_operand_tflops_profiled": ((q5 or {}).get("wall_same_process") or {}).get("zeros_tflops", {}).get("median") if isinstance(q5, dict) else None}
    if isinstance(line.get("roofline_group_gemm"), dict):
        gg = line["roofline_group_gemm"]
        clk = gg.get("sustained_clock_mhz")
        gg.update({"target": 0.80, "ceiling_at_sustained_clock": None if not clk else clk / 2400.0,
                   "note": "target 0.80 of the nominal 2.5 PF is formally missed: at the clock the chip holds under this load "
                           "a 100 % busy matrix pipe would deliver `ceiling_at_sustained_clock` of the nominal peak"})
    try:
        from mojo_opset_amd import switches
        line["switches_set"] = {k: v for k, v in switches.in_effect().items() if v is not None}     # which build / route produced the numbers
        from mojo_opset_amd.backends.hip import lib as _lib
        line["library"] = _lib.load().mojo_hip_version().decode()
    except Exception as e:
        line["switches_set"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not ns.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline()
        if not ns.no_extras:
            line["cpu_baseline_per_op"] = cpu_baselines_per_op()
    if rank == 0:                   # the line goes out before any further collective can get in its way
        from benchmarks.result_line import emit

        # everything measured -> bench_extras.json + an `EXTRAS ` stdout line; the LAST stdout line is the compact
        # (<= 4 KB) contract line the driver parses
        emit(line)
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
