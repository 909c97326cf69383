"""Turn the rocprofv3 CSVs merged into gpurun_out/prof_r1 into the committed summaries under profiles/."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_r1")
DST = os.path.join(ROOT, "profiles")
os.makedirs(DST, exist_ok=True)


def one(pattern):
    hits = glob.glob(os.path.join(SRC, pattern), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None     # gpurun merges new runs next to old ones


def counter_mean(path, kernel_substr, counter):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def kernel_avg_ns(path, kernel_substr, last=None):
    """Mean duration of the launches of one kernel; ``last`` keeps only the final N launches in start order (the timed
    steps of bench.py: the settle and warmup launches in front of them run through the clock transient)."""
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            for r in csv.DictReader(open(path)) if kernel_substr in r["Kernel_Name"]]
    rows.sort()
    d = [x[1] for x in rows]
    if last is not None:
        d = d[-last:]
    return (sum(d) / len(d), len(d)) if d else (None, 0)


out = {}
for tag in ("decode", "gg", "mla", "prefill"):
    f = one(f"{tag}_stats/**/*kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(DST, f"r1_{tag}_kernel_stats.csv"))
# decode traffic: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts wide coalesced reads at half
# their size (MI355X_MICROARCH.md §HBM) -> doubled.
fetch, nf = counter_mean(one("decode_fetch/**/*counter_collection.csv"), "decode_split_kernel", "FETCH_SIZE")
write, nw = counter_mean(one("decode_write/**/*counter_collection.csv"), "decode_split_kernel", "WRITE_SIZE")
TIMED_STEPS = 200                     # scripts/profile_r1.sh runs bench.py --steps 200
avg_ns, n = kernel_avg_ns(one("decode_stats/**/*kernel_trace.csv"), "decode_split_kernel", last=TIMED_STEPS)
all_ns, n_all = kernel_avg_ns(one("decode_stats/**/*kernel_trace.csv"), "decode_split_kernel")
merge_ns, _ = kernel_avg_ns(one("decode_stats/**/*kernel_trace.csv"), "decode_merge_kernel", last=TIMED_STEPS)
if fetch is not None and write is not None:
    out = {
        "kernel": "mojo::decode_split_kernel<bf16,4,nt,fused>",
        "launches": n,
        "avg_duration_us": avg_ns / 1e3,
        "note": "mean over the timed steps (the last 200 launches); the kernel_stats CSV also averages the settle and "
                "warmup launches, which run through the clock transient",
        "all_launches": n_all,
        "avg_duration_us_all_launches": all_ns / 1e3,
        "merge_kernel_avg_us": merge_ns / 1e3 if merge_ns else None,
        "FETCH_SIZE_KiB_raw": fetch,
        "WRITE_SIZE_KiB_raw": write,
        "fetch_correction": "x2 (gfx950: FETCH_SIZE tallies 128-B requests at 64 B for 16 B/lane streaming reads)",
        "hbm_bytes_per_launch": int(2 * fetch * 1024 + write * 1024),
        "algorithmic_bytes_per_launch": 1074856192,
    }
    out["traffic_over_algorithmic"] = out["hbm_bytes_per_launch"] / out["algorithmic_bytes_per_launch"]
    json.dump(out, open(os.path.join(DST, "decode_gqa_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))
# group gemm counters
pm = one("gg_pmc/**/*counter_collection.csv")
if pm:
    agg = {}
    for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE"):
        agg[c], _ = counter_mean(pm, "gemm256_kernel", c)
    ns, n = kernel_avg_ns(one("gg_stats/**/*kernel_trace.csv"), "gemm256_kernel")
    ff, _ = counter_mean(one("gg_fetch/**/*counter_collection.csv"), "gemm256_kernel", "FETCH_SIZE") if one("gg_fetch/**/*counter_collection.csv") else (None, 0)
    g = {"kernel": "mojo::g256::gemm256_kernel<PolBF16, W=[K,N]>", "shape": "M=16384 K=4096 N=28672 G=8",
         "avg_duration_us": ns / 1e3 if ns else None, "launches": n, "counters_mean_per_launch": agg,
         "FETCH_SIZE_KiB_raw": ff,
         "tflops": 2.0 * 16384 * 4096 * 28672 / (ns * 1e-9) / 1e12 if ns else None}
    if agg.get("SQ_VALU_MFMA_BUSY_CYCLES") and agg.get("GRBM_GUI_ACTIVE"):
        # MFMA busy cycles summed over 256 CUs x 4 SIMDs; GRBM_GUI_ACTIVE summed over 8 XCDs
        g["mfma_busy_frac_est"] = agg["SQ_VALU_MFMA_BUSY_CYCLES"] / (agg["GRBM_GUI_ACTIVE"] / 8 * 256 * 4)
    json.dump(g, open(os.path.join(DST, "r1_group_gemm_counters.json"), "w"), indent=1)
    print(json.dumps(g, indent=1))
