"""Turn the rocprofv3 CSVs of scripts/profile_r2.sh (merged into gpurun_out/prof_r2) into the committed summaries under
profiles/: decode HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 correction), GroupGemm counters incl. the
sustained clock, and one row per (case, kernel) with the mean duration of the TIMED launches only."""
import csv
import datetime
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_r2")
DST = os.path.join(ROOT, "profiles")
os.makedirs(DST, exist_ok=True)


def one(pattern):
    hits = glob.glob(os.path.join(SRC, pattern), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None


def rows_of(path):
    return list(csv.DictReader(open(path))) if path else []


def counter_mean(path, kernel_substr, counter, last=None):
    vals = [float(r["Counter_Value"]) for r in rows_of(path) if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == counter]
    if last:
        vals = vals[-last:]
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def kernel_durations(path):
    """{kernel name: [durations ns in start order]}"""
    out = {}
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in rows_of(path)))
    for _, d, k in rows:
        out.setdefault(k, []).append(d)
    return out


stamp = {"collected": datetime.date.today().isoformat(), "tool": "rocprofv3 (ROCm 7.2), one MI355X box via gpurun",
         "script": "scripts/profile_r2.sh + scripts/summarize_r2.py"}

# ---- decode traffic --------------------------------------------------------------------------------------------------
TIMED = 200
tr = one("decode_stats/**/*kernel_trace.csv")
if tr:
    shutil.copy(one("decode_stats/**/*kernel_stats.csv"), os.path.join(DST, "r2_decode_kernel_stats.csv"))
    d = kernel_durations(tr)
    name = next(k for k in d if "decode_split_kernel" in k)
    timed = d[name][-TIMED:]
    fetch, nf = counter_mean(one("decode_fetch/**/*counter_collection.csv"), "decode_split_kernel", "FETCH_SIZE", TIMED)
    write, nw = counter_mean(one("decode_write/**/*counter_collection.csv"), "decode_split_kernel", "WRITE_SIZE", TIMED)
    out = dict(stamp, kernel=name, launches=len(timed), avg_duration_us=sum(timed) / len(timed) / 1e3,
               note="mean over the timed steps (the last 200 launches of `bench.py --steps 200 --warmup 20`)",
               all_launches=len(d[name]), avg_duration_us_all_launches=sum(d[name]) / len(d[name]) / 1e3,
               FETCH_SIZE_KiB_raw=fetch, WRITE_SIZE_KiB_raw=write,
               fetch_correction="x2 (gfx950: FETCH_SIZE tallies the 128-B requests of 16 B/lane streaming reads at 64 B)",
               algorithmic_bytes_per_launch=1074856192)
    if fetch is not None and write is not None:
        out["hbm_bytes_per_launch"] = int(2 * fetch * 1024 + write * 1024)
        out["traffic_over_algorithmic"] = out["hbm_bytes_per_launch"] / out["algorithmic_bytes_per_launch"]
    json.dump(out, open(os.path.join(DST, "decode_gqa_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))

# ---- group gemm counters ----------------------------------------------------------------------------------------------
pm = one("gg_pmc/**/*counter_collection.csv")
tr = one("gg_stats/**/*kernel_trace.csv")
if pm and tr:
    shutil.copy(one("gg_stats/**/*kernel_stats.csv"), os.path.join(DST, "r2_gg_kernel_stats.csv"))
    d = kernel_durations(tr)
    name = next(k for k in d if "gemm256_kernel" in k)
    timed = d[name][2:]                                   # gemm_bench.py: 2 warm-up launches, then the timed ones
    agg = {c: counter_mean(pm, "gemm256_kernel", c)[0] for c in
           ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE")}
    pd = kernel_durations(one("gg_pmc/**/*kernel_trace.csv"))
    pname = next(k for k in pd if "gemm256_kernel" in k)
    dur_pmc = sum(pd[pname]) / len(pd[pname])            # ns, in the counter pass itself (profiled passes clock lower)
    g = dict(stamp, kernel=name, shape="M=16384 K=4096 N=28672 G=8 (Mixtral up-projection), random bf16 data",
             launches=len(timed), avg_duration_us=sum(timed) / len(timed) / 1e3, counters_mean_per_launch=agg,
             avg_duration_us_in_counter_pass=dur_pmc / 1e3)
    flops = 2.0 * 16384 * 4096 * 28672
    g["tflops"] = flops / (g["avg_duration_us"] * 1e-6) / 1e12
    if agg.get("GRBM_GUI_ACTIVE"):
        g["sustained_clock_mhz"] = agg["GRBM_GUI_ACTIVE"] / 8.0 / (dur_pmc * 1e-9) / 1e6       # sum over 8 XCDs / wall
        # SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD summed over the chip: 1024 SIMDs
        g["mfma_busy_frac"] = agg["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * agg["GRBM_GUI_ACTIVE"] / 8.0)
        g["tflops_at_that_clock_if_mfma_never_idle"] = 2.0 * 512 * 1024 * g["sustained_clock_mhz"] * 1e6 / 1e12
    json.dump(g, open(os.path.join(DST, "r2_group_gemm_counters.json"), "w"), indent=1)
    print(json.dumps(g, indent=1))

# ---- one row per (case, kernel) ------------------------------------------------------------------------------------------
# benchmarks/extras._time(fn, iters, warmup): the LAST `iters` launches of a kernel in a one-case process are the timed ones;
# ops that launch several kernels per call repeat each of them once per call, so "timed" = the last (n * iters / calls) of each.
rows = []
for path in sorted(glob.glob(os.path.join(SRC, "case_*"))):
    if not os.path.isdir(path):
        continue
    tag = os.path.basename(path)[5:]
    tr = max(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime, default=None)
    if not tr:
        continue
    d = kernel_durations(tr)
    for k, v in d.items():
        if not k.startswith("mojo::") and "mojo" not in k:
            continue
        # benchmarks/extras now replays every case for 30 ms before it times (the post-idle power transient,
        # profiles/r2_decode_sustain.txt), so a trace is: warm-up, settle launches, timed launches.  The LAST THIRD of a
        # kernel's launches is past the transient in every case (settle >= 2 x timed) and contains all timed launches.
        timed = v[-max(1, len(v) // 3):] if len(v) >= 6 else v
        rows.append({"case": tag, "kernel": k[:160], "launches": len(v), "timed_launches": len(timed),
                     "avg_us_timed": round(sum(timed) / len(timed) / 1e3, 2), "min_us": round(min(v) / 1e3, 2)})
if rows:
    with open(os.path.join(DST, "r2_kernel_cases.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)
    for r in rows:
        print(r)
