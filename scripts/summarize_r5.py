"""Turn the rocprofv3 CSVs of scripts/profile_r5.sh (gpurun_out/prof_r5) into the committed summaries under profiles/:
decode_gqa_traffic.json (headline kernel: duration of the timed launches + HBM bytes per launch), r5_traffic.json (HBM bytes
per op call of the MLA decode / prefill GQA / fp8 QuantGemm bench cases: FETCH_SIZE x 2 + WRITE_SIZE, separate passes) and
r5_kernel_cases.csv (one row per case and kernel: launches, mean / min / max duration)."""
import csv
import datetime
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_r5")
DST = os.path.join(ROOT, "profiles")


def one(pattern):
    hits = glob.glob(os.path.join(SRC, pattern), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None


def rows_of(path):
    return list(csv.DictReader(open(path))) if path else []


def durations(path):
    out = {}
    for _, d, k in sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in rows_of(path)):
        out.setdefault(k, []).append(d)
    return out


def counters(path, counter):
    out = {}
    for r in rows_of(path):
        if r["Counter_Name"] == counter:
            out.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return out


def short(k):
    return k.split("(")[0][:110]


stamp = {"collected": datetime.date.today().isoformat(), "tool": "rocprofv3 (ROCm 7.2), one MI355X box via gpurun",
         "script": "scripts/profile_r5.sh + scripts/summarize_r5.py",
         "fetch_correction": "x2 (gfx950: FETCH_SIZE tallies the 128-B requests of 16 B/lane streaming reads at 64 B; unit KiB)"}

# ---- headline: decode GQA ------------------------------------------------------------------------------------------------
TIMED = 200
tr = one("decode_stats/**/*kernel_trace.csv")
if tr:
    st = one("decode_stats/**/*kernel_stats.csv")
    if st:
        shutil.copy(st, os.path.join(DST, "r5_decode_kernel_stats.csv"))
    d = durations(tr)
    name = max((k for k in d if "decode_" in k and "mojo" in k), key=lambda k: sum(d[k]))
    timed = d[name][-TIMED:]
    f = counters(one("decode_fetch/**/*counter_collection.csv"), "FETCH_SIZE").get(name, [])[-TIMED:]
    w = counters(one("decode_write/**/*counter_collection.csv"), "WRITE_SIZE").get(name, [])[-TIMED:]
    out = dict(stamp, kernel=name, launches=len(timed), avg_duration_us=sum(timed) / len(timed) / 1e3,
               min_duration_us=min(timed) / 1e3, max_duration_us=max(timed) / 1e3,
               note="the timed steps = the last 200 launches of `bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline`",
               all_launches=len(d[name]), avg_duration_us_all_launches=sum(d[name]) / len(d[name]) / 1e3,
               FETCH_SIZE_KiB_raw=sum(f) / len(f) if f else None, WRITE_SIZE_KiB_raw=sum(w) / len(w) if w else None,
               algorithmic_bytes_per_launch=1074856192)
    if f and w:
        out["hbm_bytes_per_launch"] = int(2 * out["FETCH_SIZE_KiB_raw"] * 1024 + out["WRITE_SIZE_KiB_raw"] * 1024)
        out["traffic_over_algorithmic"] = out["hbm_bytes_per_launch"] / out["algorithmic_bytes_per_launch"]
    json.dump(out, open(os.path.join(DST, "decode_gqa_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))

# ---- per-op traffic of the other roofline blocks ----------------------------------------------------------------------
traffic = {}
case_rows = []
for tag, main_kernel in (("mla_decode", "mla512"), ("prefill_gqa_4x2048", "prefill_kernel"), ("quant_gemm_fp8_4096x7168x36864", "gemm256_kernel")):
    tr = one(f"{tag}_stats/**/*kernel_trace.csv")
    if not tr:
        continue
    d = {k: v for k, v in durations(tr).items() if "mojo" in k}
    mains = [k for k in d if main_kernel in k]
    if not mains:
        continue
    calls = max(len(d[k]) for k in mains)
    f = counters(one(f"{tag}_fetch/**/*counter_collection.csv"), "FETCH_SIZE")
    w = counters(one(f"{tag}_write/**/*counter_collection.csv"), "WRITE_SIZE")
    rec = dict(stamp, op_calls=calls, kernels=[], per_kernel={})
    total = 0.0
    complete = True
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        per_call = len(v) / calls
        fk, wk = f.get(k), w.get(k)
        pk = {"launches_per_op": per_call, "avg_duration_us": sum(v) / len(v) / 1e3}
        if fk and wk:
            pk["read_bytes"] = 2 * 1024 * sum(fk) / len(fk)
            pk["write_bytes"] = 1024 * sum(wk) / len(wk)
            total += per_call * (pk["read_bytes"] + pk["write_bytes"])
        else:
            complete = False
        rec["kernels"].append(short(k))
        rec["per_kernel"][short(k)] = pk
        case_rows.append((tag, short(k), len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3))
    rec["hbm_bytes_per_op"] = int(total) if complete and total else None
    rec["device_us_per_op"] = sum(sum(v) for v in d.values()) / calls / 1e3
    traffic[tag] = rec
if traffic:
    json.dump(traffic, open(os.path.join(DST, "r5_traffic.json"), "w"), indent=1)
    for tag, rec in traffic.items():
        print(tag, rec["hbm_bytes_per_op"], round(rec["device_us_per_op"], 1), rec["kernels"])

# ---- durations of the other cases ------------------------------------------------------------------------------------
for path in sorted(glob.glob(os.path.join(SRC, "case_*"))):
    if not os.path.isdir(path):
        continue
    tag = os.path.basename(path)[5:]
    tr = one(f"case_{tag}/**/*kernel_trace.csv")
    for k, v in sorted(durations(tr).items(), key=lambda kv: -sum(kv[1])):
        if "mojo" in k:
            case_rows.append((tag, short(k), len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3))
if case_rows:
    with open(os.path.join(DST, "r5_kernel_cases.csv"), "w", newline="") as fh:
        wr = csv.writer(fh)
        wr.writerow(["case", "kernel", "launches", "avg_us", "min_us", "max_us"])
        for r in case_rows:
            wr.writerow([r[0], r[1], r[2], f"{r[3]:.2f}", f"{r[4]:.2f}", f"{r[5]:.2f}"])
    print(f"{len(case_rows)} (case, kernel) rows")
