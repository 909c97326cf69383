"""gpurun_out/prof_r5_tile128 -> r5_gemm_tile128_counters.json (scripts/profile_r5_tile128.sh).

    python3 scripts/summarize_r5_tile128.py [outdir]      # outdir defaults to profiles/
"""
import csv, datetime, glob, json, sys

P = "gpurun_out/prof_r5_tile128"
OUT = sys.argv[1] if len(sys.argv) > 1 else "profiles"
CASES = {   # description, (m, k, n), operand bytes per element, MFMA peak (TFLOP/s)
    "d1024": ("dense bf16 1024 x 4096 x 4096, [N,K], one weight (cache-resident)", (1024, 4096, 4096), 2, 2500.0),
    "d1024c": ("dense bf16 1024 x 4096 x 4096, [N,K], every launch another copy of the weight (cold)", (1024, 4096, 4096), 2, 2500.0),
    "d2048w": ("dense bf16 2048 x 4096 x 4096, [N,K]: 128 x 256 eight-wave tiles", (2048, 4096, 4096), 2, 2500.0),
    "d256sk": ("dense bf16 256 x 8192 x 1024, [N,K]: the tiles' own K split + finalize", (256, 8192, 1024), 2, 2500.0),
    "q1024": ("MojoQuantGemm int8 1024 x 4096 x 4096, [N,K]", (1024, 4096, 4096), 1, 5000.0),
}
DROP = 3      # launches dropped at the head of every pass (warm-up, clock transient)


def _rows(pat, suffix):
    f = glob.glob(f"{P}/{pat}/**/*{suffix}", recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []


def _is_main(name):
    return "gemm128_kernel" in name


def mean(pat, counter):
    v = [float(r["Counter_Value"]) for r in _rows(pat, "counter_collection.csv") if _is_main(r["Kernel_Name"]) and r["Counter_Name"] == counter]
    return sum(v[DROP:]) / max(len(v[DROP:]), 1) if v else None


def durs(pat):
    out = {}
    for r in _rows(pat, "kernel_trace.csv"):
        out.setdefault(r["Kernel_Name"], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return {k: sum(v[DROP:]) / max(len(v[DROP:]), 1) / 1e3 for k, v in out.items() if "mojo" in k}


try:
    wall = json.loads(open(f"{P}/wall.log").read().strip().split("\n")[-1])
except Exception as e:
    wall = {"error": repr(e)}
recs = {}
for c, (desc, (m, k, n), eb, peak) in CASES.items():
    alg = (m * k + k * n) * eb + m * n * 2
    flops = 2.0 * m * k * n
    dd = durs(f"c_{c}")
    main = {kk: v for kk, v in dd.items() if _is_main(kk)}
    d = sum(main.values()) / max(len(main), 1) if main else None
    rec = {"case": desc, "algorithmic_bytes_per_launch": alg, "flops_per_launch": flops, "kernel_durations_us": {kk[:110]: round(v, 2) for kk, v in dd.items()}}
    f = mean(f"f_{c}", "FETCH_SIZE"); w = mean(f"w_{c}", "WRITE_SIZE"); h = mean(f"w_{c}", "TCC_HIT_sum"); mi = mean(f"w_{c}", "TCC_MISS_sum")
    mf = mean(f"c_{c}", "SQ_VALU_MFMA_BUSY_CYCLES"); gui = mean(f"c_{c}", "GRBM_GUI_ACTIVE")
    bc = mean(f"l_{c}", "SQ_LDS_BANK_CONFLICT"); la = mean(f"l_{c}", "SQ_LDS_IDX_ACTIVE")
    if f is not None: rec["read_bytes_beyond_L2 (FETCH_SIZE KiB x 1024 x 2)"] = f * 2048
    if w is not None: rec["write_bytes"] = w * 1024
    if f is not None and w is not None:
        rec["hbm_bytes_per_launch (tile kernel only)"] = f * 2048 + w * 1024
        rec["traffic_over_algorithmic"] = (f * 2048 + w * 1024) / alg
    if h is not None and mi is not None: rec["l2_hit_rate"] = h / max(h + mi, 1)
    if d:
        rec["profiled_duration_us (tile kernel)"] = d
        rec["profiled_tflops (tile kernel)"] = flops / d / 1e6
        rec["frac_of_mfma_peak"] = flops / d / 1e6 / peak
    if gui and d: rec["sustained_clock_mhz"] = gui / 8 / d                 # GRBM_GUI_ACTIVE is summed over the 8 XCDs
    if mf and gui: rec["mfma_busy_frac"] = mf / (gui / 8 * 1024)           # 256 CUs x 4 SIMDs (of the whole chip, busy CUs or not)
    if mf and d: rec["mfma_busy_cycles_per_simd_over_duration_at_2400MHz"] = mf / 1024 / (d * 2400.0)   # no clock estimate needed: a lower bound of busy
    if gui and d and gui / 8 / d > 2400.0:
        rec["clock_note"] = ("GRBM_GUI_ACTIVE / duration exceeds the 2 400 MHz maximum: the counter also covers the dispatch around a launch this "
                             "short, so `sustained_clock_mhz` is not a clock here and `mfma_busy_frac` (per GUI-active cycle) is deflated by the same factor")
    if la and gui: rec["lds_active_frac_of_cu_cycles"] = la / (gui / 8 * 256)
    if bc is not None and la: rec["lds_conflict_frac_of_lds_cycles"] = bc / la
    for kk in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY"):
        v = mean(f"s_{c}", kk)
        if v is not None: rec[kk] = v
    if rec.get("SQ_WAVE_CYCLES"):
        wc = rec["SQ_WAVE_CYCLES"]
        rec["wave_cycle_split"] = {kk: rec[kk] / wc for kk in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY") if kk in rec}
    if isinstance(wall, dict) and c in wall: rec["wall_same_binary_eager_events"] = wall[c]
    recs[c] = rec
out = {"collected": datetime.date.today().isoformat(),
       "tool": "rocprofv3 --kernel-trace --pmc, one counter group per pass (scripts/profile_r5_tile128.sh); first three launches of every "
               "pass dropped; FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md); counters are those of mojo::g128::gemm128_kernel launches only",
       "cases": recs}
json.dump(out, open(f"{OUT}/r5_gemm_tile128_counters.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:5000])
