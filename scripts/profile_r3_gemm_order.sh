# GroupGemm tile orders: wall (interleaved A/B) + FETCH_SIZE / WRITE_SIZE / clock / MFMA busy per order -> gpurun_out/r3_group_gemm_order.json
cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=gpurun_out/prof_r3_order; rm -rf $P; mkdir -p $P
python3 scripts/probes/gemm_order_ab.py > $P/ab.log 2>&1; echo ab rc=$?; tail -1 $P/ab.log
for o in 0 1 2; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/f$o -- python3 scripts/probes/gemm_order_ab.py one $o > $P/f$o.log 2>&1; echo fetch $o rc=$?
  rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $P/w$o -- python3 scripts/probes/gemm_order_ab.py one $o > $P/w$o.log 2>&1; echo write $o rc=$?
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $P/c$o -- python3 scripts/probes/gemm_order_ab.py one $o > $P/c$o.log 2>&1; echo clock $o rc=$?
done
python3 - <<'PY'
import csv, glob, json, datetime
P = "gpurun_out/prof_r3_order"
out = {"collected": datetime.date.today().isoformat(), "tool": "rocprofv3 --kernel-trace --pmc, separate passes; scripts/profile_r3_gemm_order.sh",
       "case": "MojoGroupGemm bf16 16384 x 4096 x 28672, 8 experts balanced, weights [G,K,N], random data",
       "orders": {"0": "today: each XCD its own run of panels (8 m x 4 n tiles per XCD at a time)", "1": "no XCD remap: the eight XCDs share every panel (64 m x 4 n in flight)",
                  "2": "as 0 with panels of 2 n-tiles (16 m x 2 n per XCD)"}}
try:
    out["wall"] = json.loads(open(f"{P}/ab.log").read().strip().split("\n")[-1])
except Exception as e:
    out["wall"] = {"error": repr(e)}
def mean(pat, counter):
    f = glob.glob(f"{P}/{pat}/**/*counter_collection.csv", recursive=True)
    if not f: return None
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if "gemm256" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(v[2:]) / max(len(v[2:]), 1) if v else None
def dur(pat):
    f = glob.glob(f"{P}/{pat}/**/*kernel_trace.csv", recursive=True)
    if not f: return None
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f[0])) if "gemm256" in r["Kernel_Name"]]
    return sum(d[2:]) / max(len(d[2:]), 1) / 1e3 if d else None
alg = 2 * (16384 * 4096 + 8 * 4096 * 28672 + 16384 * 28672)
for o in "012":
    rec = {}
    f = mean(f"f{o}", "FETCH_SIZE"); w = mean(f"w{o}", "WRITE_SIZE"); h = mean(f"w{o}", "TCC_HIT_sum"); mi = mean(f"w{o}", "TCC_MISS_sum")
    mf = mean(f"c{o}", "SQ_VALU_MFMA_BUSY_CYCLES"); gui = mean(f"c{o}", "GRBM_GUI_ACTIVE"); d = dur(f"c{o}")
    if f is not None: rec["read_bytes_beyond_L2 (FETCH_SIZE KiB x 1024 x 2)"] = f * 2048
    if w is not None: rec["write_bytes"] = w * 1024
    if f is not None and w is not None: rec["traffic_over_algorithmic"] = (f * 2048 + w * 1024) / alg
    if h is not None and mi is not None: rec["l2_hit_rate"] = h / max(h + mi, 1)
    if d: rec["profiled_duration_us"] = d
    if gui and d: rec["sustained_clock_mhz"] = gui / 8 / d
    if mf and gui: rec["mfma_busy_frac"] = mf / (gui / 8 * 1024)
    out["orders_measured_" + o] = rec
json.dump(out, open("gpurun_out/r3_group_gemm_order.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
