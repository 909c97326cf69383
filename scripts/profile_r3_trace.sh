# kernel trace of a few MLA decode steps (durations + gaps):  bash scripts/profile_r3_trace.sh <tag> [env assignments...]
cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
P=gpurun_out/prof_r3_trace_$TAG; rm -rf $P; mkdir -p $P
rocprofv3 --kernel-trace --output-format csv -d $P -- python3 scripts/probes/mla_decode_driver.py ${MLA_B:-64} ${MLA_CTX:-4096} > $P/run.log 2>&1; echo trace rc=$?
python3 - "$P" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-8 * 4:]          # the last 8 calls (4 kernels each)
prev_end = None
agg = collections.OrderedDict()
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-48:]
    d = agg.setdefault(name, {"dur": [], "gap": []})
    d["dur"].append((e - s) / 1e3)
    if prev_end is not None:
        d["gap"].append((s - prev_end) / 1e3)
    prev_end = e
tot = 0
for k, v in agg.items():
    du = sum(v["dur"]) / len(v["dur"]); ga = sum(v["gap"]) / max(len(v["gap"]), 1)
    tot += du + ga
    print(f"{k:50s} dur {du:7.2f} us   gap before {ga:6.2f} us   launches {len(v['dur'])}")
print("sum per call (dur + gaps):", round(tot, 2), "us  (eager launches: gaps include host launch time)")
PY
