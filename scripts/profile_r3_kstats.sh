# per-kernel durations of one extras function:  bash scripts/profile_r3_kstats.sh <tag> <bench function> [MOJO_BENCH_ONLY substring]
cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
TAG=$1; FN=$2; export MOJO_BENCH_ONLY="${3:-}"
P=gpurun_out/prof_r3_ks_$TAG; rm -rf $P; mkdir -p $P
rocprofv3 --kernel-trace --stats --output-format csv -d $P -- python3 benchmarks/one.py $FN > $P/run.log 2>&1; echo rc=$?
python3 - "$P" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:14]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.2f} min {float(r['MinNs'])/1e3:8.2f} pct {r['Percentage']}")
PY
