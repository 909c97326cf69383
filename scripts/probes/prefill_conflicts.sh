cd /root/repo; export TMPDIR=/tmp
P=gpurun_out/prof_conf; rm -rf $P; mkdir -p $P
MOJO_BENCH_ONLY=16_ragged rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $P/a -- python3 benchmarks/one.py bench_prefill > $P/a.log 2>&1
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob("gpurun_out/prof_conf/a/**/*counter_collection.csv", recursive=True)[0])):
    if "prefill_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
print(m, m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"])
PY
