cd /root/repo; export TMPDIR=/tmp; rm -rf gpurun_out/pmc_mla; mkdir -p gpurun_out/pmc_mla
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_mla/a -- python benchmarks/mla_bench.py > gpurun_out/pmc_mla/a.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d gpurun_out/pmc_mla/b -- python benchmarks/mla_bench.py > gpurun_out/pmc_mla/b.log 2>&1
python - <<'PY'
import csv, glob, collections
kn="mla512"
for part in "ab":
    f = glob.glob(f"gpurun_out/pmc_mla/{part}/*/*counter_collection.csv")
    if not f: print("missing", part); continue
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        if kn in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k,v in agg.items(): print(f"  {k:32s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
tail -3 gpurun_out/pmc_mla/b.log
