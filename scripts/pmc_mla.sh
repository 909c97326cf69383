cd /root/repo; export TMPDIR=/tmp; rm -rf gpurun_out/pmc_mla; mkdir -p gpurun_out/pmc_mla
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_mla/a -- python benchmarks/mla_bench.py > gpurun_out/pmc_mla/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_mla/b -- python benchmarks/mla_bench.py > gpurun_out/pmc_mla/b.log 2>&1
python - <<'PY'
import csv, glob, collections
kn="mla512_oct"
for part in "ab":
    f = glob.glob(f"gpurun_out/pmc_mla/{part}/*/*counter_collection.csv")
    if not f: print("missing", part); continue
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        if kn in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k,v in agg.items(): print(f"  {k:32s} n={len(v):3d} mean={sum(v)/len(v):.5g}")
PY
