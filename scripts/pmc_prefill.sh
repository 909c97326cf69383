cd /root/repo; export TMPDIR=/tmp; rm -rf gpurun_out/pmc_pf; mkdir -p gpurun_out/pmc_pf
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_pf/a -- python benchmarks/prefill_bench.py > gpurun_out/pmc_pf/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_pf/b -- python benchmarks/prefill_bench.py > gpurun_out/pmc_pf/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_TRANS SQ_WAIT_INST_VMEM SQ_WAVES --output-format csv -d gpurun_out/pmc_pf/c -- python benchmarks/prefill_bench.py > gpurun_out/pmc_pf/c.log 2>&1
python - <<'PY'
import csv, glob, collections
kn="prefill_kernel"
for part in "abc":
    f = glob.glob(f"gpurun_out/pmc_pf/{part}/*/*counter_collection.csv")
    if not f: print("missing", part); continue
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        if kn in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k,v in agg.items(): print(f"  {k:32s} n={len(v):3d} mean={sum(v)/len(v):.5g} min={min(v):.5g} max={max(v):.5g}")
PY
