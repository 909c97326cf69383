cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/pmc_attn
for t in prefill mla; do
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_attn/${t}_a -- python benchmarks/${t}_bench.py > gpurun_out/pmc_attn/${t}_a.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_attn/${t}_b -- python benchmarks/${t}_bench.py > gpurun_out/pmc_attn/${t}_b.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for t, kn in (("prefill","prefill_kernel"),("mla","mla_latent_kernel")):
    print("==", t)
    for part in "ab":
        f = glob.glob(f"gpurun_out/pmc_attn/{t}_{part}/*/*counter_collection.csv")
        if not f: print("missing"); continue
        agg = collections.defaultdict(list)
        for row in csv.DictReader(open(f[0])):
            if kn in row["Kernel_Name"]:
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k,v in agg.items(): print(f"  {k:32s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
    f = glob.glob(f"gpurun_out/pmc_attn/{t}_a/*/*kernel_trace.csv")[0]
    d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"])) for r in csv.DictReader(open(f)) if kn in r["Kernel_Name"]]
    print("  kernel us mean", sum(d)/len(d)/1e3, "n", len(d))
PY
