# Counter passes for the two paged-prefill kernels (prefill_kernel vs prefill_w64_kernel) on the bench cases: one case per
# process, two --pmc passes each, --kernel-trace only.  Summary -> gpurun_out/r4_attention_counters.json
cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=gpurun_out/prof_r4a; rm -rf $P; mkdir -p $P
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
B="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
one() {  # tag, W64 switch, case substring
  export MOJO_HIP_PREFILL_W64=$2 MOJO_BENCH_ONLY="$3"
  rocprofv3 --kernel-trace --pmc $A --output-format csv -d $P/$1_a -- python3 benchmarks/one.py bench_prefill > $P/$1_a.log 2>&1; echo $1 a rc=$?
  rocprofv3 --kernel-trace --pmc $B --output-format csv -d $P/$1_b -- python3 benchmarks/one.py bench_prefill > $P/$1_b.log 2>&1; echo $1 b rc=$?
}
for c in ${CASES:-16k:1x16384 4x2048:4x2048_nocache ragged:16_ragged cached:4x2048_cached2048}; do
  tag=${c%%:*}; sub=${c##*:}
  for w in ${ARMS:-0 1}; do one pf_${tag}_w$w $w $sub; done
done
python3 - <<'PY'
import csv, glob, json, os, datetime, collections
P = "gpurun_out/prof_r4a"
out = {"collected": datetime.date.today().isoformat(),
       "tool": "rocprofv3 --kernel-trace --pmc (ROCm 7.2), one MI355X box via gpurun; scripts/profile_r4_prefill.sh",
       "note": "means over the launches of the named kernel in one process per case; SQ_* are sums over the chip; derived: "
               "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs), clock = GRBM_GUI_ACTIVE / 8 / duration, "
               "lds_active_frac_per_cu = SQ_LDS_IDX_ACTIVE / cycles / 256", "cases": {}}
for d in sorted(glob.glob(f"{P}/pf_*_a")):
    tag = os.path.basename(d)[:-2]
    kn = "prefill_w64_kernel" if tag.endswith("_w1") else "prefill_kernel"
    rec = {"kernel": kn}
    for part in "ab":
        f = glob.glob(f"{P}/{tag}_{part}/**/*counter_collection.csv", recursive=True)
        if not f:
            continue
        agg = collections.defaultdict(list)
        for row in csv.DictReader(open(f[0])):
            if kn in row["Kernel_Name"]:
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in agg.items():
            rec[k] = sum(v) / len(v)
        rec["launches_" + part] = max((len(v) for v in agg.values()), default=0)
        tr = glob.glob(f"{P}/{tag}_{part}/**/*kernel_trace.csv", recursive=True)
        if tr:
            dd = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(tr[0])) if kn in r["Kernel_Name"]]
            if dd:
                rec["avg_duration_us_pass_" + part] = sum(dd) / len(dd) / 1e3
    w = rec.get("SQ_WAVE_CYCLES")
    if w:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU"):
            if k in rec:
                rec[k.lower() + "_over_wave_cycles"] = rec[k] / w
    if "GRBM_GUI_ACTIVE" in rec and "avg_duration_us_pass_b" in rec:
        clk = rec["GRBM_GUI_ACTIVE"] / 8 / rec["avg_duration_us_pass_b"]
        rec["sustained_clock_mhz"] = clk
        cycles = clk * rec["avg_duration_us_pass_b"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in rec:
            rec["mfma_busy_frac"] = rec["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * 1024)
        if "SQ_LDS_IDX_ACTIVE" in rec:
            rec["lds_active_frac_per_cu"] = rec["SQ_LDS_IDX_ACTIVE"] / (cycles * 256)
            rec["lds_bank_conflict_frac"] = rec.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(rec["SQ_LDS_IDX_ACTIVE"], 1.0)
    out["cases"][tag] = rec
json.dump(out, open("gpurun_out/r4_attention_counters.json", "w"), indent=1)
for tag, rec in out["cases"].items():
    print(tag, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in rec.items() if k.endswith("frac") or k.endswith("cycles") or "clock" in k or "duration" in k or k.endswith("per_cu")})
PY
