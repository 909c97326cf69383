# GroupGemm memory-side traffic (Mixtral up-projection): FETCH_SIZE, WRITE_SIZE and the L2 hit counters, one --pmc pass each.
cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=gpurun_out/prof_r2g; mkdir -p $P
G="python3 benchmarks/gemm_bench.py --m 16384 --k 4096 --n 28672 --groups 8"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/gg_fetch -- $G > $P/gg_fetch.log 2>&1; echo gg_fetch rc=$?
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/gg_write -- $G > $P/gg_write.log 2>&1; echo gg_write rc=$?
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $P/gg_l2 -- $G > $P/gg_l2.log 2>&1; echo gg_l2 rc=$?
python3 - <<'PY'
import csv, glob, json, os, datetime
P = "gpurun_out/prof_r2g"
def mean(pass_, counter):
    f = max(glob.glob(f"{P}/{pass_}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "gemm256_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(v) / len(v), len(v)
fetch, n = mean("gg_fetch", "FETCH_SIZE"); write, _ = mean("gg_write", "WRITE_SIZE")
hit, _ = mean("gg_l2", "TCC_HIT_sum"); miss, _ = mean("gg_l2", "TCC_MISS_sum")
m, k, nn, g = 16384, 4096, 28672, 8
alg = m * k * 2 + g * k * nn * 2 + m * nn * 2
out = {"collected": datetime.date.today().isoformat(), "tool": "rocprofv3 --pmc (ROCm 7.2), one MI355X box via gpurun; script scripts/profile_r2_gemm_traffic.sh",
       "shape": "M=16384 K=4096 N=28672 G=8 bf16 (Mixtral up-projection), random data", "launches": n,
       "FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB_raw": write,
       "fetch_correction": "x2 (gfx950: 128-B requests of 16 B/lane streaming reads tallied at 64 B)",
       "memory_side_read_bytes": 2 * fetch * 1024, "memory_side_write_bytes": write * 1024,
       "algorithmic_bytes": alg, "algorithmic_read_bytes": m * k * 2 + g * k * nn * 2, "algorithmic_write_bytes": m * nn * 2,
       "read_over_algorithmic": 2 * fetch * 1024 / (m * k * 2 + g * k * nn * 2), "write_over_algorithmic": write * 1024 / (m * nn * 2),
       "l2_hit_rate": hit / (hit + miss), "TCC_HIT_sum": hit, "TCC_MISS_sum": miss}
json.dump(out, open("gpurun_out/r2_group_gemm_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
