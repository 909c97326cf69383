# HBM-side traffic of the MLA decode step (B 64, ctx 4096, DeepSeek-V3 dims): separate --pmc passes (FETCH_SIZE costs 3 of
# the 4 TCC slots, WRITE_SIZE 2), plus an L2 hit/miss pass.  Summary -> gpurun_out/r3_mla_decode_traffic.json
cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=gpurun_out/prof_r3_mla; rm -rf $P; mkdir -p $P
TAG=${1:-r3}
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- python3 scripts/probes/mla_decode_driver.py 64 4096 > $P/stats.log 2>&1; echo stats rc=$?
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/fetch -- python3 scripts/probes/mla_decode_driver.py 64 4096 > $P/fetch.log 2>&1; echo fetch rc=$?
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/write -- python3 scripts/probes/mla_decode_driver.py 64 4096 > $P/write.log 2>&1; echo write rc=$?
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $P/l2 -- python3 scripts/probes/mla_decode_driver.py 64 4096 > $P/l2.log 2>&1; echo l2 rc=$?
python3 - "$TAG" <<'PY'
import csv, glob, json, datetime, collections, sys
P = "gpurun_out/prof_r3_mla"
tag = sys.argv[1]
def rows(pat):
    f = glob.glob(f"{P}/{pat}", recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
dur = collections.defaultdict(list)
for r in rows("stats/**/*kernel_trace.csv"):
    dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for part in ("fetch", "write", "l2"):
    for r in rows(f"{part}/**/*counter_collection.csv"):
        cnt[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
B, ctx, r_, rope = 64, 4096, 512, 64
latent_bytes = B * ctx * (r_ + rope) * 2
out = {"collected": datetime.date.today().isoformat(), "tool": "rocprofv3 --kernel-trace --pmc (ROCm 7.2), separate passes; scripts/profile_r3_mla_traffic.sh",
       "case": "MojoPagedDecodeMLA bf16, B=64, ctx=4096, H=128, nope 128 / rope 64 / v 128 / r 512, page 16 (30 calls)",
       "latent_cache_bytes": latent_bytes,
       "fetch_correction": "x2 (gfx950: FETCH_SIZE tallies the 128-B requests of 16 B/lane streaming reads at 64 B; unit KiB)",
       "kernels": {}}
for k, d in dur.items():
    if "mla" not in k and "gemm" not in k and "skinny" not in k and "merge" not in k:
        continue
    rec = {"launches": len(d), "avg_duration_us": sum(d[5:]) / max(len(d[5:]), 1) / 1e3}
    c = cnt.get(k, {})
    for name, v in c.items():
        rec[name] = sum(v[5:]) / max(len(v[5:]), 1)
    if "FETCH_SIZE" in rec:
        rec["read_bytes_corrected"] = rec["FETCH_SIZE"] * 1024 * 2
        rec["reads_over_latent_cache"] = rec["read_bytes_corrected"] / latent_bytes
    if "WRITE_SIZE" in rec:
        rec["write_bytes"] = rec["WRITE_SIZE"] * 1024
    if "TCC_HIT_sum" in rec:
        rec["l2_hit_rate"] = rec["TCC_HIT_sum"] / max(rec["TCC_HIT_sum"] + rec["TCC_MISS_sum"], 1)
    out["kernels"][k[:120]] = rec
json.dump(out, open(f"gpurun_out/{tag}_mla_decode_traffic.json", "w"), indent=1)
for k, v in out["kernels"].items():
    print(k[:70], {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
PY
