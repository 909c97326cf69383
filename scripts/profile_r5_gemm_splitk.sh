# Round 5: rocprofv3 kernel-trace summaries of K-split products on the 128-row tiles, warm and cold weights
cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=gpurun_out/prof_r5_splitk; rm -rf $P; mkdir -p $P
for c in "256 8192 1024" "256 8192 1024 cold" "512 4096 4096" "512 4096 4096 cold" "1024 4096 4096" "1024 4096 4096 cold" "1024 14336 4096 cold"; do
  tag=$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $P/$tag -- python3 scripts/probes/gemm_splitk_trace.py $c > $P/$tag.log 2>&1; echo "$tag rc=$?"; tail -1 $P/$tag.log
done
python3 - <<'PY'
import csv, glob, os
out = open('gpurun_out/prof_r5_splitk/summary.csv', 'w')
out.write('case,kernel,calls,avg_us,min_us,max_us\n')
for d in sorted(glob.glob('gpurun_out/prof_r5_splitk/*/')):
    for f in glob.glob(d + '**/*kernel_stats.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            name = r['Name']
            if 'gemm' in name or 'finalize' in name:
                out.write('%s,"%s",%s,%.2f,%.2f,%.2f\n' % (os.path.basename(d.rstrip('/')), name[:90], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
out.close()
print(open('gpurun_out/prof_r5_splitk/summary.csv').read())
PY
