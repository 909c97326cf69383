cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out
( while true; do sleep 60; echo "[alive $(date +%T)]" >> gpurun_out/bench_alive.log; done ) &
ALIVE=$!
T0=$(date +%s); timeout -k 10 900 python -u bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
echo "bench rc=$? wall=$(( $(date +%s) - T0 )) s"; kill $ALIVE

python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_full.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','n_gpus')}, d['roofline']['frac'], d['roofline'].get('traffic_source'))
print(d.get('roofline_group_gemm'))
print(d.get('cpu_baseline'))
PY
