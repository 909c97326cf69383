cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out
( while true; do sleep 60; echo "[alive $(date +%T)]" >> gpurun_out/r2l_alive.log; done ) &
ALIVE=$!
# `--gpus 2` with no launcher around it: bench.py must start the two ranks itself.  gloo: both ranks share the box's one GPU
# (RCCL refuses that); the direct peer-exchange variant runs for real, the "rccl" variant goes over gloo's host staging.
MOJO_BENCH_DIST_BACKEND=gloo MOJO_HIP_PEER_TIMEOUT_MS=8000 OMP_NUM_THREADS=6 timeout -k 10 1000 python -u bench.py --gpus 2 --steps 20 --warmup 3 --extras-deadline 800 > gpurun_out/r2l_bench2.json 2> gpurun_out/r2l_bench2.err
echo "bench --gpus 2 rc=$?"
kill $ALIVE
python - <<'PY'
import json
try:
    d=json.loads([l for l in open('gpurun_out/r2l_bench2.json') if l.startswith('{')][-1])
    print('n_gpus', d['n_gpus'], 'value', round(d['value']), 'ms', round(d['ms_per_step'],3))
    cc=d['extras']['compute_comm_bf16']
    if 'error' in cc: print(cc)
    for k,v in cc.items():
        if 'M4096_K28672' in k or 'error' in v: print(k, {a:(round(b,1) if isinstance(b,float) else b) for a,b in v.items()})
except Exception as e:
    print('parse failed', e); print(open('gpurun_out/r2l_bench2.err').read()[-3000:])
PY
