cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out; rm -f gpurun_out/comm_ranks_progress*.log gpurun_out/r2h_alive.log
L=gpurun_out/r2h.log; : > $L
( while true; do sleep 60; echo "[alive $(date +%T)]" >> gpurun_out/r2h_alive.log; done ) &
ALIVE=$!
run() { echo "== $1" | tee -a $L; shift; timeout -k 10 "$@" >> $L 2>&1; echo "rc=$?" | tee -a $L; }
MOJO_HIP_PEER_TIMEOUT_MS=8000 run suite 900 python -u -m pytest tests -q -m gpu --durations=6
run bench 500 python -u bench.py
tail -1 $L > gpurun_out/r2h_bench.json
kill $ALIVE
grep -E "^== |^rc=|passed|failed|^E  |s call" $L | cut -c1-300 | tail -30
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2h_bench.json').read())
print({k:d[k] for k in ('value','ms_per_step')}, 'frac', d['roofline']['frac'], 'cpu', d.get('cpu_baseline'))
print('gg', d.get('roofline_group_gemm',{}).get('achieved'))
ex=d['extras']
for k in ('MojoPagedDecodeGQA_bf16_other_contexts','MojoPagedPrefillMLA_bf16','MojoPagedDecodeMLA_bf16'):
    print(k, json.dumps(ex.get(k))[:900])
cc=ex.get('compute_comm_bf16',{})
print('cc keys', len(cc), list(cc.items())[:2])
PY
