cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out; L=gpurun_out/r2m.log; : > $L
timeout -k 10 600 python -u -m pytest tests/test_hip_prefill_gqa.py tests/test_hip_graph.py -q -m gpu -x > gpurun_out/r2m_tests.log 2>&1; echo "tests rc=$?" | tee -a $L
for rep in 1 2; do for sk in 1 0; do
echo "== prefill bench compact=$sk" | tee -a $L
MOJO_HIP_PREFILL_COMPACT=$sk timeout -k 10 300 python -u benchmarks/one.py bench_prefill >> $L 2>&1; echo "rc=$?" | tee -a $L
done; done
tail -3 gpurun_out/r2m_tests.log
python - <<'PY'
import json,re
for line in open('gpurun_out/r2m.log'):
    if line.startswith('=='): print(line.strip())
    if line.startswith('{'):
        d=json.loads(line)['bench_prefill']; print({k:round(v['us'],1) for k,v in d.items()})
PY
