cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out
L=gpurun_out/r2p.log; : > $L
run() { echo "== $1" | tee -a $L; shift; timeout -k 10 "$@" >> $L 2>&1; echo "rc=$?" | tee -a $L; }
run gemm_tests 500 python -u -m pytest tests/test_hip_group_gemm.py tests/test_hip_quant_gemm.py tests/test_hip_moe.py tests/test_hip_gemm_skinny.py -x -q -m gpu
for i in 1 2; do
  for SH in "--m 16384 --k 4096 --n 28672 --groups 8" "--m 16384 --k 4096 --n 28672 --groups 8 --trans" "--m 16384 --k 14336 --n 4096 --groups 8" "--m 20480 --k 4096 --n 4096 --groups 8" "--m 4096 --k 4096 --n 28672 --groups 8" "--m 10240 --k 512 --n 32768 --groups 1 --trans"; do
    echo "case $SH" >> $L
    timeout -k 10 120 python -u benchmarks/gemm_bench.py $SH >> $L 2>&1
  done
done
run quant_bench 200 python -u benchmarks/one.py bench_quant_gemm
grep -E "^== |^rc=|passed|failed|^E  " $L | tail
python - <<'PY'
import json
cur=None;res={}
for l in open('gpurun_out/r2p.log'):
    if l.startswith('case '): cur=l.strip()
    elif l.startswith('{"us"') and cur: res.setdefault(cur,[]).append(round(json.loads(l)['tflops']))
    elif l.startswith('{"bench_quant_gemm"'):
        q=json.loads(l)['bench_quant_gemm']; print({k:round(v['tflops']) for k,v in q.items() if 'tflops' in v})
for k,v in res.items(): print(k,v)
PY
