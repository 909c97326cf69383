cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out; L=gpurun_out/r2m.log; : > $L
timeout -k 10 800 python -u -m pytest tests/test_hip_quant_gemm.py tests/test_c_abi.py -q -m gpu -x > gpurun_out/r2m_tests.log 2>&1; echo "tests rc=$?" | tee -a $L
tail -2 gpurun_out/r2m_tests.log
MOJO_BENCH_ONLY=_32x timeout -k 10 200 python -u benchmarks/one.py bench_quant_gemm >> $L 2>&1
MOJO_BENCH_ONLY=_128x timeout -k 10 200 python -u benchmarks/one.py bench_quant_gemm >> $L 2>&1
python - <<'PY'
import json
for line in open('gpurun_out/r2m.log'):
    if line.startswith('{'):
        d=json.loads(line)['bench_quant_gemm']; print({k:(round(v['us'],1), round(v['frac_of_hbm_peak'],3)) for k,v in d.items()})
PY
