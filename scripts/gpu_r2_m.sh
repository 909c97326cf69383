cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out; L=gpurun_out/r2m.log; : > $L
run() {
  MOJO_BENCH_ONLY=_32x timeout -k 10 200 python -u benchmarks/one.py bench_quant_gemm >> $L 2>&1
  MOJO_BENCH_ONLY=_128x timeout -k 10 200 python -u benchmarks/one.py bench_quant_gemm >> $L 2>&1
  timeout -k 10 200 python -u benchmarks/one.py bench_dense_decode bench_mla_decode >> $L 2>&1
}
echo "== depth 3" | tee -a $L; run
for d in 5 7; do
echo "== depth $d" | tee -a $L
touch mojo_opset_amd/csrc/gemm_skinny.hip mojo_opset_amd/csrc/quant_gemm.hip
MOJO_HIP_EXTRA_CXXFLAGS=-DSKINNY_DEPTH=$d timeout -k 10 600 python -m mojo_opset_amd.csrc.build -j 8 > gpurun_out/r2m_build.log 2>&1; echo "build rc=$?" | tee -a $L
run
done
timeout -k 10 600 python -u -m pytest tests/test_hip_quant_gemm.py tests/test_hip_gemm.py tests/test_hip_mla.py -q -m gpu -x 2>&1 | tail -1 | tee -a $L
python - <<'PY'
import json
for line in open('gpurun_out/r2m.log'):
    if line.startswith('==') or 'passed' in line or 'failed' in line: print(line.strip())
    if line.startswith('{'):
        for name,d in json.loads(line).items():
            print('  ',name[:18],{k:round(v['us'],1) for k,v in d.items()})
PY
