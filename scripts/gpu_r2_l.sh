cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out; L=gpurun_out/r2l.log; : > $L
echo "== mla tests" | tee -a $L
timeout -k 10 600 python -u -m pytest tests/test_hip_mla.py tests/test_c_abi.py tests/test_hip_graph.py -q -m gpu -x >> $L 2>&1; echo "rc=$?" | tee -a $L
echo "== mla decode bench fused" | tee -a $L
timeout -k 10 200 python -u benchmarks/one.py bench_mla_decode >> $L 2>&1
echo "== mla decode bench separate merge" | tee -a $L
MOJO_HIP_MLA_FUSED_MERGE=0 timeout -k 10 200 python -u benchmarks/one.py bench_mla_decode >> $L 2>&1
grep -E "^== |^rc=|passed|failed|^E  |bench_mla" $L | cut -c1-400
