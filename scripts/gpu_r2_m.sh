cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out; L=gpurun_out/r2m.log; : > $L
timeout -k 10 600 python -u -m pytest tests/test_hip_mla.py tests/test_hip_graph.py -q -m gpu -x > gpurun_out/r2m_tests.log 2>&1; echo "tests rc=$?" | tee -a $L
tail -3 gpurun_out/r2m_tests.log | cut -c1-220
for k in oct oct; do
MOJO_HIP_MLA_KERNEL=$k timeout -k 10 200 python -u benchmarks/one.py bench_mla_decode >> $L 2>&1
done
grep -E "bench_mla" $L | cut -c1-120
