cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out
L=gpurun_out/r2n.log; : > $L
run() { echo "== $1" | tee -a $L; shift; timeout -k 10 "$@" >> $L 2>&1; echo "rc=$?" | tee -a $L; }
run mla 300 python -u -m pytest tests/test_hip_mla.py tests/test_hip_graph.py -x -q -m gpu
run bench_mla 200 python -u benchmarks/one.py bench_mla_prefill
grep -E "^== |^rc=|passed|failed|^E  |bench_mla" $L | cut -c1-900 | tail -20
