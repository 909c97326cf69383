cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out; rm -f gpurun_out/comm_ranks_progress*.log gpurun_out/mla_error_triples.jsonl
L=gpurun_out/r2d.log; : > $L
run() { echo "== $1" | tee -a $L; shift; timeout -k 10 "$@" >> $L 2>&1; echo "rc=$?" | tee -a $L; }
run mla 600 python -u -m pytest tests/test_hip_mla.py -x -q -m gpu
run decode 300 python -u -m pytest tests/test_hip_decode_gqa.py tests/test_hip_graph.py tests/test_hip_paged_cache.py -x -q -m gpu
MOJO_HIP_PEER_TIMEOUT_MS=3000 run direct 200 python -u -m pytest tests/test_hip_comm_ranks.py::test_hip_compute_comm_two_ranks_direct_peer_exchange -x -q -m gpu -s
run chunks 200 python -u -m pytest tests/test_hip_comm_ranks.py::test_hip_compute_comm_two_ranks_rccl_pipeline_layout -x -q -m gpu -s
run bench_small 400 python -u benchmarks/one.py bench_mla_prefill bench_decode_variants
grep -E "^== |^rc=|passed|failed|^E  |MLA_ERROR|bench_" $L | cut -c1-1500 | tail -60
for r in 0 1; do echo "--- rank $r progress"; tail -5 gpurun_out/comm_ranks_progress_rank$r.log 2>/dev/null | cut -c1-300; done
