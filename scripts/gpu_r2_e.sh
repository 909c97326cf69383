cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out; rm -f gpurun_out/comm_ranks_progress*.log gpurun_out/mla_error_triples.jsonl
L=gpurun_out/r2e.log; : > $L
run() { echo "== $1" | tee -a $L; shift; timeout -k 10 "$@" >> $L 2>&1; echo "rc=$?" | tee -a $L; }
python -c "import os,torch; print('visible', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)), 'torch', torch.get_num_threads()); print(open('/sys/fs/cgroup/cpu.max').read() if os.path.exists('/sys/fs/cgroup/cpu.max') else 'no cpu.max')" | tee -a $L
run chunks 450 python -u -m pytest tests/test_hip_comm_ranks.py::test_hip_compute_comm_two_ranks_rccl_pipeline_layout -x -q -m gpu -s
MOJO_HIP_PEER_TIMEOUT_MS=5000 run direct 450 python -u -m pytest tests/test_hip_comm_ranks.py::test_hip_compute_comm_two_ranks_direct_peer_exchange -x -q -m gpu -s
run mla 600 python -u -m pytest tests/test_hip_mla.py tests/test_hip_store_mla.py -x -q -m gpu
run bench_small 400 python -u benchmarks/one.py bench_mla_prefill
grep -E "^== |^rc=|passed|failed|^E  |bench_|visible|max" $L | cut -c1-1200 | tail -50
for r in 0 1; do echo "--- rank $r progress"; tail -4 gpurun_out/comm_ranks_progress_rank$r.log 2>/dev/null | cut -c1-300; done
