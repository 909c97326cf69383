cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out; rm -f gpurun_out/comm_ranks_progress*.log gpurun_out/mla_error_triples.jsonl
L=gpurun_out/r2f.log; : > $L
run() { echo "== $1" | tee -a $L; shift; timeout -k 10 "$@" >> $L 2>&1; echo "rc=$?" | tee -a $L; }
MOJO_HIP_PEER_TIMEOUT_MS=8000 run comm 460 python -u -m pytest tests/test_hip_comm_ranks.py -x -q -m gpu -s
run mla 300 python -u -m pytest tests/test_hip_mla.py tests/test_hip_store_mla.py -x -q -m gpu
run quant 500 python -u -m pytest tests/test_hip_quant_gemm.py tests/test_hip_group_gemm.py -x -q -m gpu
grep -E "^== |^rc=|passed|failed|^E  |direct_exchange" $L | cut -c1-600 | tail -40
for r in 0 1; do echo "--- rank $r progress"; tail -3 gpurun_out/comm_ranks_progress_rank$r.log 2>/dev/null | cut -c1-300; done
