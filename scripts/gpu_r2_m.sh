cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out; rm -f gpurun_out/comm_ranks_progress*.log
L=gpurun_out/r2m.log; : > $L
( while true; do sleep 60; echo "[alive $(date +%T)]" >> gpurun_out/r2m_alive.log; done ) &
ALIVE=$!
run() { echo "== $1" | tee -a $L; shift; timeout -k 10 "$@" >> $L 2>&1; echo "rc=$?" | tee -a $L; }
MOJO_HIP_PEER_TIMEOUT_MS=8000 run comm 460 python -u -m pytest tests/test_hip_comm_ranks.py tests/test_hip_comm.py -x -q -m gpu
run decode 300 python -u -m pytest tests/test_hip_decode_gqa.py -x -q -m gpu
kill $ALIVE
grep -E "^== |^rc=|passed|failed|^E  " $L | cut -c1-300 | tail -20
grep -c "direct:\|check" gpurun_out/comm_ranks_progress_rank0.log
