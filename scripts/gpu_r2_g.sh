cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out; rm -f gpurun_out/comm_ranks_progress*.log
L=gpurun_out/r2g.log; : > $L
( while true; do sleep 60; echo "[alive $(date +%T)]" >> gpurun_out/r2g_alive.log; done ) &
ALIVE=$!
run() { echo "== $1" | tee -a $L; shift; timeout -k 10 "$@" >> $L 2>&1; echo "rc=$?" | tee -a $L; }
run quant 400 python -u -m pytest tests/test_hip_quant_gemm.py -x -q -m gpu -k "full_size or split_k"
MOJO_HIP_PEER_TIMEOUT_MS=8000 run suite 1000 python -u -m pytest tests -x -q -m gpu --durations=8
kill $ALIVE
grep -E "^== |^rc=|passed|failed|^E  |s call|s setup" $L | cut -c1-400 | tail -40
