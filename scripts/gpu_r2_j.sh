cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out; rm -f gpurun_out/comm_ranks_progress*.log gpurun_out/r2j_alive.log
L=gpurun_out/r2j.log; : > $L
( while true; do sleep 60; echo "[alive $(date +%T)] $(tail -c 120 $L | tr '\n' ' ')" >> gpurun_out/r2j_alive.log; done ) &
ALIVE=$!
echo "== suite" | tee -a $L
MOJO_HIP_PEER_TIMEOUT_MS=8000 timeout -k 10 1100 python -u -m pytest tests -q -m gpu --durations=15 >> $L 2>&1; echo "rc=$?" | tee -a $L
kill $ALIVE
grep -E "^== |^rc=|passed|failed|^E  |s call|s setup" $L | cut -c1-300 | tail -40
