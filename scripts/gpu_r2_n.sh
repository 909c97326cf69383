cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out; rm -f gpurun_out/comm_ranks_progress*.log gpurun_out/r2n_alive.log
L=gpurun_out/r2n.log; : > $L
( while true; do sleep 60; echo "[alive $(date +%T)] $(tail -c 120 $L | tr '\n' ' ')" >> gpurun_out/r2n_alive.log; done ) &
ALIVE=$!
echo "== dispatch probe" | tee -a $L
timeout -k 10 120 scripts/probes/build/dispatch_rate > gpurun_out/r2_dispatch_placement.txt 2>&1; echo "rc=$?" | tee -a $L
echo "== suite" | tee -a $L
MOJO_HIP_PEER_TIMEOUT_MS=8000 timeout -k 10 900 python -u -m pytest tests -q -m gpu --durations=10 >> $L 2>&1; echo "rc=$?" | tee -a $L
echo "== bench" | tee -a $L
timeout -k 10 400 python -u bench.py > gpurun_out/r2n_bench.json 2>> $L; echo "rc=$?" | tee -a $L
kill $ALIVE
grep -E "^== |^rc=|passed|failed|^E  |s call|s setup" $L | cut -c1-300 | tail -30
