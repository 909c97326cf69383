# One GPU-box pass of the round: the whole GPU suite, the default bench line and the dispatcher probe.
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash scripts/gpu_round.sh'
# Progress goes to gpurun_out/ (a silent command is taken for hung after 7 minutes).
cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out; rm -f gpurun_out/comm_ranks_progress*.log gpurun_out/round_alive.log
L=gpurun_out/round.log; : > $L
( while true; do sleep 60; echo "[alive $(date +%T)] $(tail -c 120 $L | tr '\n' ' ')" >> gpurun_out/round_alive.log; done ) &
ALIVE=$!
echo "== suite" | tee -a $L
MOJO_HIP_PEER_TIMEOUT_MS=8000 timeout -k 10 900 python -u -m pytest tests -q -m gpu --durations=10 >> $L 2>&1; echo "rc=$?" | tee -a $L
echo "== bench" | tee -a $L
timeout -k 10 400 python -u bench.py > gpurun_out/round_bench.json 2>> $L; echo "rc=$?" | tee -a $L
echo "== two ranks, control flow over gloo (one GPU)" | tee -a $L
MOJO_BENCH_DIST_BACKEND=gloo timeout -k 10 600 python -u bench.py --gpus 2 --steps 20 --warmup 3 > gpurun_out/round_bench_2ranks.json 2>> $L; echo "rc=$?" | tee -a $L
echo "== dispatcher placement probe" | tee -a $L
[ -x scripts/probes/build/dispatch_rate ] && timeout -k 10 120 scripts/probes/build/dispatch_rate > gpurun_out/r2_dispatch_placement.txt 2>&1; echo "rc=$?" | tee -a $L
kill $ALIVE
grep -E "^== |^rc=|passed|failed|^E  |s call" $L | cut -c1-300 | tail -30
