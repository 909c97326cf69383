# Round 5, final pass: the default bench (N = 1, full extras) and smoke().
cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out; L=gpurun_out/round5_final.log; : > $L
( while true; do sleep 60; echo "[alive $(date +%T)] $(tail -c 120 $L | tr '\n' ' ')" >> gpurun_out/round5_alive.log; done ) &
ALIVE=$!
echo "== smoke" | tee -a $L
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a $L
echo "== bench N=1" | tee -a $L
T0=$(date +%s); timeout -k 10 900 python -u bench.py > gpurun_out/r5_bench.out 2>> $L; echo "rc=$? wall=$(( $(date +%s) - T0 )) s" | tee -a $L
cp bench_extras.json gpurun_out/r5_bench_extras.json 2>/dev/null
tail -1 gpurun_out/r5_bench.out > gpurun_out/r5_bench_compact_line.json

tail -c 1500 gpurun_out/r5_bench_compact_line.json; echo; grep -E "^== |^rc=|smoke" $L
# the N > 1 control flow once more (2 ranks over gloo on the one GPU; single-GPU extras skipped at N > 1)
echo "== bench N=2 over gloo (control flow)" | tee -a $L
T0=$(date +%s); MOJO_HIP_PEER_BLOCKS=16 MOJO_BENCH_COMM_INPROC=1 MOJO_BENCH_DIST_BACKEND=gloo timeout -k 10 400 python -u bench.py --gpus 2 --steps 20 --warmup 3 --extras-deadline 300 > gpurun_out/r5_bench_g2_gloo.out 2>> $L; echo "rc=$? wall=$(( $(date +%s) - T0 )) s" | tee -a $L
tail -1 gpurun_out/r5_bench_g2_gloo.out | tail -c 1200; echo
kill $ALIVE
