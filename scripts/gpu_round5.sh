# Round 5 GPU pass: default bench (N = 1, full extras), 4-rank control-flow dry run over gloo on the one GPU (GEMM + collective
# cases inside the rank processes: the pool's process guard allows 6 GPU processes, so 8 ranks or 4 ranks + 4 children cannot run here).
cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out; L=gpurun_out/round5.log; : > $L
( while true; do sleep 60; echo "[alive $(date +%T)] $(tail -c 120 $L | tr '\n' ' ')" >> gpurun_out/round5_alive.log; done ) &
ALIVE=$!
echo "== bench N=1" | tee -a $L
timeout -k 10 500 python -u bench.py > gpurun_out/r5_bench.out 2>> $L; echo "rc=$?" | tee -a $L
cp bench_extras.json gpurun_out/r5_bench_extras.json 2>/dev/null
tail -1 gpurun_out/r5_bench.out > gpurun_out/r5_bench_compact_line.json
echo "== 4 ranks over gloo on one GPU (control flow only)" | tee -a $L
MOJO_HIP_PEER_BLOCKS=16 MOJO_BENCH_COMM_INPROC=1 MOJO_BENCH_DIST_BACKEND=gloo timeout -k 10 600 python -u bench.py --gpus 4 --steps 20 --warmup 3 --extras-deadline 500 > gpurun_out/r5_bench_g4_gloo.out 2>> $L; echo "rc=$?" | tee -a $L
cp bench_extras.json gpurun_out/r5_bench_g4_gloo_extras.json 2>/dev/null
tail -1 gpurun_out/r5_bench_g4_gloo.out > gpurun_out/r5_bench_g4_gloo_compact_line.json
kill $ALIVE
tail -c 1500 gpurun_out/r5_bench_compact_line.json; echo; tail -c 1500 gpurun_out/r5_bench_g4_gloo_compact_line.json; echo; grep -E "^== |^rc=" $L
