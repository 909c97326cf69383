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
kill $ALIVE
tail -c 1500 gpurun_out/r5_bench_compact_line.json; echo; grep -E "^== |^rc=|smoke" $L
