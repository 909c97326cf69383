cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out; L=gpurun_out/r2k.log; : > $L
echo "== peer ranks" | tee -a $L
MOJO_HIP_PEER_TIMEOUT_MS=8000 timeout -k 10 500 python -u -m pytest tests/test_hip_comm_ranks.py -q -m gpu -x >> $L 2>&1; echo "rc=$?" | tee -a $L
for c in default 128 512 1024; do
  echo "== decode ctx1024 chunk=$c" | tee -a $L
  if [ $c = default ]; then unset MOJO_HIP_DECODE_CHUNK; else export MOJO_HIP_DECODE_CHUNK=$c; fi
  MOJO_BENCH_ONLY=uniform_ctx1024 timeout -k 10 200 python -u benchmarks/one.py bench_decode_variants >> $L 2>&1 || echo "rc=$?" | tee -a $L
done
unset MOJO_HIP_DECODE_CHUNK
echo "== pair off" | tee -a $L
MOJO_HIP_DECODE_PAIR=0 MOJO_BENCH_ONLY=uniform_ctx1024 timeout -k 10 200 python -u benchmarks/one.py bench_decode_variants >> $L 2>&1
grep -E "^== |^rc=|passed|failed|^E  |uniform" $L | cut -c1-400
