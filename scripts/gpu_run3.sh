mkdir -p gpurun_out; cd /root/repo
python -m pytest tests/test_hip_decode_gqa.py -m gpu -q 2>&1 | tail -5 > gpurun_out/pytest3.log
rm -f gpurun_out/chunk_sweep3.log
for nt in 0 1; do for c in 0 256 512 1024 2048 4096; do
  if [ $c = 0 ]; then unset MOJO_HIP_DECODE_CHUNK; else export MOJO_HIP_DECODE_CHUNK=$c; fi
  MOJO_HIP_DECODE_NT=$nt python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('nt=$nt chunk=$c', round(d['roofline']['device_us_per_launch'],1), round(d['roofline']['frac'],4), round(d['ms_per_step']*1000,1))" >> gpurun_out/chunk_sweep3.log; done; done
cat gpurun_out/pytest3.log gpurun_out/chunk_sweep3.log
