mkdir -p gpurun_out; cd /root/repo
python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/pytest2.log
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1_decode -- python bench.py --steps 50 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/bench_prof.log 2>&1
for c in 64 128 256 512 1024; do MOJO_HIP_DECODE_CHUNK=$c python bench.py --steps 100 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print($c, d['roofline']['device_us_per_launch'], d['roofline']['frac'])" >> gpurun_out/chunk_sweep.log; done
cat gpurun_out/pytest2.log gpurun_out/chunk_sweep.log; find gpurun_out/prof_r1_decode -name "*stats*" | head
