cd /root/repo; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_decode_gqa.py tests/test_hip_graph.py -x -q -m gpu > gpurun_out/t.log 2>&1; grep -E "passed|failed|Error|^E " gpurun_out/t.log | head
for f in 1 0; do MOJO_HIP_DECODE_FUSE=$f python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('fuse=$f', round(d[\"value\"]), round(d[\"roofline\"][\"frac\"],4), round(d[\"roofline\"][\"device_us_per_launch\"],1))"; done
