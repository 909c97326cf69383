cd /root/repo; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_prefill_gqa.py -x -q -m gpu > gpurun_out/t.log 2>&1; grep -E "passed|failed|Error|^E " gpurun_out/t.log | head
python benchmarks/prefill_bench.py
