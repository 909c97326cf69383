cd /root/repo; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_graph.py tests/test_hip_moe.py -x -q -m gpu > gpurun_out/t.log 2>&1; grep -E "passed|failed|Error|^E " gpurun_out/t.log | head -30
