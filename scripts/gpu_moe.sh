cd /root/repo; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_moe.py -x -q -m gpu > gpurun_out/t.log 2>&1; grep -E "passed|failed|Error|^E " gpurun_out/t.log | head -30
python - <<'PY'
import json, torch, sys
sys.path.insert(0, '.')
from benchmarks.extras import bench_moe
print(json.dumps(bench_moe(torch.device('cuda', 0)), indent=1))
PY
