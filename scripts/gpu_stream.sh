cd /root/repo; export TMPDIR=/tmp
python -m pytest tests/test_hip_streaming.py -x -q -m gpu > gpurun_out/t.log 2>&1; grep -E "passed|failed|Error" gpurun_out/t.log | tail -5
python - <<'PY'
import json, torch, sys
sys.path.insert(0, '.')
from benchmarks.extras import bench_streaming
for k, v in bench_streaming(torch.device('cuda', 0)).items(): print(k, round(v["us"], 1), "us", round(v["frac_of_hbm_peak"], 3))
PY
