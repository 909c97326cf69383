cd /root/repo; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_quantizers.py -x -q -m gpu > gpurun_out/t.log 2>&1; grep -E "passed|failed|Error|^E " gpurun_out/t.log | head -30
python - <<'PY'
import json, torch, sys
sys.path.insert(0, '.')
from benchmarks.extras import bench_streaming
for k, v in bench_streaming(torch.device('cuda', 0)).items():
    if 'quant' in k: print(k, round(v["us"], 1), "us", round(v["frac_of_hbm_peak"], 3))
PY
