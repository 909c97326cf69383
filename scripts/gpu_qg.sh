cd /root/repo; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_quant_gemm.py -x -q -m gpu > gpurun_out/t.log 2>&1; grep -E "passed|failed|Error|^E " gpurun_out/t.log | head
python - <<'PY'
import json, torch, sys
sys.path.insert(0, '.')
from benchmarks.extras import bench_quant_gemm
r = bench_quant_gemm(torch.device('cuda', 0))
for k, v in r.items(): print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
PY
