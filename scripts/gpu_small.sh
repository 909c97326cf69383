cd /root/repo; export TMPDIR=/tmp
python -m pytest tests/test_hip_gemm_skinny.py tests/test_hip_group_gemm.py tests/test_hip_moe.py tests/test_hip_graph.py -q -m gpu > gpurun_out/t.log 2>&1; grep -E "passed|failed|Error|^E " gpurun_out/t.log | tail -8
python - <<'PY'
import json, torch, sys, os
sys.path.insert(0, '.')
from benchmarks.extras import bench_moe
r = bench_moe(torch.device('cuda', 0))
print("ragged", {k: round(v["us"], 1) for k, v in r.items()}, r["moe_layer_decode_T64_E64_k8_H4096_I2048"])
PY
