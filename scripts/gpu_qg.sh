export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_hip_quant_gemm.py -m gpu -q -x 2>&1 | tail -3 || exit 1
python - <<'PY'
import sys, torch
sys.path.insert(0, '.')
from benchmarks.extras import bench_quant_gemm
for k, v in bench_quant_gemm(torch.device("cuda", 0)).items(): print(k, {a: round(b, 3) for a, b in v.items()})
PY
