cd /root/repo; export TMPDIR=/tmp
python - <<'PY'
import json, torch, sys
sys.path.insert(0, '.')
from benchmarks.extras import bench_moe
print(json.dumps(bench_moe(torch.device('cuda', 0)), indent=1))
PY
