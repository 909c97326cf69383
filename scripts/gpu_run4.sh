mkdir -p gpurun_out; cd /root/repo
timeout 900 python -m pytest tests/test_hip_group_gemm.py -m gpu -q -x 2>&1 | tail -30 > gpurun_out/pytest4.log
timeout 600 python -c "
import torch, json, sys
sys.path.insert(0,'.')
from benchmarks.extras import run_extras
print(json.dumps(run_extras(torch.device('cuda',0),1), indent=1))
" > gpurun_out/gg_bench.log 2>&1
cat gpurun_out/pytest4.log gpurun_out/gg_bench.log
