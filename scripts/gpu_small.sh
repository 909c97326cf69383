cd /root/repo; export TMPDIR=/tmp
python -m pytest tests/test_hip_gemm_skinny.py tests/test_hip_group_gemm.py tests/test_hip_comm.py tests/test_hip_moe.py -q -m gpu > gpurun_out/t.log 2>&1; grep -E "passed|failed|Error|^E " gpurun_out/t.log | tail -8
echo skinny; python scripts/probes/dense_small.py
echo tile256; MOJO_HIP_GEMM_SKINNY=0 python scripts/probes/dense_small.py
