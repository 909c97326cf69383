cd /root/repo; export TMPDIR=/tmp
python -m pytest tests/test_hip_mla.py tests/test_hip_group_gemm.py tests/test_hip_quant_gemm.py tests/test_hip_comm.py -x -q -m gpu > gpurun_out/t.log 2>&1; grep -E "passed|failed|Error" gpurun_out/t.log | tail -5
python benchmarks/mla_bench.py
