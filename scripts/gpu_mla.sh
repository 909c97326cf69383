cd /root/repo; export TMPDIR=/tmp
python -m pytest tests/test_hip_gemm_skinny.py tests/test_hip_mla.py tests/test_hip_graph.py tests/test_hip_group_gemm.py tests/test_hip_comm.py -x -q -m gpu > gpurun_out/t.log 2>&1; grep -E "passed|failed|Error|^E " gpurun_out/t.log | tail -8
python benchmarks/mla_bench.py
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/mla_prof -- python benchmarks/mla_bench.py > /dev/null 2>&1
