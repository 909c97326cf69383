cd /root/repo; export TMPDIR=/tmp
python -m pytest tests/test_hip_mla.py -x -q -m gpu > gpurun_out/t.log 2>&1; grep -E "passed|failed|Error" gpurun_out/t.log | tail -5
python benchmarks/mla_bench.py
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/mla_prof -- python benchmarks/mla_bench.py > /dev/null 2>&1
