cd /root/repo; export TMPDIR=/tmp
python -m pytest tests/test_hip_mla.py -x -q -m gpu > gpurun_out/t.log 2>&1; grep -E "passed|failed|Error" gpurun_out/t.log | tail -5
python benchmarks/mla_bench.py
for a in 0 1 2 3; do
  MOJO_HIP_MLA_ABLATE=$a rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl_$a -- python benchmarks/mla_bench.py > /dev/null 2>&1
  echo "ABL=$a $(grep mla512 gpurun_out/abl_$a/*/*kernel_stats.csv | cut -d, -f1,4 | cut -c1-40,60-)"
done
