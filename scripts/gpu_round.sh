# full GPU pass: test suite, smoke, bench, profiling pass (outputs under gpurun_out/)
mkdir -p gpurun_out; cd /root/repo; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/pytest_full.log
cat gpurun_out/pytest_full.log | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash scripts/gpu_bench.sh
rm -rf gpurun_out/prof_r1; bash scripts/profile_r1.sh > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1/mla_stats -- python benchmarks/mla_bench.py > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1/prefill_stats -- python benchmarks/prefill_bench.py > /dev/null 2>&1
