mkdir -p gpurun_out; cd /root/repo
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/pytest_full.log
cat gpurun_out/pytest_full.log
