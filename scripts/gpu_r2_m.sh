cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_hip_determinism.py -q -m gpu > gpurun_out/r2m_tests.log 2>&1; echo "tests rc=$?"
tail -25 gpurun_out/r2m_tests.log | cut -c1-200
