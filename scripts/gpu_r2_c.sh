cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out
echo "== chunks" > gpurun_out/r2c.log
timeout -k 10 400 python -u -m pytest tests/test_hip_comm_ranks.py::test_hip_compute_comm_two_ranks_rccl_pipeline_layout -x -q -m gpu -s >> gpurun_out/r2c.log 2>&1
echo "chunks rc=$?" | tee -a gpurun_out/r2c.log
grep -E "passed|failed|^E  " gpurun_out/r2c.log | tail -5
echo "== direct" >> gpurun_out/r2c.log
MOJO_HIP_PEER_TIMEOUT_MS=3000 timeout -k 10 300 python -u -m pytest tests/test_hip_comm_ranks.py::test_hip_compute_comm_two_ranks_direct_peer_exchange -x -q -m gpu -s >> gpurun_out/r2c.log 2>&1
echo "direct rc=$?" | tee -a gpurun_out/r2c.log
grep -E "passed|failed|^E  |direct_exchange|Error|error" gpurun_out/r2c.log | tail -30
