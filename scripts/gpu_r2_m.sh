cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out; L=gpurun_out/r2m.log; : > $L
echo "== default" >> $L
timeout -k 10 300 python -u scripts/probes/decompress_gemm.py >> $L 2>&1; echo "rc=$?" >> $L
echo "== persistent" >> $L
MOJO_HIP_GEMM_PERSIST=1 timeout -k 10 300 python -u scripts/probes/decompress_gemm.py >> $L 2>&1; echo "rc=$?" >> $L
grep -v amdgpu.ids $L
