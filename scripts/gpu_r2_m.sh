cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out; L=gpurun_out/r2m.log; : > $L
echo "== default" | tee -a $L
for i in 1 2; do timeout -k 10 200 python -u benchmarks/one.py bench_mla_decode >> $L 2>&1; done
echo "== DMA front" | tee -a $L
touch mojo_opset_amd/csrc/mla_attn.hip
MOJO_HIP_EXTRA_CXXFLAGS=-DMLA_DMA_FRONT timeout -k 10 600 python -m mojo_opset_amd.csrc.build -j 8 > gpurun_out/r2m_build.log 2>&1; echo "build rc=$?" | tee -a $L
for i in 1 2; do timeout -k 10 200 python -u benchmarks/one.py bench_mla_decode >> $L 2>&1; done
timeout -k 10 300 python -u -m pytest tests/test_hip_mla.py -q -m gpu -x -k "decode" 2>&1 | tail -1 | tee -a $L
grep -E "^==|bench_mla|passed|failed|rc=" $L | cut -c1-200
