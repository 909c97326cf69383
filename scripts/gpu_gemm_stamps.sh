cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out
MOJO_HIP_EXTRA_CXXFLAGS=-DGEMM_STAMPS python -m mojo_opset_amd.csrc.build --force -j 16 > gpurun_out/stamp_build.log 2>&1; echo build rc=$?
python -u scripts/probes/gemm_stamps.py > gpurun_out/gemm_stamps_kn.txt 2>&1; echo kn rc=$?
python -u scripts/probes/gemm_stamps.py nk > gpurun_out/gemm_stamps_nk.txt 2>&1; echo nk rc=$?
cat gpurun_out/gemm_stamps_kn.txt gpurun_out/gemm_stamps_nk.txt | tail -40
