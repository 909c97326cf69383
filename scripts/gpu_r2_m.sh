cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out
touch mojo_opset_amd/csrc/mla_attn.hip
MOJO_HIP_EXTRA_CXXFLAGS=-DMLA_STAMPS timeout -k 10 600 python -m mojo_opset_amd.csrc.build -j 8 > gpurun_out/r2m_build.log 2>&1 || { echo build failed; tail -20 gpurun_out/r2m_build.log; }
timeout -k 10 200 python -u scripts/probes/mla_pp_stamps.py 2>&1 | grep -v amdgpu.ids | cut -c1-200
