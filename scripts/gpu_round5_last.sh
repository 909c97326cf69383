# Round 5, last GPU pass on the final binary: fuzz soak at fresh seed offsets, then the profiling pass (scripts/profile_r5.sh).
cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out
bash scripts/gpu_fuzz_soak.sh ${1:-600} ${2:-615} && cp gpurun_out/r5_fuzz_soak.log gpurun_out/r5_fuzz_soak_final.log &&
bash scripts/profile_r5.sh > gpurun_out/profile_r5.log 2>&1; echo "profile rc=$?"; tail -5 gpurun_out/profile_r5.log
du -sh gpurun_out/prof_r5 | tail -1
