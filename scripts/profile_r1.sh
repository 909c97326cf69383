# Round-1 profiling pass (run through gpurun).  Outputs under gpurun_out/prof_r1/*; summaries are copied to profiles/.
mkdir -p gpurun_out/prof_r1; cd /root/repo; export TMPDIR=/tmp
B="python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1/decode_stats -- $B > gpurun_out/prof_r1/decode_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_r1/decode_fetch -- $B > gpurun_out/prof_r1/decode_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_r1/decode_write -- $B > gpurun_out/prof_r1/decode_write.log 2>&1
G="python benchmarks/gemm_bench.py --m 16384 --k 4096 --n 28672 --groups 8"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1/gg_stats -- $G > gpurun_out/prof_r1/gg_stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof_r1/gg_pmc -- $G > gpurun_out/prof_r1/gg_pmc.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_r1/gg_fetch -- $G > gpurun_out/prof_r1/gg_fetch.log 2>&1
ls -R gpurun_out/prof_r1 | head -40
