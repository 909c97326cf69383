# Round-2 profiling pass (run through gpurun; ~12 min).  Raw rocprofv3 output under gpurun_out/prof_r2/*; the summaries
# scripts/summarize_r2.py writes under profiles/ are what is committed.  Counter passes (--pmc) are separate runs with
# --kernel-trace only, one counter group per pass (MI355X_MICROARCH.md, HBM / rocprofv3 section).
cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=gpurun_out/prof_r2; mkdir -p $P
B="python3 bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $P/decode_stats -- $B > $P/decode_stats.log 2>&1; echo decode_stats rc=$?
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/decode_fetch -- $B > $P/decode_fetch.log 2>&1; echo decode_fetch rc=$?
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/decode_write -- $B > $P/decode_write.log 2>&1; echo decode_write rc=$?
G="python3 benchmarks/gemm_bench.py --m 16384 --k 4096 --n 28672 --groups 8"
rocprofv3 --kernel-trace --stats --output-format csv -d $P/gg_stats -- $G > $P/gg_stats.log 2>&1; echo gg_stats rc=$?
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $P/gg_pmc -- $G > $P/gg_pmc.log 2>&1; echo gg_pmc rc=$?
# one case per process: every row of a trace belongs to that case
one() {  # tag, bench function, case substring
  MOJO_BENCH_ONLY="$3" rocprofv3 --kernel-trace --output-format csv -d $P/case_$1 -- python3 benchmarks/one.py $2 > $P/case_$1.log 2>&1; echo case_$1 rc=$?
}
one pf_4x2048_nocache bench_prefill 4x2048_nocache
one pf_4x2048_cached bench_prefill 4x2048_cached2048
one pf_ragged bench_prefill 16_ragged
one pf_16k bench_prefill 1x16384
one mla_decode bench_mla_decode ""
one mlapf_nocache bench_mla_prefill 4x512_nocache
one mlapf_cached bench_mla_prefill 4x512_cached2048
one qg_int8_up bench_quant_gemm int8_4096x7168x36864
one qg_fp8_up bench_quant_gemm fp8_e4m3_4096x7168x36864
one qg_int8_m32 bench_quant_gemm int8_32x7168x4096
one dec_ragged bench_decode_variants ragged_ctx2048_4096
one dec_1024 bench_decode_variants uniform_ctx1024
ls $P | head -40
