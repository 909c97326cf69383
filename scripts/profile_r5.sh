# Round-5 profiling pass (run through gpurun; ~10 min).  Raw rocprofv3 output under gpurun_out/prof_r5/*; scripts/summarize_r5.py
# writes the committed summaries under profiles/.  Counter passes (--pmc) are separate runs with --kernel-trace only, one counter
# per pass (FETCH_SIZE takes 3 of the 4 TCC slots) — MI355X_MICROARCH.md, HBM / rocprofv3 section.
cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=gpurun_out/prof_r5; rm -rf $P; mkdir -p $P
B="python3 bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $P/decode_stats -- $B > $P/decode_stats.log 2>&1; echo decode_stats rc=$?
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/decode_fetch -- $B > $P/decode_fetch.log 2>&1; echo decode_fetch rc=$?
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/decode_write -- $B > $P/decode_write.log 2>&1; echo decode_write rc=$?
# one case per process: every mojo:: row of a trace belongs to that case
one() {  # tag, bench function, case substring
  MOJO_BENCH_ONLY="$3" rocprofv3 --kernel-trace --stats --output-format csv -d $P/$1_stats -- python3 benchmarks/one.py $2 > $P/$1_stats.log 2>&1; echo $1 stats rc=$?
  MOJO_BENCH_ONLY="$3" rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/$1_fetch -- python3 benchmarks/one.py $2 > $P/$1_fetch.log 2>&1; echo $1 fetch rc=$?
  MOJO_BENCH_ONLY="$3" rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/$1_write -- python3 benchmarks/one.py $2 > $P/$1_write.log 2>&1; echo $1 write rc=$?
}
one mla_decode bench_mla_decode B64_H128
one prefill_gqa_4x2048 bench_prefill 4x2048_nocache
one quant_gemm_fp8_4096x7168x36864 bench_quant_gemm fp8_e4m3_4096x7168x36864
# the other cases: durations only
dur() {
  MOJO_BENCH_ONLY="$3" rocprofv3 --kernel-trace --output-format csv -d $P/case_$1 -- python3 benchmarks/one.py $2 > $P/case_$1.log 2>&1; echo case_$1 rc=$?
}
dur pf_ragged bench_prefill 16_ragged
dur pf_16k bench_prefill 1x16384
dur pf_chunked bench_prefill chunked_1x512
dur dec_1024 bench_decode_variants uniform_ctx1024
dur dec_ragged bench_decode_variants ragged_ctx2048_4096
dur dec_g8 bench_decode_geometries G8_llama3_70b
dur dec_d64 bench_decode_geometries G4_32q_8kv_d64
dur mlapf_nocache bench_mla_prefill 4x512_nocache
dur qg_int8_m32 bench_quant_gemm int8_32x7168x4096
python3 scripts/summarize_r5.py
