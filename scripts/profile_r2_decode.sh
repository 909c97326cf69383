# Refresh of the decode passes of scripts/profile_r2.sh (the decode kernel changed after the full pass: paired workgroups).
cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=gpurun_out/prof_r2; mkdir -p $P; rm -rf $P/decode_stats $P/decode_fetch $P/decode_write $P/case_dec_ragged $P/case_dec_1024 $P/case_mlapf_nocache $P/case_mlapf_cached
B="python3 bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $P/decode_stats -- $B > $P/decode_stats.log 2>&1; echo decode_stats rc=$?
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/decode_fetch -- $B > $P/decode_fetch.log 2>&1; echo decode_fetch rc=$?
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/decode_write -- $B > $P/decode_write.log 2>&1; echo decode_write rc=$?
one() { MOJO_BENCH_ONLY="$3" rocprofv3 --kernel-trace --output-format csv -d $P/case_$1 -- python3 benchmarks/one.py $2 > $P/case_$1.log 2>&1; echo case_$1 rc=$?; }
one dec_ragged bench_decode_variants ragged_ctx2048_4096
one dec_ragged16k bench_decode_variants ragged_ctx8192_16384
one dec_1024 bench_decode_variants uniform_ctx1024
one mlapf_nocache bench_mla_prefill 4x512_nocache
one mlapf_cached bench_mla_prefill 4x512_cached2048
