# Round 5: counters of the 128-row-tile GEMM (csrc/gemm_tile128_core.h) on the final binary — MFMA busy + clock, LDS activity and
# bank conflicts, HBM traffic, wave-cycle split — one counter group per pass (--pmc with --kernel-trace only).
#   -> gpurun_out/prof_r5_tile128 ; summarised by scripts/summarize_r5_tile128.py into profiles/r5_gemm_tile128_counters.json
cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=gpurun_out/prof_r5_tile128; rm -rf $P; mkdir -p $P
python3 scripts/probes/tile128_counters.py wall > $P/wall.log 2>&1; echo wall rc=$?; tail -1 $P/wall.log
for c in d1024 d1024c d2048w d256sk q1024; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/f_$c -- python3 scripts/probes/tile128_counters.py one $c > $P/f_$c.log 2>&1; echo fetch $c rc=$?
  rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $P/w_$c -- python3 scripts/probes/tile128_counters.py one $c > $P/w_$c.log 2>&1; echo write $c rc=$?
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $P/c_$c -- python3 scripts/probes/tile128_counters.py one $c > $P/c_$c.log 2>&1; echo clock $c rc=$?
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $P/l_$c -- python3 scripts/probes/tile128_counters.py one $c > $P/l_$c.log 2>&1; echo lds $c rc=$?
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $P/s_$c -- python3 scripts/probes/tile128_counters.py one $c > $P/s_$c.log 2>&1; echo waits $c rc=$?
done
python3 scripts/summarize_r5_tile128.py gpurun_out > $P/summary.log 2>&1; tail -3 $P/summary.log
