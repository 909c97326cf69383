# Round 5: counters of the MFMA-bound GEMM rows on the FINAL binary (Missing 5):
#   MojoGroupGemm bf16 Mixtral up [G,K,N], MojoQuantGemm fp8 / int8 4096 x 7168 x 36864
#   wall (shipped / zero operands / hipBLASLt on the same box) + FETCH_SIZE, WRITE_SIZE + L2 hit, MFMA busy + clock, LDS
#   -> gpurun_out/prof_r5_gemm ; summarised by scripts/summarize_r5_gemm.py into profiles/r5_group_gemm_counters.json and
#   profiles/r5_quant_gemm_counters.json
cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=gpurun_out/prof_r5_gemm; rm -rf $P; mkdir -p $P
python3 scripts/probes/gemm_counters_r5.py wall > $P/wall.log 2>&1; echo wall rc=$?; tail -1 $P/wall.log
for c in gg_kn qg_fp8 qg_i8; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/f_$c -- python3 scripts/probes/gemm_counters_r5.py one $c > $P/f_$c.log 2>&1; echo fetch $c rc=$?
  rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $P/w_$c -- python3 scripts/probes/gemm_counters_r5.py one $c > $P/w_$c.log 2>&1; echo write $c rc=$?
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $P/c_$c -- python3 scripts/probes/gemm_counters_r5.py one $c > $P/c_$c.log 2>&1; echo clock $c rc=$?
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $P/l_$c -- python3 scripts/probes/gemm_counters_r5.py one $c > $P/l_$c.log 2>&1; echo lds $c rc=$?
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $P/s_$c -- python3 scripts/probes/gemm_counters_r5.py one $c > $P/s_$c.log 2>&1; echo waits $c rc=$?
done
python3 scripts/summarize_r5_gemm.py gpurun_out > $P/summary.log 2>&1; tail -5 $P/summary.log
