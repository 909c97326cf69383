mkdir -p gpurun_out; cd /root/repo; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_group_gemm.py -m gpu -q -x 2>&1 | tail -5 > gpurun_out/pytest5.log
for t in "" "--trans"; do
  tag=nn; [ -n "$t" ] && tag=nt
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc_gemm_${tag}_a -- python benchmarks/gemm_bench.py --m 8192 $t > gpurun_out/pmc_${tag}_a.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL --output-format csv -d gpurun_out/pmc_gemm_${tag}_b -- python benchmarks/gemm_bench.py --m 8192 $t > gpurun_out/pmc_${tag}_b.log 2>&1
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d gpurun_out/pmc_gemm_${tag}_c -- python benchmarks/gemm_bench.py --m 8192 $t > gpurun_out/pmc_${tag}_c.log 2>&1
done
cat gpurun_out/pytest5.log; tail -2 gpurun_out/pmc_*_a.log; ls gpurun_out/pmc_gemm_nn_a/*/ | head
