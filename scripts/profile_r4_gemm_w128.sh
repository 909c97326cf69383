# GroupGemm 8-wave (shipped) vs 4-wave 128x128-per-wave experiment: wall A/B + clock / MFMA busy per arm -> gpurun_out/prof_r4_w128
# (experiments build: MOJO_HIP_BUILD_EXPERIMENTS=1 python -m mojo_opset_amd.csrc.build); summarised by scripts/summarize_r4_w128.py
cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=gpurun_out/prof_r4_w128; rm -rf $P; mkdir -p $P
./scripts/probes/mfma16_issue_probe.bin > $P/mfma_issue_probe.txt 2>&1; echo probe rc=$?
python3 scripts/probes/gemm_w128_ab.py > $P/ab.log 2>&1; echo ab rc=$?; tail -1 $P/ab.log
for o in 0 1; do
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $P/c$o -- python3 scripts/probes/gemm_w128_ab.py one $o > $P/c$o.log 2>&1; echo clock $o rc=$?
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $P/l$o -- python3 scripts/probes/gemm_w128_ab.py one $o > $P/l$o.log 2>&1; echo lds $o rc=$?
done
