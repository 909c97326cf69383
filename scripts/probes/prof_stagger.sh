cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=gpurun_out/prof_stagger; rm -rf $P; mkdir -p $P
for c in 0 2048; do for st in 0 150; do
  CACHED=$c MOJO_HIP_GEMM_STAGGER=$st rocprofv3 --kernel-trace --stats --output-format csv -d $P/c${c}_st$st -- python3 scripts/probes/mla_prefill_one.py > $P/c${c}_st$st.log 2>&1
  f=$(find $P/c${c}_st$st -name "*kernel_stats.csv" | head -1); echo "== cached $c stagger $st: $(tail -1 $P/c${c}_st$st.log)"; cut -d, -f1-4 $f | head -4 | cut -c1-160
done; done
