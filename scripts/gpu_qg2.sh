cd /root/repo; export TMPDIR=/tmp; rm -rf gpurun_out/qg_prof
for sk in 0 2 7 14 28; do
  if [ $sk = 0 ]; then unset MOJO_HIP_QGEMM_SPLITK; else export MOJO_HIP_QGEMM_SPLITK=$sk; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/qg_prof/sk$sk -- python scripts/probes/qg_small.py > /dev/null 2>&1
  echo "splitk=$sk"; grep -h "quant_skinny\|quant_finalize" gpurun_out/qg_prof/sk$sk/*/*kernel_stats.csv | awk -F'",' '{print "   ", substr($1,1,46), $2}' | cut -c1-110
done
