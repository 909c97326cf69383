cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out
L=gpurun_out/r2i.log; : > $L
run() { echo "== $1" | tee -a $L; shift; timeout -k 10 "$@" >> $L 2>&1; echo "rc=$?" | tee -a $L; }
run gemm_tests 500 python -u -m pytest tests/test_hip_group_gemm.py tests/test_hip_moe.py tests/test_hip_gemm_skinny.py tests/test_hip_comm.py -x -q -m gpu
for P in 0 1 0 1; do
  for SH in "--m 16384 --k 4096 --n 28672 --groups 8" "--m 16384 --k 4096 --n 28672 --groups 8 --trans" "--m 16384 --k 14336 --n 4096 --groups 8" "--m 20480 --k 4096 --n 4096 --groups 8" "--m 4096 --k 4096 --n 28672 --groups 8" "--m 10240 --k 512 --n 32768 --groups 1 --trans" "--m 16384 --k 4096 --n 28672 --groups 8 --split skewed"; do
    echo "persist=$P $SH" >> $L
    MOJO_HIP_GEMM_PERSIST=$P timeout -k 10 120 python -u benchmarks/gemm_bench.py $SH >> $L 2>&1
  done
done
grep -E "^== |^rc=|passed|failed|^E  |persist=|tflops" $L | cut -c1-200 | tail -70
