# Fuzz soak on the current binary: tests/test_hip_fuzz.py + the decode-fusion random shapes at seed offsets $1 .. $2
cd /root/repo; export PYTHONUNBUFFERED=1
mkdir -p gpurun_out; L=gpurun_out/r5_fuzz_soak.log; : > $L
for off in $(seq $1 $2); do
  MOJO_FUZZ_OFFSET=$off timeout -k 10 300 python -u -m pytest tests/test_hip_fuzz.py tests/test_hip_gemm_skinny.py -q -m gpu -p no:cacheprovider -k "fuzz or random" 2>&1 | tail -1 | sed "s/^/offset $off: /" | tee -a $L
done
