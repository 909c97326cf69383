cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out
L=gpurun_out/r2k.log; : > $L
run() { echo "== $1" | tee -a $L; shift; timeout -k 10 "$@" >> $L 2>&1; echo "rc=$?" | tee -a $L; }
run decode_tests 400 python -u -m pytest tests/test_hip_decode_gqa.py tests/test_hip_graph.py tests/test_hip_paged_cache.py -x -q -m gpu
run bench_pair 300 python -u benchmarks/one.py bench_decode_variants
MOJO_HIP_DECODE_PAIR=0 run bench_nopair 300 python -u benchmarks/one.py bench_decode_variants
run headline 200 python -u bench.py --no-extras --no-cpu-baseline
MOJO_HIP_DECODE_PAIR=0 run headline_nopair 200 python -u bench.py --no-extras --no-cpu-baseline
grep -E "^== |^rc=|passed|failed|^E  " $L | tail -20
python - <<'PY'
import json
for l in open('gpurun_out/r2k.log'):
    if l.startswith('{"bench_decode_variants"'):
        d=json.loads(l)['bench_decode_variants']; print({k:(round(v['us'],1), round(v['frac_of_hbm_peak'],3)) for k,v in d.items()})
    if l.startswith('{"metric"'):
        d=json.loads(l); print('headline', round(d['value']), round(d['roofline']['frac'],4), round(d['roofline']['device_us_per_launch'],1))
PY
