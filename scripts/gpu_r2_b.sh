cd /root/repo; export TMPDIR=/tmp
mkdir -p gpurun_out; rm -f gpurun_out/mla_error_triples.jsonl
timeout -k 10 1000 python -m pytest tests/test_hip_comm_ranks.py tests/test_hip_decode_gqa.py tests/test_hip_mla.py tests/test_hip_quant_gemm.py tests/test_hip_group_gemm.py tests/test_hip_moe.py -x -q -m gpu -s > gpurun_out/r2b_tests.log 2>&1
echo "tests rc=$?"; grep -E "passed|failed|^E  |MLA_ERROR_TRIPLE|direct_exchange" gpurun_out/r2b_tests.log | tail -40
# the bench launcher: `--gpus 2` with no torchrun around it must start the ranks itself (gloo: both ranks share the one GPU)
MOJO_BENCH_DIST_BACKEND=gloo timeout -k 10 900 python bench.py --gpus 2 --steps 20 --warmup 3 --extras-deadline 500 > gpurun_out/r2b_bench2.json 2> gpurun_out/r2b_bench2.err
echo "bench --gpus 2 rc=$?"; python - <<'PY'
import json
try:
    d=json.loads([l for l in open('gpurun_out/r2b_bench2.json') if l.startswith('{')][-1])
    print('n_gpus', d['n_gpus'], 'value', d['value'])
    cc=d['extras']['compute_comm_bf16']
    for k,v in cc.items():
        if 'M4096' in k: print(k, {a:(round(b,1) if isinstance(b,float) else b) for a,b in v.items()})
except Exception as e:
    print('parse failed', e); print(open('gpurun_out/r2b_bench2.err').read()[-3000:])
PY
timeout -k 10 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r2b_bench1.json 2> gpurun_out/r2b_bench1.err
echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r2b_bench1.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'])
print({k:round(v['frac_of_hbm_peak'],3) for k,v in d['extras']['MojoPagedDecodeGQA_bf16_other_contexts'].items()})
PY
