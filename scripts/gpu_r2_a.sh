cd /root/repo; export TMPDIR=/tmp
mkdir -p gpurun_out
# new / changed tests first (fast signal), then the whole suite, then the bench
timeout -k 10 900 python -m pytest tests/test_hip_paged_cache.py tests/test_hip_decode_gqa.py tests/test_hip_graph.py tests/test_hip_streaming.py tests/test_hip_comm_ranks.py::test_hip_compute_comm_two_ranks_rccl_pipeline_layout -x -q -m gpu > gpurun_out/r2a_new.log 2>&1
echo "new tests rc=$?"; tail -25 gpurun_out/r2a_new.log
timeout -k 10 900 python bench.py --steps 100 --warmup 10 > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench.err
echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r2a_bench.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d.get('roofline_group_gemm'))
print(json.dumps(d['extras']['MojoPagedDecodeGQA_bf16_other_contexts'], indent=0))
PY
