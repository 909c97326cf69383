cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out
# the driver's own N=2 invocation form (bench.py starts its ranks itself), collectives over gloo since the box has one GPU
MOJO_BENCH_DIST_BACKEND=gloo timeout -k 10 800 python -u bench.py --gpus 2 --steps 20 --warmup 3 > gpurun_out/r2m_dry.json 2> gpurun_out/r2m_dry.err; echo "rc=$?"
python - <<'PY'
import json
lines=[l for l in open('gpurun_out/r2m_dry.json') if l.startswith('{')]
print(len(lines),'json lines')
d=json.loads(lines[-1])
print({k:d[k] for k in ('value','n_gpus','ms_per_step','scaling')})
ex=d['extras']; print(list(ex.keys()) if isinstance(ex,dict) else ex)
cc=ex.get('compute_comm_bf16',{})
print(len(cc),'comm cases'); 
for k,v in list(cc.items())[:40]: print(' ',k,{a:(round(b,1) if isinstance(b,float) else b) for a,b in v.items() if a in ('us','error','exposed_exchange_us','link_GB/s')})
PY
tail -5 gpurun_out/r2m_dry.err | cut -c1-300
