mkdir -p gpurun_out; cd /root/repo; export TMPDIR=/tmp
python bench.py > gpurun_out/bench_full.log 2>&1
tail -1 gpurun_out/bench_full.log | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('value', d['value'], d['unit'], 'frac', d['roofline']['frac'], 'us', d['roofline']['device_us_per_launch'])
print('cpu', d.get('cpu_baseline'))
print(json.dumps(d.get('extras'), indent=1))
"
