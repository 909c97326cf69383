cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=gpurun_out/prof_ab; rm -rf $P; mkdir -p $P
B="python3 bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline"
$B > $P/live_pair1.json 2>/dev/null
MOJO_HIP_DECODE_PAIR=0 $B > $P/live_pair0.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $P/pair1 -- $B > $P/pair1.log 2>&1
export MOJO_HIP_DECODE_PAIR=0
rocprofv3 --kernel-trace --stats --output-format csv -d $P/pair0 -- $B > $P/pair0.log 2>&1
unset MOJO_HIP_DECODE_PAIR
$B > $P/live_pair1_b.json 2>/dev/null
python3 - <<'PY'
import json,glob,csv
for n in ('live_pair1','live_pair0','live_pair1_b'):
    d=json.loads(open(f'gpurun_out/prof_ab/{n}.json').read().strip().splitlines()[-1]); print(n, round(d['roofline']['device_us_per_launch'],2), round(d['roofline']['frac'],4))
for n in ('pair1','pair0'):
    f=glob.glob(f'gpurun_out/prof_ab/{n}/**/*kernel_trace.csv', recursive=True)[0]
    rows=sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])-int(r['Start_Timestamp'])) for r in csv.DictReader(open(f)) if 'decode_split' in r['Kernel_Name'])
    d=[x[1] for x in rows][-200:]
    print(n, 'profiled avg us', sum(d)/len(d)/1e3, 'min', min(d)/1e3)
PY
