# prefill ablations (timing only; results are wrong by design)
export TMPDIR=/tmp
for a in 0 1 2 4 3 7; do echo "ablate=$a"; MOJO_HIP_PREFILL_ABLATE=$a python benchmarks/prefill_bench.py | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print({k: round(v['us'], 1) for k, v in d.items()})"; done
