# Same-box A/B of the K-loop read placement: the in-tree library (A0 of the next K-tile read in P4) against a build with
# -DGEMM_A0_IN_P1 (round 1's placement) made on the box into a scratch copy of the package.
cd /root/repo; export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out /tmp/alt && cp -r mojo_opset_amd /tmp/alt/ && cp -r include /tmp/alt/
( cd /tmp/alt && MOJO_HIP_EXTRA_CXXFLAGS=-DGEMM_A0_IN_P1 python -m mojo_opset_amd.csrc.build --force -j 16 > /root/repo/gpurun_out/alt_build.log 2>&1 ); echo alt build rc=$?
L=gpurun_out/gemm_ab.log; : > $L
for i in 1 2 3; do
  for SH in "--m 16384 --k 4096 --n 28672 --groups 8" "--m 16384 --k 4096 --n 28672 --groups 8 --trans" "--m 16384 --k 14336 --n 4096 --groups 8" "--m 10240 --k 512 --n 32768 --groups 1 --trans"; do
    echo "new $SH" >> $L; timeout -k 10 120 python -u benchmarks/gemm_bench.py $SH >> $L 2>&1
    echo "old $SH" >> $L; MOJO_HIP_LIB=/tmp/alt/mojo_opset_amd/lib/libmojo_hip.so timeout -k 10 120 python -u benchmarks/gemm_bench.py $SH >> $L 2>&1
  done
done
python - <<'PY'
import json
cur=None;res={}
for l in open('gpurun_out/gemm_ab.log'):
    if l.startswith(('new ','old ')): cur=l.strip()
    elif l.startswith('{"us"') and cur: res.setdefault(cur,[]).append(round(json.loads(l)['tflops']))
for k in sorted(res, key=lambda x:(x.split(' ',1)[1], x)): print(k,res[k])
PY
