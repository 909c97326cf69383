#!/bin/bash
# builds scripts/probes/_bin/t128_<variant> (see tile128_anatomy.hip); variant = a<ablate>p<piece>s<sched>, e.g. a0p1s1.  The
# namespace is renamed per variant so that the probe's kernels do not collide with the library's registration of the same stubs.
set -e
cd "$(dirname "$0")/../.."
mkdir -p scripts/probes/_bin
for v in "$@"; do
  a=${v:1:1}; p=${v:3:1}; s=${v:5:1}   # (p: piece shape of the first runs; the shipped kernel has whole-line pieces only)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DT128_ABLATE=$a -DT128_PIECE=$p -DT128_SCHED=$s -Dg128=g128_probe_$v -Iinclude -Imojo_opset_amd/csrc \
    scripts/probes/tile128_anatomy.hip mojo_opset_amd/csrc/gemm_tile128.hip -Lmojo_opset_amd/lib -lmojo_hip \
    -Wl,-rpath,'$ORIGIN/../../../mojo_opset_amd/lib' -o scripts/probes/_bin/t128_$v &
done
wait
