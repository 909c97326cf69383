"""A/B of two BUILDS of the library over the prefill bench cases: the in-tree library against MOJO_AB_PREV (a library built from an
earlier tree, e.g. mojo_opset_amd/lib/libmojo_hip_prev.so), alternated in child processes, three rounds each.
    git stash && python -m mojo_opset_amd.csrc.build && cp mojo_opset_amd/lib/libmojo_hip.so mojo_opset_amd/lib/libmojo_hip_prev.so
    git stash pop && python -m mojo_opset_amd.csrc.build && python scripts/probes/prefill_lib_ab.py"""
import json, os, subprocess, sys
HERE = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
prev = os.environ.get("MOJO_AB_PREV", os.path.join(HERE, "mojo_opset_amd", "lib", "libmojo_hip_prev.so"))
what = os.environ.get("MOJO_AB_BENCH", "bench_prefill")
child = ("import sys, json, torch; sys.path.insert(0, %r); from benchmarks import extras as X; "
         "print('RES ' + json.dumps(X.%s(torch.device('cuda:0'))))" % (HERE, what))
res = {}
for rnd in range(3):
    for arm, env in (("prev", {"MOJO_HIP_LIB": prev, "MOJO_HIP_ALLOW_STALE": "1"}), ("new", {})):
        out = subprocess.run([sys.executable, "-c", child], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith("RES ")]
        if not line:
            print(arm, "failed", out.stderr[-1500:]); sys.exit(1)
        for k, v in json.loads(line[0][4:]).items():
            res.setdefault(k, {}).setdefault(arm, []).append((v["us"], v.get("tflops", v.get("GB/s"))))
rec = {"prev": prev, "bench": what}
for k, v in res.items():
    rec[k] = {a: {"us_min": round(min(x[0] for x in xs), 1), "rate_max": round(max(x[1] for x in xs)), "us_all": [round(x[0], 1) for x in xs]} for a, xs in v.items()}
    print(k, {a: (b["us_min"], b["rate_max"]) for a, b in rec[k].items()}, flush=True)
json.dump(rec, open(os.path.join(HERE, "gpurun_out", "prefill_lib_ab.json"), "w"), indent=1)
