"""Round 5: MojoGroupGemm over group counts and rows per group (MoE prefill with few large or many small experts), both weight
layouts, against ONE dense product of the same total rows (no ragged edges, one weight matrix), and with the 128-row tiles forced off / on; device times (HIP graphs)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.extras import _time_graph, hip
from mojo_opset_amd import switches
from mojo_opset_amd.backends.hip import lib as L
from mojo_opset_amd.backends.hip.operators.gemm import dense_gemm
dev = torch.device("cuda", 0)
dt = torch.bfloat16
for k, n in ((4096, 14336), (2048, 1408), (7168, 4096), (1408, 2048)):
    for groups in (8, 64, 256):
        if groups * k * n * 2 > 6e9:
            continue
        for trans in (False, True):
            w = (torch.randn(groups, n, k, device=dev, dtype=dt) if trans else torch.randn(groups, k, n, device=dev, dtype=dt)) * 0.02
            op = hip("MojoGroupGemm")(w, trans)
            for rows in (8, 48, 80, 128, 200, 300, 512, 1024):
                total = groups * rows
                if total * n > 2 ** 30 or total > 65536:
                    continue
                g = torch.Generator().manual_seed(rows)
                counts = torch.full((groups,), rows, dtype=torch.int64)
                jitter = (torch.rand(groups, generator=g) * 0.6 + 0.7)                 # ragged: 0.7 .. 1.3 of the mean
                counts = (counts * jitter).long().clamp(min=0)
                counts[-1] += total - int(counts.sum())
                counts = counts.clamp(min=0)
                total = int(counts.sum())
                x = torch.randn(total, k, device=dev, dtype=dt)
                gl = counts.to(torch.int32).to(dev)
                reps = 6 if total * k * n < 2 ** 37 else 2
                t = _time_graph(lambda: op(x, gl), reps=reps)
                L.launch_history(clear=True); op(x, gl); form = L.launch_history()
                legs = {}
                for leg in ("0", "1"):                                                  # both kernels forced (the ragged weight stream stays where it applies)
                    os.environ["MOJO_HIP_GEMM_TILE128"] = leg; switches.reload()
                    legs[leg] = _time_graph(lambda: op(x, gl), reps=reps)
                    L.launch_history(clear=True); op(x, gl); legs[leg + "f"] = L.launch_history()
                os.environ.pop("MOJO_HIP_GEMM_TILE128"); switches.reload()
                wd = w[0] if trans else w[0]
                td = _time_graph(lambda: dense_gemm(x, wd, None, not trans), reps=reps)
                print(json.dumps({"k": k, "n": n, "groups": groups, "layout": "NK" if trans else "KN", "rows_mean": rows, "total_rows": total,
                                  "us": round(t * 1e6, 1), "tflops": round(2.0 * total * k * n / t / 1e12), "dense_same_rows_us": round(td * 1e6, 1),
                                  "vs_dense": round(t / td, 2), "form": form,
                                  "t256_us": round(legs["0"] * 1e6, 1), "t128_us": round(legs["1"] * 1e6, 1), "f256": legs["0f"], "f128": legs["1f"]}), flush=True)
            del op, w
