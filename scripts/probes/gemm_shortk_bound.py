"""Round 5: what would hiding the epilogue of the short-K GEMM buy?  The MLA decompression product
[T_kv, 512] x [H * 256, 512]^T spends ~35 % of each 256 x 256 tile outside its K loop.  Experiments build only:
    shipped           : the default launch
    direct stores     : MOJO_HIP_GEMM_STAGE_ROWS=0
    no C stores       : + MOJO_HIP_GEMM_ABLATE=1 (timing only: one row per tile is stored) = the tile WITHOUT its epilogue, i.e. the
                        ceiling of any design that runs the epilogue under another workgroup's K loop
    persistent        : MOJO_HIP_GEMM_PERSIST=1 (next tile's loads requested before this tile's stores, same workgroup)
python scripts/probes/gemm_shortk_bound.py"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.extras import _time, hip  # noqa: E402
from mojo_opset_amd import switches
dev = torch.device("cuda", 0)
ARMS = {"shipped": {}, "direct_stores": {"MOJO_HIP_GEMM_STAGE_ROWS": "0"}, "no_C_stores(timing only)": {"MOJO_HIP_GEMM_STAGE_ROWS": "0", "MOJO_HIP_GEMM_ABLATE": "1"},
        "persistent": {"MOJO_HIP_GEMM_PERSIST": "1"}, "no_stagger": {"MOJO_HIP_GEMM_STAGGER": "0"}}
out = {}
for m, k, n in ((2048, 512, 32768), (10240, 512, 32768), (2048, 1024, 32768), (16384, 4096, 4096)):
    x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    w = torch.randn(1, n, k, device=dev, dtype=torch.bfloat16) * 0.05
    op = hip("MojoGroupGemm")(w, True)
    counts = torch.tensor([m], dtype=torch.int32, device=dev)
    rec = {}
    for rep in range(2):
        for arm, env in ARMS.items():
            for kk in ("MOJO_HIP_GEMM_STAGE_ROWS", "MOJO_HIP_GEMM_ABLATE", "MOJO_HIP_GEMM_PERSIST", "MOJO_HIP_GEMM_STAGGER"):
                os.environ.pop(kk, None)
            os.environ.update(env)
            switches.reload()
            t = _time(lambda: op(x, counts), 20, 5)
            rec.setdefault(arm, []).append(round(t * 1e6, 1))
    out[f"{m}x{k}x{n}"] = {a: {"us": min(v), "tflops": round(2.0 * m * k * n / (min(v) * 1e-6) / 1e12)} for a, v in rec.items()}
    print(json.dumps({f"{m}x{k}x{n}": out[f"{m}x{k}x{n}"]}), flush=True)
for kk in ("MOJO_HIP_GEMM_STAGE_ROWS", "MOJO_HIP_GEMM_ABLATE", "MOJO_HIP_GEMM_PERSIST", "MOJO_HIP_GEMM_STAGGER"):
    os.environ.pop(kk, None)
switches.reload()
