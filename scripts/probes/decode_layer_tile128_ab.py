"""Round 5: the Llama-3-8B decode layer (B 64, ctx 4096; benchmarks/extras.py bench_decode_layer: a graph of the whole layer, so every
weight is read cold) with the projections on the weight-streaming kernel (default at 64 rows) and forced onto the 128-row tiles
with their own K split (MOJO_HIP_GEMM_TILE128=1), same process, same box, twice each."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks import extras
from mojo_opset_amd import switches
dev = torch.device("cuda", 0)
BSZ = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for rep in range(4):
    for val in (None, "1"):
        os.environ.pop("MOJO_HIP_GEMM_TILE128", None)
        if val:
            os.environ["MOJO_HIP_GEMM_TILE128"] = val
        switches.reload()
        d = extras.bench_decode_layer(dev, BSZ)
        tag = f"llama3_8b_layer_B{BSZ}_ctx4096"
        print(json.dumps({"tile128": val or "default", "rep": rep, "B": BSZ,
                          **{k.replace(tag, "layer"): round(v["us"], 1) for k, v in d.items()},
                          "fused_ops": {a: round(b, 1) for a, b in d[tag + "_fused"]["per_op_us"].items()}}), flush=True)
