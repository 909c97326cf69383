"""(Historical: needs a build that instantiates the 8-wave form and reads MOJO_HIP_MLA_PREFILL_WAVES.)  A/B in one process: MLA prefill attention with 4-wave (128 query rows) and 8-wave (256 rows) workgroups
(MOJO_HIP_MLA_PREFILL_WAVES, read per call).  python scripts/probes/mla_prefill_waves_ab.py > profiles/r4_mla_prefill_waves_ab.json"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks import extras  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    out = {}
    for rep in range(2):
        for nw in ("4", "8"):
            os.environ["MOJO_HIP_MLA_PREFILL_WAVES"] = nw
            res = extras.bench_mla_prefill(dev)
            out[f"waves{nw}_rep{rep}"] = {k: {"us": v["us"], "tflops": v["tflops"]} for k, v in res.items()}
            print(nw, {k: round(v["us"], 1) for k, v in res.items()}, file=sys.stderr, flush=True)
    os.environ.pop("MOJO_HIP_MLA_PREFILL_WAVES")
    print(json.dumps(out, indent=1))
