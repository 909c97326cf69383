"""A/B in one process: head_dim 64 decode (B 64, 32 q / 8 kv, ctx 4096) at the default chunking and at finer ones
(MOJO_HIP_DECODE_CHUNK, read per call): does twice the resident waves pay once the instance fits four waves per SIMD?"""
import os, sys, json, torch
sys.path.insert(0, ".")
from benchmarks import extras as X
dev = torch.device("cuda:0")
for name, (hq, hkv, d, lens) in {"G4_d64": (32, 8, 64, [4096] * 64), "G4_d64_ctx1024": (32, 8, 64, [1024] * 64),
                                 "G8_d64": (64, 8, 64, [4096] * 64)}.items():
    op = X.hip("MojoPagedDecodeGQA")(is_causal=True, gqa_layout="AABB")
    bsz = len(lens)
    sets = []
    for _ in range(2):
        k, v, table = X._paged(dev, lens, hkv, d, 16)
        q = torch.randn(bsz, hq, d, device=dev, dtype=torch.bfloat16)
        sets.append((q, k, v, torch.tensor(lens, dtype=torch.int32, device=dev), table))
    it = [0]
    def step():
        q, k, v, ln, tb = sets[it[0] % len(sets)]
        it[0] += 1
        return op(q, k, v, ln, tb, max_total_seq_len=max(lens))
    nbytes = sum(lens) * hkv * d * 2 * 2 + 2 * bsz * hq * d * 2 + 4 * bsz * (sets[0][4].shape[1] + 1)
    res = {}
    for rnd in range(2):
        for c in ("", "1024", "688", "512", "352", "256"):
            if c: os.environ["MOJO_HIP_DECODE_CHUNK"] = c
            else: os.environ.pop("MOJO_HIP_DECODE_CHUNK", None)
            it[0] = 0
            res.setdefault(c or "default", []).append(X._time_graph(step, reps=10, replays=10))
    os.environ.pop("MOJO_HIP_DECODE_CHUNK", None)
    print(name, json.dumps({c: {"us": round(min(ts) * 1e6, 1), "frac": round(nbytes / min(ts) / 8e12, 3)} for c, ts in res.items()}), flush=True)
    del sets
    torch.cuda.empty_cache()
