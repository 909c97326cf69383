"""Chunk-length sweep for paged decode GQA on small grids (8 q / 1 kv heads, B 64: 64 grid heads; and B 8 of the headline heads)."""
import os, sys, json, torch
sys.path.insert(0, ".")
from benchmarks import extras as X
dev = torch.device("cuda:0")
for name, (hq, hkv, d, lens) in {"8q1kv_B64_ctx4096": (8, 1, 128, [4096] * 64), "32q8kv_B8_ctx4096": (32, 8, 128, [4096] * 8),
                                 "8q1kv_B16_ctx16384": (8, 1, 128, [16384] * 16), "64q8kv_B8_ctx8192": (64, 8, 128, [8192] * 8)}.items():
    op = X.hip("MojoPagedDecodeGQA")(is_causal=True, gqa_layout="AABB")
    bsz = len(lens)
    sets = []
    for _ in range(3):
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
        for mf in ("0", "1"):
            os.environ["MOJO_HIP_DECODE_MFMA"] = mf
            for c in ("", "128", "256", "512", "1024", "2048"):
                if c: os.environ["MOJO_HIP_DECODE_CHUNK"] = c
                else: os.environ.pop("MOJO_HIP_DECODE_CHUNK", None)
                it[0] = 0
                res.setdefault(f"mfma{mf}_chunk{c or 'default'}", []).append(X._time_graph(step, reps=10, replays=10))
    os.environ.pop("MOJO_HIP_DECODE_CHUNK", None)
    os.environ.pop("MOJO_HIP_DECODE_MFMA", None)
    print(name, json.dumps({c: [round(min(ts) * 1e6, 1), round(nbytes / min(ts) / 8e12, 3)] for c, ts in res.items()}), flush=True)
    del sets
    torch.cuda.empty_cache()
