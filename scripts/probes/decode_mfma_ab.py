"""A/B in one process: MojoPagedDecodeGQA with the vector-unit kernel (MOJO_HIP_DECODE_MFMA=0) against the matrix-core
kernel (=1) over the bench geometries; the switch is read per call, the two are timed alternately (graph replay)."""
import os, sys, json, torch
sys.path.insert(0, ".")
from benchmarks import extras as X
dev = torch.device("cuda:0")
cases = {
    "headline_32q8kv_d128_ctx4096": (32, 8, 128, [4096] * 64),
    "ctx1024": (32, 8, 128, [1024] * 64),
    "ctx16384": (32, 8, 128, [16384] * 64),
    "ragged_2048_4096": (32, 8, 128, torch.randint(2048, 4097, (64,), generator=torch.Generator().manual_seed(20260716)).tolist()),
    "G8_70b": (64, 8, 128, [4096] * 64),
    "G8_tp8": (8, 1, 128, [4096] * 64),
    "G4_d64": (32, 8, 64, [4096] * 64),
    "G2_16q8kv": (16, 8, 128, [4096] * 64),
    "G1_8q8kv": (8, 8, 128, [4096] * 64),
    "B8_ctx4096": (32, 8, 128, [4096] * 8),
    "B256_ctx1024": (32, 8, 128, [1024] * 256),
}
out = {}
for name, (hq, hkv, d, lens) in cases.items():
    op = X.hip("MojoPagedDecodeGQA")(is_causal=True, gqa_layout="AABB")
    bsz = len(lens)
    sets = []
    for _ in range(2 if max(lens) <= 4096 else 1):
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
        for m in ("0", "1"):
            os.environ["MOJO_HIP_DECODE_MFMA"] = m
            it[0] = 0
            t = X._time_graph(step, reps=10, replays=10)
            res.setdefault(m, []).append(t)
    out[name] = {m: {"us": min(ts) * 1e6, "frac": nbytes / min(ts) / 8e12} for m, ts in res.items()}
    print(name, json.dumps(out[name]), flush=True)
    del sets
    torch.cuda.empty_cache()
