"""Does the decode kernel hold its burst rate?  Back-to-back launches (one HIP graph of 20, replayed) at ctx 1024 and 4096
with 2 and 6 K/V sets in rotation; run under `rocprofv3 --kernel-trace` and read the duration series with
scripts/probes/decode_sustain_read.py (the series, not the mean, is the result)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__  # noqa
from benchmarks.extras import _paged, hip
dev = torch.device("cuda:0")
hq, hkv, d, page, bsz = 32, 8, 128, 16, 64
op = hip("MojoPagedDecodeGQA")(is_causal=True, gqa_layout="AABB")

def run(ctx, n_sets, replays):
    sets = []
    for _ in range(n_sets):
        k, v, table = _paged(dev, [ctx] * bsz, hkv, d, page)
        sets.append((torch.randn(bsz, hq, d, device=dev, dtype=torch.bfloat16), k, v, table))
    lens = torch.full((bsz,), ctx, dtype=torch.int32, device=dev)
    it = [0]
    def fn():
        it[0] += 1
        q, k, v, table = sets[it[0] % n_sets]
        return op(q, k, v, lens, table)
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): fn()
    torch.cuda.synchronize(); time.sleep(1.0)                 # idle: the chip returns to its resting state
    for _ in range(replays): g.replay()
    torch.cuda.synchronize(); time.sleep(1.0)
    del sets, g
    torch.cuda.empty_cache()

run(1024, 2, 20)      # 400 launches
run(4096, 2, 10)      # 200 launches
run(1024, 6, 20)
run(1024, 1, 20)
