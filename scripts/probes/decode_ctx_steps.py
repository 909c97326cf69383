"""MojoPagedDecodeGQA around a context boundary: B = 64, Hq/Hkv = 32/8, D = 128, page 16; time per launch under graph replay
for total lengths 4096 +- a few tokens (a serving loop sits at every one of them in turn)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__  # noqa
from benchmarks.extras import _time_graph, _paged, hip
dev = torch.device("cuda:0")
hq, hkv, d, page, bsz = 32, 8, 128, 16, 64
op = hip("MojoPagedDecodeGQA")(is_causal=True, gqa_layout="AABB")
sets = []
for _ in range(2):
    k, v, table = _paged(dev, [4352] * bsz, hkv, d, page)
    sets.append((torch.randn(bsz, hq, d, device=dev, dtype=torch.bfloat16), k, v, table))
for n in (4080, 4095, 4096, 4097, 4100, 4112, 4128, 4160, 4224, 4352):
    lens = torch.full((bsz,), n, dtype=torch.int32, device=dev)
    for hint in (None, n):
        it = [0]
        def fn():
            it[0] += 1
            q, k, v, table = sets[it[0] % 2]
            return op(q, k, v, lens, table, max_total_seq_len=hint)
        t = _time_graph(fn, reps=4, replays=5)
        print(f"len {n} hint {hint}: {t*1e6:7.1f} us  {bsz*n*hkv*d*2*2/t/1e12:5.2f} TB/s", flush=True)
