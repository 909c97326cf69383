"""Decode GQA: where does a ragged batch lose bandwidth?  Uniform vs odd lengths vs ragged, same shapes as bench.py."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.extras import _paged, _time, hip  # noqa: E402

dev = torch.device("cuda", 0)
hq, hkv, d, page, bsz = 32, 8, 128, 16, 64
op = hip("MojoPagedDecodeGQA")(is_causal=True, gqa_layout="AABB")
g = torch.Generator().manual_seed(20260716)
cases = {
    "uniform 4096": [4096] * bsz,
    "uniform 4090": [4090] * bsz,
    "uniform 3072": [3072] * bsz,
    "ragged 2048..4096 step 16": (torch.randint(128, 257, (bsz,), generator=g) * 16).tolist(),
    "ragged 2048..4096": torch.randint(2048, 4097, (bsz,), generator=g).tolist(),
    "half 2048 half 4096": [2048, 4096] * (bsz // 2),
    "sorted ragged": sorted(torch.randint(2048, 4097, (bsz,), generator=g).tolist()),
}
for name, lens in cases.items():
    sets = []
    for _ in range(2):
        k, v, table = _paged(dev, lens, hkv, d, page)
        q = torch.randn(bsz, hq, d, device=dev, dtype=torch.bfloat16)
        sets.append((q, k, v, torch.tensor(lens, dtype=torch.int32, device=dev), table))
    it = [0]

    def step():
        q, k, v, ln, tb = sets[it[0] % 2]
        it[0] += 1
        return op(q, k, v, ln, tb, max_total_seq_len=max(lens))
    t = _time(step, 40, 10)
    nbytes = sum(lens) * hkv * d * 4
    print(f"{name:28s} {t * 1e6:8.1f} us  {nbytes / t / 1e9:7.0f} GB/s")
    del sets
    torch.cuda.empty_cache()
