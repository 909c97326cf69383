import os, sys, random, torch
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
os.environ["MOJO_FUZZ_OFFSET"] = "208"
import test_hip_fuzz as F
from test_hip_mla import make_mla, build, exact_mla
from hip_utils import DEV, to_cpu
seed, OFFSET = 12, 208000
rnd = random.Random(3000 + seed + OFFSET)
nope, rope, vd, r = rnd.choice([(128, 64, 128, 512), (64, 32, 64, 32), (96, 32, 128, 64)])
h = rnd.choice([8, 16, 40, 128] if r == 512 else [8, 16])
page = rnd.choice([16, 32, 64]); sink = rnd.random() < 0.5; batch = rnd.choice([1, 2, 3, 6])
wscale = 0.05 if r == 512 else 0.2
dec = rnd.random() < 0.5
print("dims", nope, rope, vd, r, "h", h, "page", page, "sink", sink, "batch", batch, "decode", dec)
lens = [rnd.choice([0, 1, rnd.randint(1, 700), rnd.randint(200, 1500)]) for _ in range(batch)]
print("lens", lens)
ckv, kpe, table, w, sk = make_mla(lens, h, nope, rope, vd, r, page, sink, seed=seed, wscale=wscale)
g = torch.Generator().manual_seed(seed)
q = torch.randn(batch, h, nope + rope, generator=g).to(torch.bfloat16)
lens_t = torch.tensor(lens, dtype=torch.int32)
ref = build("MojoPagedDecodeMLA", h, nope, rope, vd, r, sink, w, sk, "cpu")
exact = exact_mla(q, ckv, kpe, table, w, sk, h, nope, rope, vd, r, lens)
gold = ref(q, ckv, kpe, lens_t, table).double()
print("golden err", float((gold - exact).abs().max()), "|exact| max", float(exact.abs().max()))
for kern in ("ps", "oct", "pp", "pair"):
    os.environ["MOJO_HIP_MLA_KERNEL"] = kern
    op = build("MojoPagedDecodeMLA", h, nope, rope, vd, r, sink, w, sk, DEV)
    got = to_cpu(op(q.to(DEV), ckv.to(DEV), kpe.to(DEV), lens_t.to(DEV), table.to(DEV))).double()
    e = (got - exact).abs()
    print(kern, "max err", float(e.max()), "n>8e-3", int((e > 8e-3).sum()), "mean", float(e.mean()))
