"""GPU parity of MojoStorePagedMLAKVCache: bit-exact against the reference vectors and the oracle."""
import pytest
import torch

from conftest import load_golden
from hip_utils import DEV, hip_cls, run_hip_case, to_cpu, torch_cls

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", [pytest.param(c, id=f"store-mla-{i}") for i, c in enumerate(load_golden("store_paged_mla"))])
def test_store_mla_vectors_bit_exact(case):
    ckv_cache, kpe_cache = to_cpu(run_hip_case(case))
    assert torch.equal(ckv_cache, case["out"][0]) and torch.equal(kpe_cache, case["out"][1])


def _scenario(batch, page, r, rope, decode, seed, dtype=torch.bfloat16, holes=False):
    g = torch.Generator().manual_seed(seed)
    ctx = torch.randint(-1, 6 * page, (batch,), generator=g).to(torch.int32)
    new = [1] * batch if decode else torch.randint(0, 5 * page, (batch,), generator=g).tolist()
    need = [max((max(int(c), 0) + n + page - 1) // page, 1) for c, n in zip(ctx.tolist(), new)]
    total = sum(need) + 3
    ids = torch.randperm(total, generator=g).to(torch.int32)
    table = torch.full((batch, max(need)), -1, dtype=torch.int32)
    at = 0
    for b, n in enumerate(need):
        table[b, :n] = ids[at: at + n]
        at += n
    if holes:
        for b in range(0, batch, 3):
            table[b, int(torch.randint(0, need[b], (1,), generator=g))] = -1
    tokens = sum(new)
    ckv, kpe = torch.randn(tokens, r, generator=g).to(dtype), torch.randn(tokens, rope, generator=g).to(dtype)
    ckv_cache, kpe_cache = torch.randn(total, 1, page, r, generator=g).to(dtype), torch.randn(total, 1, page, rope, generator=g).to(dtype)
    cu_q = None if decode else torch.tensor([0] + torch.tensor(new).cumsum(0).tolist(), dtype=torch.int32)
    return ckv, kpe, ckv_cache, kpe_cache, table, cu_q, ctx


@pytest.mark.parametrize("batch,page,r,rope", [(7, 16, 512, 64), (64, 16, 512, 64), (5, 128, 64, 32), (3, 8, 24, 6)])
@pytest.mark.parametrize("decode", [True, False])
@pytest.mark.parametrize("holes", [False, True])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_store_mla_matches_oracle_bit_exact(batch, page, r, rope, decode, holes, dtype):
    args = _scenario(batch, page, r, rope, decode, seed=batch * 31 + page, dtype=dtype, holes=holes)
    want = torch_cls("MojoStorePagedMLAKVCache")()(*[None if a is None else a.clone() for a in args])
    dev_args = [None if a is None else a.to(DEV) for a in args]
    got = hip_cls("MojoStorePagedMLAKVCache")()(*dev_args)
    assert got[0].data_ptr() == dev_args[2].data_ptr() and got[1].data_ptr() == dev_args[3].data_ptr()   # in place
    assert torch.equal(to_cpu(got[0]), want[0]) and torch.equal(to_cpu(got[1]), want[1])


def test_store_then_decode_mla_round_trip():
    """Tokens written by the store are the ones the MLA decode kernel reads back: attention over a cache filled by the
    store equals attention over the same cache filled by indexing on the host."""
    torch.manual_seed(0)
    b, h, nope, rope, vd, r, page = 3, 16, 64, 32, 64, 64, 16
    lens = [40, 17, 64]
    tokens = sum(lens)
    ckv, kpe = torch.randn(tokens, r, dtype=torch.bfloat16), torch.randn(tokens, rope, dtype=torch.bfloat16)
    need = [(n + page - 1) // page for n in lens]
    total = sum(need) + 2
    table = torch.full((b, max(need)), -1, dtype=torch.int32)
    ids = torch.randperm(total, dtype=torch.int32)
    at = 0
    for i, n in enumerate(need):
        table[i, :n] = ids[at: at + n]
        at += n
    cu_q = torch.tensor([0, 40, 57, 121], dtype=torch.int32)
    ctx = torch.zeros(b, dtype=torch.int32)
    caches = [torch.zeros(total, 1, page, r, dtype=torch.bfloat16), torch.zeros(total, 1, page, rope, dtype=torch.bfloat16)]
    host = torch_cls("MojoStorePagedMLAKVCache")()(ckv, kpe, caches[0].clone(), caches[1].clone(), table, cu_q, ctx)
    dev = hip_cls("MojoStorePagedMLAKVCache")()(ckv.to(DEV), kpe.to(DEV), caches[0].to(DEV), caches[1].to(DEV), table.to(DEV),
                                                 cu_q.to(DEV), ctx.to(DEV))
    attn = hip_cls("MojoPagedDecodeMLA")(h, nope, rope, vd, r).to(torch.bfloat16).to(DEV)
    with torch.no_grad():
        attn.kv_b_proj.copy_(torch.randn_like(attn.kv_b_proj) * 0.05)
    q = torch.randn(b, h, nope + rope, dtype=torch.bfloat16, device=DEV)
    seq = torch.tensor(lens, dtype=torch.int32, device=DEV)
    a = attn(q, dev[0], dev[1], seq, table.to(DEV))
    bb = attn(q, host[0].to(DEV), host[1].to(DEV), seq, table.to(DEV))
    assert torch.equal(a, bb)
