"""Generate tests/golden/*.pt from the imported reference.  RUN IN THE AUTHORING CONTAINER ONLY.

    python oracle/make_golden.py            # needs /root/reference (never present on the GPU box)

For every op on the hot path this script builds small seeded inputs, runs the
reference's own torch-native golden class (``mojo_opset.<Mojo*>._registry.get("torch")``) on CPU
and stores ``{ctor, state, args, kwargs, out}`` per case.  It then runs this repo's restatement
(``oracle.torch_golden``) on the same inputs and aborts unless every output is bit-identical
(`torch.equal`).  The fixtures are data only: tensors and scalars, no reference source.
"""
import math
import functools
import os
import sys

import torch

REF = os.environ.get("MOJO_REFERENCE_ROOT", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

os.environ.setdefault("MOJO_OPSET_PLUGIN_AUTOLOAD", "0")
import mojo_opset as ref                      # noqa: E402  the reference
import mojo_opset.experimental as ref_exp     # noqa: E402
from mojo_opset.core.operators.kv_cache import build_paged_kv_chunk_metadata as ref_build_plan  # noqa: E402

import mojo_opset_amd as mine                 # noqa: E402
import oracle                                 # noqa: E402,F401  registers Torch* into mine
from mojo_opset_amd.core.operators.kv_cache import build_paged_kv_chunk_metadata as my_build_plan  # noqa: E402


def ref_cls(name):
    mod = ref if hasattr(ref, name) else ref_exp
    return getattr(mod, name)._registry.get("torch")


def my_cls(name):
    return getattr(mine, name)._registry.get("torch", strict=True)


def clone(x):
    if isinstance(x, torch.Tensor):
        return x.clone()
    if isinstance(x, (list, tuple)):
        return type(x)(clone(v) for v in x)
    if isinstance(x, dict):
        return {k: clone(v) for k, v in x.items()}
    return x


def same(a, b):
    if isinstance(a, (tuple, list)):
        return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
    if a.dtype != b.dtype or a.shape != b.shape:
        return False
    if a.is_floating_point():
        return torch.equal(torch.nan_to_num(a.float(), nan=12345.0), torch.nan_to_num(b.float(), nan=12345.0))
    return torch.equal(a, b)


def run_case(op_name, ctor, state, args, kwargs, cast=None, keep_dtype=()):
    """Run reference and restatement; return the fixture record."""
    outs = []
    for cls in (ref_cls(op_name), my_cls(op_name)):
        op = cls(*clone(ctor.get("args", ())), **clone(ctor.get("kwargs", {})))
        if cast is not None:
            op = op.to(cast)
        with torch.no_grad():
            for k, v in state.items():
                slot = functools.reduce(getattr, k.split("."), op)
                if k in keep_dtype:                  # e.g. the fp32 router weight inside a bf16 layer
                    slot.data = v.clone()
                else:
                    slot.copy_(v)
            outs.append(op.forward(*clone(args), **clone(kwargs)))
    if not same(outs[0], outs[1]):
        raise SystemExit(f"restatement of {op_name} is NOT bit-identical to the reference for ctor={ctor}")
    return {"op": op_name, "ctor": ctor, "state": state, "args": args, "kwargs": kwargs, "cast": cast,
            "keep_dtype": tuple(keep_dtype), "out": outs[0]}


# --------------------------------------------------------------------------------------------
# input builders (this repo's own; seeded)
# --------------------------------------------------------------------------------------------
def paged_cache(lens, heads, page, dim, dtype, spare=3, hole_after=None):
    """Random K/V pools and a shuffled block table, -1 padded; ``hole_after[b] = j`` puts a -1 at
    logical page j of sequence b (and leaves later entries valid) to exercise truncation."""
    need = [(n + page - 1) // page for n in lens]
    total = sum(need) + spare
    width = max(max(need), 1)
    k = torch.randn(total, heads, page, dim, dtype=dtype)
    v = torch.randn(total, heads, page, dim, dtype=dtype)
    ids = torch.randperm(total, dtype=torch.int32)
    table = torch.full((len(lens), width), -1, dtype=torch.int32)
    at = 0
    for b, n in enumerate(need):
        table[b, :n] = ids[at: at + n]
        at += n
    if hole_after:
        for b, j in hole_after.items():
            table[b, j] = -1
    return k, v, table


def gen_decode_gqa():
    recs = []
    cfgs = [  # lens, Hq, Hkv, D, page
        ([37, 1, 200, 16], 8, 2, 128, 16),
        ([5, 130, 64], 8, 2, 96, 32),
        ([0, 3, 0, 129], 8, 1, 128, 128),
        ([70, 33], 4, 4, 64, 16),
    ]
    for ci, (lens, hq, hkv, d, page) in enumerate(cfgs):
        for layout in ("AABB", "ABAB"):
            torch.manual_seed(100 + ci)
            k, v, table = paged_cache(lens, hkv, page, d, torch.bfloat16)
            q = torch.randn(len(lens), hq, d, dtype=torch.bfloat16)
            recs.append(run_case("MojoPagedDecodeGQA", {"kwargs": {"is_causal": True, "gqa_layout": layout}}, {},
                                 (q, k, v, torch.tensor(lens, dtype=torch.int32), table),
                                 {"softmax_scale": 1.0 / math.sqrt(d), "max_total_seq_len": max(lens)}))
    # truncated table: -1 at logical page 2 of sequence 0 -> zero K/V tail (reference `break`)
    torch.manual_seed(150)
    lens = [100, 40]
    k, v, table = paged_cache(lens, 2, 16, 128, torch.bfloat16, hole_after={0: 2})
    q = torch.randn(2, 8, 128, dtype=torch.bfloat16)
    recs.append(run_case("MojoPagedDecodeGQA", {"kwargs": {}}, {},
                         (q, k, v, torch.tensor(lens, dtype=torch.int32), table), {}))
    # fp16
    torch.manual_seed(151)
    lens = [45, 17, 96]
    k, v, table = paged_cache(lens, 2, 16, 128, torch.float16)
    q = torch.randn(3, 8, 128, dtype=torch.float16)
    recs.append(run_case("MojoPagedDecodeGQA", {"kwargs": {}}, {},
                         (q, k, v, torch.tensor(lens, dtype=torch.int32), table), {}))
    return recs


def cu(lens):
    return torch.tensor([0] + list(torch.tensor(lens).cumsum(0).tolist()), dtype=torch.int32)


def gen_prefill_gqa():
    recs = []
    cfgs = [  # q_lens, cached, Hq, Hkv, D, page
        ([40, 17], [0, 0], 8, 2, 128, 16),
        ([33, 64, 1], [20, 0, 77], 8, 2, 128, 16),
        ([50, 9], [14, 120], 4, 1, 96, 32),
        ([0, 25, 0], [0, 10, 5], 4, 2, 128, 16),
    ]
    for ci, (q_lens, cached, hq, hkv, d, page) in enumerate(cfgs):
        for layout in ("AABB", "ABAB"):
            torch.manual_seed(200 + ci)
            kv_lens = [a + b for a, b in zip(q_lens, cached)]
            k, v, table = paged_cache(kv_lens, hkv, page, d, torch.bfloat16)
            q = torch.randn(sum(q_lens), hq, d, dtype=torch.bfloat16)
            kwargs = {"softmax_scale": 1.0 / math.sqrt(d), "max_q_len": max(q_lens), "max_total_seq_len": max(kv_lens)}
            if any(cached):
                kwargs["cu_total_seq_lens"] = cu(kv_lens)
            recs.append(run_case("MojoPagedPrefillGQA", {"kwargs": {"is_causal": True, "gqa_layout": layout}}, {},
                                 (q, k, v, cu(q_lens), table), kwargs))
    return recs


def gen_mla():
    recs = []
    for ci, (lens, h, nope, rope, vd, r, page, sink) in enumerate([
        ([40, 7, 65], 8, 64, 32, 64, 32, 16, False),
        ([33, 100], 16, 96, 32, 128, 64, 32, True),
        ([0, 12, 0], 8, 64, 32, 64, 32, 16, False),
    ]):
        torch.manual_seed(300 + ci)
        need = [(n + page - 1) // page for n in lens]
        total = sum(need) + 3
        ckv = torch.randn(total, 1, page, r, dtype=torch.bfloat16)
        kpe = torch.randn(total, 1, page, rope, dtype=torch.bfloat16)
        ids = torch.randperm(total, dtype=torch.int32)
        table = torch.full((len(lens), max(max(need), 1)), -1, dtype=torch.int32)
        at = 0
        for b, n in enumerate(need):
            table[b, :n] = ids[at: at + n]
            at += n
        wproj = (torch.randn(h * (nope + vd), r) * 0.2).to(torch.bfloat16)
        state = {"kv_b_proj": wproj}
        if sink:
            state["attn_sink"] = torch.randn(h)
        ctor = {"kwargs": dict(num_heads=h, qk_nope_head_dim=nope, qk_rope_head_dim=rope, v_head_dim=vd,
                               kv_lora_rank=r, use_attn_sink=sink)}
        q = torch.randn(len(lens), h, nope + rope, dtype=torch.bfloat16)
        recs.append(run_case("MojoPagedDecodeMLA", ctor, state,
                             (q, ckv, kpe, torch.tensor(lens, dtype=torch.int32), table), {}, cast=torch.bfloat16))
        # prefill over the same pools: the last `q_len` tokens of each sequence are queries
        q_lens = [min(n, 9 + 3 * b) for b, n in enumerate(lens)]
        qp = torch.randn(sum(q_lens), h, nope + rope, dtype=torch.bfloat16)
        ctor_p = {"kwargs": dict(ctor["kwargs"], is_causal=True)}
        recs.append(run_case("MojoPagedPrefillMLA", ctor_p, state, (qp, ckv, kpe, cu(q_lens), table),
                             {"cu_total_seq_lens": cu(lens)}, cast=torch.bfloat16))
    return recs


def gen_norm():
    recs = []
    for ci, (rows, d) in enumerate([(8, 1024), (5, 738), (2, 256), (3, 4096)]):
        for dtype in (torch.bfloat16, torch.float16, torch.float32):
            for pos in ("pre", "post"):
                torch.manual_seed(400 + ci)
                x = torch.randn(rows, d, dtype=dtype)
                r = torch.randn(rows, d, dtype=dtype)
                w = torch.randn(d, dtype=dtype)
                recs.append(run_case("MojoResidualAddRMSNorm",
                                     {"kwargs": {"norm_size": d, "eps": 1e-5, "norm_pos": pos, "dtype": dtype}},
                                     {"weight": w}, (x, r), {}))
            torch.manual_seed(450 + ci)
            x = torch.randn(2, rows, d, dtype=dtype)
            w = torch.randn(d, dtype=dtype)
            recs.append(run_case("MojoRMSNorm", {"kwargs": {"norm_size": d, "eps": 1e-6, "dtype": dtype}},
                                 {"weight": w}, (x,), {}))
            # experimental/operators/normalization.py:95-140 (q/k-norm of the Qwen3 stack: [T, heads, head_dim])
            for inplace in (False, True):
                torch.manual_seed(470 + ci)
                x = torch.randn(rows, 4, d // 4 if d % 4 == 0 else d, dtype=dtype)
                w = torch.randn(x.shape[-1], dtype=dtype)
                recs.append(run_case("MojoRMSNormInplace",
                                     {"kwargs": {"norm_size": x.shape[-1], "eps": 1e-6, "inplace": inplace, "dtype": dtype}},
                                     {"weight": w}, (x,), {}))
    return recs


def gen_swiglu():
    recs = []
    for ci, (shape, dtype, lim) in enumerate([
        ((16, 128), torch.bfloat16, 0.0), ((9, 999), torch.bfloat16, 0.0), ((4, 8, 256), torch.float16, 0.0),
        ((7, 512), torch.float32, 0.0), ((16, 128), torch.bfloat16, 1.5), ((5, 40), torch.float32, 0.75),
    ]):
        torch.manual_seed(500 + ci)
        g = torch.randn(*shape, dtype=dtype) * 3
        u = torch.randn(*shape, dtype=dtype) * 3
        recs.append(run_case("MojoSwiGLU", {"kwargs": {"swiglu_limit": lim}}, {}, (g, u), {}))
    return recs


def gen_rope():
    recs = []
    # rotary embedding: cached and uncached, the three calling modes
    for ci, (d, cached) in enumerate([(64, 512), (88, 512), (128, None), (32, 256)]):
        ctor = {"kwargs": {"rope_theta": 10000.0, "rope_dim": d, "attention_scaling": 1.0 if ci % 2 == 0 else 0.8,
                           "init_max_length": cached}}
        torch.manual_seed(600 + ci)
        q_lens, tot = [5, 0, 17, 3], [40, 7, 17, 90]
        x = torch.randn(sum(q_lens), 16)
        recs.append(run_case("MojoRotaryEmbedding", ctor, {}, (x,), {"cu_q_lens": cu(q_lens)}))
        recs.append(run_case("MojoRotaryEmbedding", ctor, {}, (x,),
                             {"cu_q_lens": cu(q_lens), "total_seq_lens": torch.tensor(tot, dtype=torch.int32)}))
        recs.append(run_case("MojoRotaryEmbedding", ctor, {}, (torch.randn(2, 19, 16),), {}))
        recs.append(run_case("MojoRotaryEmbedding", ctor, {}, (torch.randn(6, 16),),
                             {"position_ids": torch.tensor([3, 0, 200, 77, 1, 255], dtype=torch.int32)}))
    # apply rope
    cfgs = [  # dtype, Hq, Hk, head_first, D, rope_dim
        (torch.float16, 8, 2, True, 96, 96), (torch.bfloat16, 8, 2, False, 96, 32),
        (torch.float16, 4, 4, True, 128, 128), (torch.bfloat16, 16, 2, False, 88, 88),
        (torch.float16, 8, 1, True, 128, 48), (torch.float32, 4, 2, False, 64, 64),
    ]
    for ci, (dtype, hq, hk, head_first, d, rd) in enumerate(cfgs):
        torch.manual_seed(650 + ci)
        bs, seq = 2, 13
        table = torch.randn(64, rd), torch.randn(64, rd)

        def qk(*lead):
            shp_q = (*lead[:-1], hq, lead[-1], d) if head_first else (*lead, hq, d)
            shp_k = (*lead[:-1], hk, lead[-1], d) if head_first else (*lead, hk, d)
            return torch.randn(*shp_q, dtype=dtype), torch.randn(*shp_k, dtype=dtype)

        q, k = qk(seq)                                     # varlen / decode: [T,N,D] or [N,T,D]
        recs.append(run_case("MojoApplyRoPE", {"kwargs": {}}, {}, (q, k, table[0][:seq], table[1][:seq]),
                             {"head_first": head_first}))
        q, k = qk(bs, seq)                                 # padded: cos [S,d]
        recs.append(run_case("MojoApplyRoPE", {"kwargs": {}}, {}, (q, k, table[0][:seq], table[1][:seq]),
                             {"head_first": head_first}))
        cos_b = torch.randn(bs, seq, rd)                   # padded: cos [B,S,d]
        sin_b = torch.randn(bs, seq, rd)
        recs.append(run_case("MojoApplyRoPE", {"kwargs": {}}, {}, (q, k, cos_b, sin_b), {"head_first": head_first}))
    return recs


def gen_store_kv():
    recs, plans = [], []
    cfgs = [  # (ctx, q_len) per sequence, Hkv, D, page, dtype
        ([(0, 5), (16, 16), (30, 3), (-1, 4), (7, 0)], 2, 128, 16, torch.bfloat16),
        ([(100, 60), (0, 1), (255, 2)], 4, 64, 128, torch.float16),
        ([(3, 9), (8, 8), (0, 24)], 3, 96, 8, torch.bfloat16),
        ([(0, 33)], 1, 128, 32, torch.float32),
    ]
    for ci, (seqs, hkv, d, page, dtype) in enumerate(cfgs):
        torch.manual_seed(700 + ci)
        ctx = torch.tensor([c for c, _ in seqs], dtype=torch.int32)
        q_lens = [q for _, q in seqs]
        pages_needed = [(max(c, 0) + q + page - 1) // page for c, q in seqs]
        total = sum(pages_needed) + 2
        ids = torch.randperm(total, dtype=torch.int32)
        table = torch.full((len(seqs), max(pages_needed) + 1), -1, dtype=torch.int32)
        at = 0
        for b, n in enumerate(pages_needed):
            table[b, :n] = ids[at: at + n]
            at += n
        T = sum(q_lens)
        ks, vs = torch.randn(T, hkv, d, dtype=dtype), torch.randn(T, hkv, d, dtype=dtype)
        kc, vc = torch.randn(total, hkv, page, d, dtype=dtype), torch.randn(total, hkv, page, d, dtype=dtype)
        # legacy prefill arguments
        recs.append(run_case("MojoStorePagedKVCache", {"kwargs": {}}, {}, (ks, vs, kc, vc, table, cu(q_lens), ctx), {}))
        # prebuilt plan
        plan_ref = ref_build_plan(table, cu(q_lens), ctx, page)
        plan_mine = my_build_plan(table, cu(q_lens), ctx, page)
        assert torch.equal(plan_ref, plan_mine), "prefill plan builder differs from the reference"
        plans.append({"args": (table, cu(q_lens), ctx, page), "out": plan_ref})
        recs.append(run_case("MojoStorePagedKVCache", {"kwargs": {}}, {}, (ks, vs, kc, vc), {"chunk_metadata": plan_ref}))
        # decode mode: one token per sequence
        B = len(seqs)
        ks1, vs1 = torch.randn(B, hkv, d, dtype=dtype), torch.randn(B, hkv, d, dtype=dtype)
        ctx_d = ctx.clone()
        ctx_d[-1] = (table.shape[1] + 2) * page            # beyond the table -> dropped
        recs.append(run_case("MojoStorePagedKVCache", {"kwargs": {}}, {}, (ks1, vs1, kc, vc, table, None, ctx_d), {}))
        plan_ref = ref_build_plan(table, None, ctx_d, page)
        assert torch.equal(plan_ref, my_build_plan(table, None, ctx_d, page)), "decode plan builder differs"
        plans.append({"args": (table, None, ctx_d, page), "out": plan_ref})
    return recs, plans


def gen_group_gemm():
    recs = []
    for ci, (counts, k, n, trans, dtype) in enumerate([
        ([16, 64, 32, 80], 128, 96, False, torch.bfloat16), ([16, 64, 32, 80], 128, 96, True, torch.bfloat16),
        ([0, 40, 0, 7], 64, 128, False, torch.float16), ([130], 256, 64, True, torch.bfloat16),
        ([33, 1], 72, 40, False, torch.float32),
    ]):
        torch.manual_seed(800 + ci)
        g = len(counts)
        x = torch.randn(sum(counts), k, dtype=dtype)
        w = torch.randn(g, n, k, dtype=dtype) if trans else torch.randn(g, k, n, dtype=dtype)
        recs.append(run_case("MojoGroupGemm", {"args": (w,), "kwargs": {"trans_weight": trans}}, {},
                             (x, torch.tensor(counts, dtype=torch.int32)), {}))
    return recs


def gen_dense():
    """`MojoGemm` (core/operators/gemm.py:12-56) and `MojoSwiGLUMLP` (core/operators/mlp.py:7-37): the reference's hooks for the
    dense projections and the gated MLP of a decoder layer (decode-sized and prefill-sized rows, with / without bias, the
    ``weight=`` constructor form, 3-D inputs)."""
    recs = []
    for ci, (m, k, n, dtype, bias) in enumerate([
        ((33,), 256, 192, torch.bfloat16, True), ((5,), 384, 64, torch.float16, False), ((2, 17), 128, 96, torch.bfloat16, True),
        ((300,), 256, 136, torch.float16, True), ((7,), 64, 40, torch.float32, True), ((64,), 512, 256, torch.bfloat16, False),
    ]):
        torch.manual_seed(2100 + ci)
        x = torch.randn(*m, k).to(dtype)
        state = {"weight": (torch.randn(n, k) * 0.1).to(dtype)}
        if bias:
            state["bias"] = torch.randn(n).to(dtype)
        recs.append(run_case("MojoGemm", {"args": (k, n), "kwargs": {"bias": bias, "dtype": dtype}}, state, (x,), {}))
    torch.manual_seed(2150)
    w = (torch.randn(48, 128) * 0.1).to(torch.bfloat16)
    recs.append(run_case("MojoGemm", {"kwargs": {"weight": w}}, {}, (torch.randn(9, 128).to(torch.bfloat16),), {}))
    for ci, (m, inp, outp, hidden, dtype) in enumerate([
        ((40,), 128, 64, 96, torch.bfloat16), ((3,), 128, 128, 192, torch.bfloat16), ((2, 9), 64, 32, 72, torch.float16),
        ((130,), 96, 64, 128, torch.float16), ((6,), 32, 16, 24, torch.float32),
    ]):
        torch.manual_seed(2200 + ci)
        x = torch.randn(*m, inp).to(dtype)
        state = {"fc1.weight": (torch.randn(2 * hidden, inp) * 0.1).to(dtype), "fc2.weight": (torch.randn(outp, hidden) * 0.1).to(dtype)}
        recs.append(run_case("MojoSwiGLUMLP", {"args": (inp, outp, hidden)}, state, (x,), {}, cast=dtype))
    return recs


def gen_moe():
    """Gating / dispatch / experts / combine on the reference's own small shapes (tests/accuracy/operators/test_moe.py)
    plus ragged and empty-bucket cases."""
    recs = []
    for ci, (experts, k, hidden, tokens, std) in enumerate([(16, 4, 1024, 64, 0.02), (8, 2, 256, 37, 0.02), (64, 8, 512, 48, 0.05),
                                                            (384, 8, 128, 16, 0.5)]):
        torch.manual_seed(1300 + ci)
        w = torch.randn(hidden, experts) * std
        x = torch.rand(tokens, hidden, dtype=torch.bfloat16)
        probs = torch.softmax(x.float() @ w, dim=-1).sort(dim=-1, descending=True).values
        assert float((probs[:, :k] - probs[:, 1: k + 1]).min()) > 1e-5, "near-tie in the gating vector: pick another seed"
        recs.append(run_case("MojoMoEGating", {"kwargs": {"hidden_size": hidden, "num_experts": experts, "top_k": k}},
                             {"gate_weight": w}, (x,), {}))
    for ci, (experts, k, hidden, tokens) in enumerate([(16, 4, 256, 64), (8, 2, 128, 37), (64, 8, 64, 48), (384, 8, 96, 16), (4, 1, 64, 0)]):
        torch.manual_seed(1320 + ci)
        x = torch.rand(tokens, hidden, dtype=torch.bfloat16)
        probs = torch.softmax(torch.randn(tokens, experts), dim=-1)
        gates, idx = torch.topk(probs, k, dim=-1)
        gates = (gates / gates.sum(dim=-1, keepdim=True)).contiguous()
        recs.append(run_case("MojoMoEDispatch", {"kwargs": {"num_experts": experts}}, {},
                             (x, gates, idx.to(torch.int32).contiguous()), {}))
    for ci, (experts, hidden, inter, counts) in enumerate([(4, 256, 512, [3, 0, 5, 4]), (8, 128, 64, [2, 1, 0, 3, 4, 0, 5, 2])]):
        torch.manual_seed(1340 + ci)
        up = (torch.randn(experts, 2 * inter, hidden) * 0.02).to(torch.bfloat16)
        down = (torch.randn(experts, hidden, inter) * 0.02).to(torch.bfloat16)
        x = torch.rand(sum(counts), hidden, dtype=torch.bfloat16)
        recs.append(run_case("MojoExperts", {"kwargs": {"num_experts": experts, "hidden_size": hidden, "intermediate_size": inter}},
                             {"up_proj_weight": up, "down_proj_weight": down},
                             (x, torch.tensor(counts, dtype=torch.int32)), {}, cast=torch.bfloat16))
    for ci, (tokens, k, hidden, by_gates, dtype) in enumerate([(64, 4, 256, True, torch.bfloat16), (33, 2, 96, True, torch.float16),
                                                               (16, 8, 128, False, torch.bfloat16), (5, 3, 64, True, torch.float32)]):
        torch.manual_seed(1360 + ci)
        n = tokens * k
        perm = torch.randperm(n)
        outs = torch.randn(n, hidden, dtype=dtype)[perm].contiguous()
        gates = torch.rand(n, 1)[perm].contiguous()
        tok = torch.arange(tokens, dtype=torch.int32).unsqueeze(1).expand(-1, k).reshape(-1)[perm].contiguous()
        if ci == 1:                      # an EP-style slice: not every token appears, some appear once
            outs, gates, tok = outs[: n // 2], gates[: n // 2], tok[: n // 2]
        recs.append(run_case("MojoMoECombine", {"kwargs": {"multiply_by_gates": by_gates}}, {},
                             (torch.zeros(tokens, hidden, dtype=dtype), outs, gates, tok), {}))
    return recs


def gen_moe_layer():
    """The composite MojoMoE (gating -> dispatch -> experts -> combine, `core/operators/moe.py:12-130`) set up the way the
    reference's own test does (test_moe.py:73-103): bf16 layer, fp32 router weight, normal(0.02) parameters."""
    recs = []
    for experts, k, hidden, inter, tokens, std, seed in [(16, 4, 128, 64, 64, 0.2, 1380), (8, 2, 128, 64, 37, 0.2, 1381),
                                                         (64, 8, 128, 32, 48, 0.5, 1452), (4, 1, 64, 32, 1, 0.2, 1383)]:
        torch.manual_seed(seed)                   # seeds picked so that no routing decision sits on a near-tie
        w = torch.randn(hidden, experts) * std
        x = torch.rand(tokens, hidden, dtype=torch.bfloat16)
        probs = torch.softmax(x.float() @ w, dim=-1).sort(dim=-1, descending=True).values
        if k < experts:
            assert float((probs[:, :k] - probs[:, 1: k + 1]).min()) > 1e-5, "near-tie in the gating vector: pick another seed"
        state = {"gating.gate_weight": w,
                 "experts.up_proj_weight": (torch.randn(experts, 2 * inter, hidden) * 0.02).to(torch.bfloat16),
                 "experts.down_proj_weight": (torch.randn(experts, hidden, inter) * 0.02).to(torch.bfloat16)}
        recs.append(run_case("MojoMoE", {"kwargs": {"num_experts": experts, "top_k": k, "hidden_size": hidden,
                                                    "intermediate_size": inter}}, state, (x,), {}, cast=torch.bfloat16,
                             keep_dtype=("gating.gate_weight",)))
    return recs


def gen_quantizers():
    """MojoDynamicQuant and MojoResidualAddRMSNormQuant on the reference's test shapes (test_quantize.py:41-50,
    test_normalization.py:442-446) plus zero rows, a non-multiple-of-8 width and the fp8 branch."""
    recs = []
    for ci, (shape, dtype, smooth) in enumerate([((1, 128), torch.bfloat16, True), ((17, 320), torch.float16, True),
                                                 ((3, 129), torch.bfloat16, True), ((7, 257), torch.float16, False),
                                                 ((2, 5, 512), torch.bfloat16, True), ((48, 1536), torch.float32, False)]):
        torch.manual_seed(1400 + ci)
        x = torch.randn(shape, dtype=dtype)
        x.view(-1, shape[-1])[0].zero_()                       # an all-zero token: scale falls back to 1.0
        state = {"inv_smooth_scale": 1.0 / (torch.rand(shape[-1]) + 0.1)} if smooth else {}
        recs.append(run_case("MojoDynamicQuant", {"kwargs": {"input_size": shape[-1] if smooth else None}}, state, (x,), {}))
    for ci, (shape, dtype, pos, qd, smooth) in enumerate([
            ((32, 1024), torch.bfloat16, "pre", torch.int8, False), ((32, 1024), torch.float16, "post", torch.int8, False),
            ((2, 256), torch.bfloat16, "pre", torch.int8, True), ((5, 3, 200), torch.float16, "pre", torch.int8, False),
            ((8, 1024), torch.bfloat16, "pre", torch.float8_e4m3fn, False), ((8, 1024), torch.float16, "post", torch.float8_e4m3fn, True)]):
        torch.manual_seed(1420 + ci)
        x, r = torch.randn(shape, dtype=dtype), torch.randn(shape, dtype=dtype)
        w = torch.randn(shape[-1])
        kwargs = {"smooth_scale": torch.rand(shape[-1]) + 0.5} if smooth else {}
        recs.append(run_case("MojoResidualAddRMSNormQuant",
                             {"kwargs": {"norm_size": shape[-1], "norm_pos": pos, "quant_dtype": qd}}, {"weight": w}, (x, r), kwargs))
    return recs


def gen_store_mla():
    """MojoStorePagedMLAKVCache: prefill and decode, ragged, padding (-1 context), a hole in the table (the store of that
    sequence ends there), a sequence whose first page is missing, tokens running past the table."""
    recs = []
    specs = [
        # (page, r, rope, q_lens | None for decode, ctx, table rows)
        (16, 64, 32, [5, 0, 40, 17], [3, -1, 0, 30], None),
        (8, 32, 16, None, [0, 7, 8, -1, 23], None),
        (16, 128, 64, [33, 20, 9], [10, 0, 15], "holes"),
        (4, 16, 8, [30], [2], "short"),
    ]
    for ci, (page, r, rope, q_lens, ctx, mode) in enumerate(specs):
        torch.manual_seed(1500 + ci)
        batch = len(ctx)
        new = [1] * batch if q_lens is None else q_lens
        need = [max((max(c, 0) + n + page - 1) // page, 1) for c, n in zip(ctx, new)]
        width = max(need) + 1
        total = sum(need) + 4
        ids = torch.randperm(total, dtype=torch.int32)
        table = torch.full((batch, width), -1, dtype=torch.int32)
        at = 0
        for b, n in enumerate(need):
            table[b, :n] = ids[at: at + n]
            at += n
        if mode == "holes":
            table[0, 1] = -1                 # sequence 0 stops after its first page
            table[1, 0] = -1                 # sequence 1 is skipped entirely
        if mode == "short":
            table = table[:, :3].contiguous()   # 30 tokens from position 2 need 8 pages of 4: only 3 exist
        tokens = sum(new)
        ckv, kpe = torch.randn(tokens, r, dtype=torch.bfloat16), torch.randn(tokens, rope, dtype=torch.bfloat16)
        ckv_cache = torch.randn(total, 1, page, r, dtype=torch.bfloat16)
        kpe_cache = torch.randn(total, 1, page, rope, dtype=torch.bfloat16)
        cu_q = None if q_lens is None else cu(q_lens)
        recs.append(run_case("MojoStorePagedMLAKVCache", {}, {},
                             (ckv, kpe, ckv_cache, kpe_cache, table, cu_q, torch.tensor(ctx, dtype=torch.int32)), {}))
    return recs


def quantize_rows(x):
    scale = x.abs().amax(dim=-1).clamp_min(1e-8) / 127.0
    return torch.clamp(torch.round(x / scale.unsqueeze(-1)), -128, 127).to(torch.int8), scale


def gen_quant_gemm():
    recs = []
    for ci, (m, k, n, trans, odt) in enumerate([
        (1, 256, 128, False, torch.bfloat16), (32, 512, 96, True, torch.bfloat16),
        (17, 128, 64, False, torch.float16), (8, 1024, 48, True, torch.float32),
    ]):
        torch.manual_seed(900 + ci)
        xq, xs = quantize_rows(torch.randn(m, k))
        wq, ws = quantize_rows(torch.randn(n, k))            # [N,K] int8, per-channel scale
        w = wq if trans else wq.t().contiguous()
        s1 = xs if ci % 2 == 0 else xs.unsqueeze(-1)
        recs.append(run_case("MojoQuantGemm",
                             {"kwargs": dict(in_features=k, out_features=n, output_dtype=odt, trans_weight=trans)},
                             {"weight": w, "weight_scale": ws.to(torch.bfloat16)}, (xq, s1), {}))
    return recs


# --------------------------------------------------------------------------------------------
# gemm + collective: reference classes over gloo, 2 processes
# --------------------------------------------------------------------------------------------
def _comm_worker(rank, ws, port, spec, ret):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    out = {}
    for name, op_name, ctor_kw, m, k, n, dtype in spec:
        torch.manual_seed(42 + rank)
        x = torch.randn(m, k, dtype=dtype)
        w = torch.randn(k, n, dtype=dtype) * 0.1
        res = {}
        for which, cls in (("ref", ref_cls(op_name)), ("mine", my_cls(op_name))):
            op = cls(w.clone(), None, True, **ctor_kw)
            y = op.forward(x.clone())
            if hasattr(y, "wait"):
                y = y.wait()
            res[which] = torch.as_tensor(y).clone()
        assert same(res["ref"], res["mine"]), f"{name}: restatement differs from reference on rank {rank}"
        out[name] = {"x": x, "w": w, "out": res["ref"]}
    ret[rank] = out
    dist.destroy_process_group()


def gen_comm():
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    spec = [
        ("allreduce_bf16", "MojoGemmAllReduce", {}, 24, 64, 48, torch.bfloat16),
        ("allreduce_f32", "MojoGemmAllReduce", {}, 8, 32, 16, torch.float32),
        ("allgather_bf16", "MojoAllGatherGemm", {"gather_dim": 0}, 12, 64, 40, torch.bfloat16),
        ("reducescatter_bf16", "MojoGemmReduceScatter", {"scatter_dim": 0}, 24, 64, 48, torch.bfloat16),
    ]
    # world sizes 2, 4 and 8 (round 5: the reference's own harness runs these operators on 8 ranks,
    # tests/dist_common.py:38-81; the fixtures used to stop at 2).  Every row count is divisible by 8.
    cases = []
    for ws in (2, 4, 8):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        mgr = mp.Manager()
        ret = mgr.dict()
        mp.spawn(_comm_worker, args=(ws, port, spec, ret), nprocs=ws, join=True)
        tag = "" if ws == 2 else f"_ws{ws}"
        for name, op_name, ctor_kw, m, k, n, dtype in spec:
            cases.append({"name": name + tag, "op": op_name, "ctor_kwargs": ctor_kw, "world_size": ws,
                          "ranks": [dict(ret[r][name]) for r in range(ws)]})
        # all-to-all: closed form the reference test builds without communication
        # (tests/accuracy/operators/test_compute_with_comm.py:226-232)
        m, k, n = 32, 64, 128
        ranks = []
        ys = []
        for r in range(ws):
            torch.manual_seed(42 + r)
            x = torch.randn(m, k)
            w = torch.randn(k, n) * 0.1
            ys.append(x @ w)
            ranks.append({"x": x, "w": w})
        for r in range(ws):
            ranks[r]["out"] = torch.cat([ys[src].chunk(ws, dim=0)[r] for src in range(ws)], dim=0)
        cases.append({"name": "all2all_f32" + tag, "op": "MojoGemmAll2All", "ctor_kwargs": {"scatter_dim": 0, "gather_dim": 0},
                      "world_size": ws, "ranks": ranks})
    return cases


def gen_paged_cache():
    """The reference's `PagedDummyCache` (modeling/qwen3/mojo_qwen3_dense.py:41-135) driven through a prefill and decode
    steps on two layers; records every update's inputs and the state after it.  The restatement
    (oracle/paged_cache_ref.py) must end in the identical state after every step."""
    from types import SimpleNamespace

    from mojo_opset.modeling.qwen3.mojo_qwen3_dense import PagedDummyCache as RefCache

    from oracle.paged_cache_ref import PagedDummyCacheRef

    recs = []
    for ci, (layers, heads, dim, max_pos, batch, page, steps) in enumerate([
        (2, 2, 16, 64, 3, 8, [5, 1, 1, 1, 1, 7, 1]), (1, 4, 32, 96, 5, 16, [16, 1, 1, 33, 1]), (3, 1, 8, 40, 2, 4, [3, 1, 1, 1, 1, 1, 1]),
    ]):
        torch.manual_seed(1700 + ci)
        cfg = SimpleNamespace(num_hidden_layers=layers, num_key_value_heads=heads, head_dim=dim, max_position_embeddings=max_pos)
        ref = RefCache(cfg, batch, "cpu", block_size=page)
        mine_ = PagedDummyCacheRef(layers, heads, dim, max_pos, batch, block_size=page)
        trace = []
        for si, new_len in enumerate(steps):
            for layer in range(layers):
                k = torch.randn(batch, heads, new_len, dim).to(torch.bfloat16)
                v = torch.randn(batch, heads, new_len, dim).to(torch.bfloat16)
                ref.update(k.clone(), v.clone(), layer)
                mine_.update(k.clone(), v.clone(), layer)
                for a, b in ((ref.block_tables, mine_.block_tables), (ref.seq_lens, mine_.seq_lens), (ref.k_cache, mine_.k_cache),
                             (ref.v_cache, mine_.v_cache)):
                    if not torch.equal(a, b):
                        raise SystemExit("restatement of PagedDummyCache is NOT identical to the reference")
                assert ref.num_free_blocks == mine_.num_free_blocks
                dk, dv, dt = ref.get_kv_for_decode(layer)
                mk, mv, mt = mine_.get_kv_for_decode(layer)
                assert torch.equal(dt, mt)
                trace.append({"layer": layer, "k": k, "v": v, "block_tables": ref.block_tables.clone(), "seq_lens": ref.seq_lens.clone(),
                              "num_free": ref.num_free_blocks, "decode_table": dt.clone()})
        recs.append({"op": "PagedDummyCache", "config": dict(num_hidden_layers=layers, num_key_value_heads=heads, head_dim=dim,
                                                             max_position_embeddings=max_pos),
                     "batch": batch, "block_size": page, "trace": trace, "k_cache": ref.k_cache.clone(), "v_cache": ref.v_cache.clone()})
    # exhaustion: the reference raises ValueError (:78-79)
    cfg = SimpleNamespace(num_hidden_layers=1, num_key_value_heads=1, head_dim=8, max_position_embeddings=8)
    ref = RefCache(cfg, 2, "cpu", block_size=4)
    ref.update(torch.zeros(2, 1, 8, 8, dtype=torch.bfloat16), torch.zeros(2, 1, 8, 8, dtype=torch.bfloat16), 0)
    try:
        ref.update(torch.zeros(2, 1, 1, 8, dtype=torch.bfloat16), torch.zeros(2, 1, 1, 8, dtype=torch.bfloat16), 0)
        raise SystemExit("reference did not raise on exhaustion")
    except ValueError as e:
        recs.append({"op": "PagedDummyCache.oom", "message": str(e)})
    return recs


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(4)
    makers = {
        "paged_decode_gqa": gen_decode_gqa, "paged_prefill_gqa": gen_prefill_gqa, "paged_mla": gen_mla, "rmsnorm": gen_norm,
        "swiglu": gen_swiglu, "rope": gen_rope, "group_gemm": gen_group_gemm, "dense": gen_dense, "quant_gemm": gen_quant_gemm, "moe": gen_moe,
        "moe_layer": gen_moe_layer, "quantizers": gen_quantizers, "store_paged_mla": gen_store_mla,
        "compute_with_comm": gen_comm, "paged_cache": gen_paged_cache,
    }
    only = set(sys.argv[1:])                       # `python oracle/make_golden.py moe_layer` regenerates just that group
    groups = {name: make() for name, make in makers.items() if not only or name in only}
    if not only or only & {"store_paged_kv", "kv_plan"}:
        groups["store_paged_kv"], groups["kv_plan"] = gen_store_kv()
    meta = {"torch": torch.__version__, "reference": "XPU-Forces/mojo_opset @ /root/reference (0.0.3.post27)"}
    for name, recs in groups.items():
        path = os.path.join(OUT, f"{name}.pt")
        torch.save({"meta": meta, "cases": recs}, path)
        print(f"{name:22s} {len(recs):3d} cases  {os.path.getsize(path) / 1e6:6.2f} MB")


if __name__ == "__main__":
    main()
