"""Secondary measurements reported under ``extras`` on bench.py's JSON line: the other ops of the hot path
at the BASELINE shapes, as absolute rate and fraction of the roofline that bounds them (HBM 8 TB/s, bf16
MFMA 2.5 PFLOP/s dense, int8 5 POP/s, fp8 5 PFLOP/s).  Timing: HIP events on the launch stream, >= 10 launches after warm-up.
With world > 1 every rank runs the GEMM + collective cases together (they contain collectives)."""

import os

import torch
import torch.distributed as dist

import mojo_opset_amd as mo

HBM_PEAK_GBS = 8000.0
MFMA_BF16_PEAK_TFLOPS = 2500.0
MFMA_I8_PEAK_TOPS = 5000.0
MFMA_FP8_PEAK_TFLOPS = 5000.0


REPEATS = 5            # every figure is the MEDIAN of this many timed regions (min and max are kept beside it)
_LAST = {}             # statistics of the most recent _time / _time_graph call, picked up by _mfma / _hbm


def _summarise(samples):
    """Median of the repeats (seconds); min / max / count remembered for the record built from it.  One mean over a single
    region let a 3x outlier (a neighbour's clock ramp, a stray host stall) into a committed evidence file in round 2."""
    xs = sorted(samples)
    med = xs[len(xs) // 2] if len(xs) % 2 else 0.5 * (xs[len(xs) // 2 - 1] + xs[len(xs) // 2])
    _LAST.clear()
    _LAST.update({"t": med, "us_min": xs[0] * 1e6, "us_max": xs[-1] * 1e6, "repeats": len(xs)})
    return med


def _stats(t):
    return {k: v for k, v in _LAST.items() if k != "t"} if _LAST.get("t") == t else {}


def _time(fn, iters=10, warmup=2, settle_s=0.03, settle_n=None, repeats=REPEATS):
    """Eager launches timed with HIP events, after ``warmup`` calls and ``settle_s`` of back-to-back device work (the
    power-management transient after an idle moment, see _time_graph, lasts 10-30 ms).  ``settle_n`` fixes the number of
    settle calls instead: a case that contains collectives must make the SAME number of calls on every rank, and a count
    derived from a rank's own clock would not.  ``repeats`` back-to-back regions of ``iters`` calls; returns their median."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    one = max(e0.elapsed_time(e1) * 1e-3, 1e-6)
    for _ in range(min(200, int(settle_s / one)) if settle_n is None else settle_n):
        fn()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(repeats + 1)]
    evs[0].record()
    for r in range(repeats):
        for _ in range(iters):
            fn()
        evs[r + 1].record()
    torch.cuda.synchronize()
    return _summarise([evs[r].elapsed_time(evs[r + 1]) / iters * 1e-3 for r in range(repeats)])


def _time_graph(fn, reps=20, replays=5, settle_s=0.03, min_timed_s=0.01):
    """Device time of ``fn`` with launch overhead amortised: ``reps`` calls captured in one HIP graph, replayed.  For
    decode-sized ops the eager figure measures the Python shim (~20 us per call), not the kernels; a serving loop
    replays graphs, so this is the figure that matters there.

    SUSTAINED rate: the graph is first replayed for ``settle_s`` of device time without a pause, then timed over at least
    ``min_timed_s``.  After an idle moment (capture, allocation, a host sync) the chip runs a power-management transient:
    the same decode kernel takes 52 us for the first ~40 back-to-back launches, 60-67 us between 2 and 8 ms, and is back at
    52 us from ~10 ms on (ctx 1024; at ctx 4096: 209-217 us at first, 179 us after ~28 ms) — profiles/r2_decode_sustain.txt,
    scripts/probes/decode_sustain.py.  Timing 100 launches right after the capture measured that transient."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(reps):
            fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    graph.replay()
    e1.record()
    torch.cuda.synchronize()
    one = max(e0.elapsed_time(e1) * 1e-3, 1e-6)                       # seconds per replay (first, cold estimate)
    settle = min(400, max(1, int(settle_s / one)))
    replays = min(400, max(replays, int(min_timed_s / one) + 1))
    for _ in range(settle):
        graph.replay()
    per = max(1, -(-replays // REPEATS))                                # replays per timed region
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(REPEATS + 1)]
    evs[0].record()
    for r in range(REPEATS):
        for _ in range(per):
            graph.replay()
        evs[r + 1].record()
    torch.cuda.synchronize()
    return _summarise([evs[r].elapsed_time(evs[r + 1]) / (reps * per) * 1e-3 for r in range(REPEATS)])


def hip(name):
    return getattr(mo, name).get_backend_impl("hip", strict=True)


def _want(case_name):
    """MOJO_BENCH_ONLY=<substring>: run only the cases whose name contains it (one case per profiled process, so the rows of
    a kernel trace belong to one case)."""
    only = os.environ.get("MOJO_BENCH_ONLY")
    return not only or only in case_name


def _mfma(t, flops, peak=MFMA_BF16_PEAK_TFLOPS):
    tf = flops / t / 1e12
    return {"us": t * 1e6, "tflops": tf, "frac_of_mfma_peak": tf / peak, **_stats(t)}


def _hbm(t, nbytes):
    gbs = nbytes / t / 1e9
    return {"us": t * 1e6, "GB/s": gbs, "frac_of_hbm_peak": gbs / HBM_PEAK_GBS, **_stats(t)}


def group_gemm_case(device, m, k, n, groups, trans, split="balanced", dtype=torch.bfloat16, data="randn"):
    """data: "randn" (the figure to quote), "zeros" (same instruction stream, least switching power: shows how much of
    the gap to the MFMA peak is the clock the chip holds under load), or "torch" (torch.matmul = hipBLASLt per group,
    a calibration point for what a tuned library GEMM reaches on this chip — not a product path)."""
    make = torch.zeros if data == "zeros" else torch.randn
    x = make(m, k, device=device, dtype=dtype)
    w = make(groups, n, k, device=device, dtype=dtype) if trans else make(groups, k, n, device=device, dtype=dtype)
    if data == "torch":
        per = m // groups
        wt = w.transpose(1, 2) if trans else w
        return _mfma(_time(lambda: [torch.matmul(x[g * per:(g + 1) * per], wt[g]) for g in range(groups)]), 2.0 * m * k * n)
    if split == "balanced":
        counts = torch.full((groups,), m // groups, dtype=torch.int32)
    else:  # one expert takes half of the rows
        counts = torch.full((groups,), (m // 2) // (groups - 1), dtype=torch.int32)
        counts[0] = m - int(counts[1:].sum())
    counts[-1] += m - int(counts.sum())
    counts = counts.to(device)
    op = hip("MojoGroupGemm")(w, trans)
    return _mfma(_time(lambda: op(x, counts)), 2.0 * m * k * n)


def bench_group_gemm(device):
    out = {}
    for name, (m, k, n, g, trans, split) in {
        "ref_case_20480x4096x4096_G8_KN": (20480, 4096, 4096, 8, False, "balanced"),
        "mixtral_up_16384x4096x28672_G8_KN": (16384, 4096, 28672, 8, False, "balanced"),
        "mixtral_up_16384x4096x28672_G8_NK": (16384, 4096, 28672, 8, True, "balanced"),
        "mixtral_down_16384x14336x4096_G8_KN": (16384, 14336, 4096, 8, False, "balanced"),
        "mixtral_up_16384_skewed_KN": (16384, 4096, 28672, 8, False, "skewed"),
        "mixtral_up_4096x4096x28672_G8_KN": (4096, 4096, 28672, 8, False, "balanced"),
    }.items():
        if not _want(name):
            continue
        out[name] = group_gemm_case(device, m, k, n, g, trans, split)
        torch.cuda.empty_cache()
    # Calibration points for the headline shape ON THIS BOX, in this run (the "at the vendor library / clock-
    # limited" claim belongs on a driver-run record): the same instruction stream on ZERO operands (least switching power: the
    # clock the chip would hold if the data cost nothing) and hipBLASLt through torch.matmul, one product per expert.
    if _want("mixtral_up_16384x4096x28672_G8_KN_calibration"):
        try:
            cal = {}
            cal["zero_operand_tflops"] = group_gemm_case(device, 16384, 4096, 28672, 8, False, data="zeros")["tflops"]
            torch.cuda.empty_cache()
            cal["hipblaslt_same_box_tflops"] = group_gemm_case(device, 16384, 4096, 28672, 8, False, data="torch")["tflops"]
            torch.cuda.empty_cache()
            cal["hipblaslt_same_box_tflops_NK"] = group_gemm_case(device, 16384, 4096, 28672, 8, True, data="torch")["tflops"]
            cal["note"] = ("zero operands: this kernel, same launch, all-zero data; hipBLASLt: torch.matmul per expert on random data — "
                           "calibration, not product paths")
            out["mixtral_up_16384x4096x28672_G8_KN_calibration"] = cal
        except Exception as e:
            out["mixtral_up_16384x4096x28672_G8_KN_calibration"] = {"error": repr(e)}
        torch.cuda.empty_cache()
    return out


def bench_quant_gemm(device):
    out = {}
    for qname, qd, peak in (("int8", torch.int8, MFMA_I8_PEAK_TOPS), ("fp8_e4m3", torch.float8_e4m3fn, MFMA_FP8_PEAK_TFLOPS)):
        for m, k, n in ((4096, 7168, 36864), (4096, 18432, 7168), (128, 7168, 4096), (32, 7168, 4096), (1, 7168, 4096),
                        (32, 18432, 7168)):
            if not _want(f"{qname}_{m}x{k}x{n}_NK"):
                continue
            op = hip("MojoQuantGemm")(k, n, trans_weight=True, quant_dtype=qd, weight_dtype=qd, device=device)
            if qd == torch.int8:
                op.weight.copy_(torch.randint(-127, 128, (n, k), dtype=torch.int8, device=device))
                x = torch.randint(-127, 128, (m, k), dtype=torch.int8, device=device)
            else:
                op.weight.copy_(torch.randn(n, k, device=device).to(qd))
                x = torch.randn(m, k, device=device).to(qd)
            op.weight_scale.fill_(0.01)
            s = torch.rand(m, device=device)
            t = _time(lambda: op(x, s), 20 if m <= 128 else 10, 3)
            res = _mfma(t, 2.0 * m * k * n, peak)
            if m <= 128:                                   # decode-sized M: the weight stream is the roofline
                tg = _time_graph(lambda: op(x, s))
                res = {"us_eager": t * 1e6, **_hbm(tg, k * n + m * k + m * n * 2)}
            if m >= 1024:
                # calibration: the vendor library's 8-bit product on the same box in the same run — torch._int_mm (int32 out, NO
                # dequantising epilogue) / torch._scaled_mm (fp8, unit scales, bf16 out), both hipBLASLt
                try:
                    wl = op.weight
                    if qd == torch.int8:
                        t_lib = _time_graph(lambda: torch._int_mm(x, wl.t()), reps=3, replays=3)
                    else:
                        one = torch.ones((), device=device)
                        t_lib = _time_graph(lambda: torch._scaled_mm(x, wl.t(), scale_a=one, scale_b=one, out_dtype=torch.bfloat16), reps=3, replays=3)
                    res.update({"hipblaslt_same_box_us": t_lib * 1e6, "time_vs_hipblaslt": t / t_lib})
                except Exception as e:
                    res["hipblaslt_same_box_us"] = repr(e)[:80]
            out[f"{qname}_{m}x{k}x{n}_NK"] = res
            del op, x
            torch.cuda.empty_cache()
    return out


def _paged(device, lens, hkv, d, page, dtype=torch.bfloat16):
    need = [(n + page - 1) // page for n in lens]
    total = sum(need) + 4
    k = torch.randn(total, hkv, page, d, device=device, dtype=dtype)
    v = torch.randn(total, hkv, page, d, device=device, dtype=dtype)
    perm = torch.randperm(total, dtype=torch.int32)
    table = torch.full((len(lens), max(need)), -1, dtype=torch.int32)
    at = 0
    for b, n in enumerate(need):
        table[b, :n] = perm[at: at + n]
        at += n
    return k, v, table.to(device)


def bench_decode_variants(device):
    """MojoPagedDecodeGQA at the other context lengths of SURVEY §8(d) config 2 (the headline is uniform ctx = 4096):
    uniform 1024 / 16384 and ragged randint(ctx/2, ctx), seed 20260716; random page permutation, -1 padded tables."""
    hq, hkv, d, page, bsz = 32, 8, 128, 16, 64
    op = hip("MojoPagedDecodeGQA")(is_causal=True, gqa_layout="AABB")
    g = torch.Generator().manual_seed(20260716)
    out = {}
    for name, lens in {"uniform_ctx1024": [1024] * bsz, "uniform_ctx16384": [16384] * bsz,
                       "ragged_ctx2048_4096": torch.randint(2048, 4097, (bsz,), generator=g).tolist(),
                       "ragged_ctx8192_16384": torch.randint(8192, 16385, (bsz,), generator=g).tolist()}.items():
        if not _want(name):
            continue
        sets = []
        for _ in range(2 if max(lens) <= 4096 else 1):                      # 2 x >= 256 MB defeats the MALL at the short contexts
            k, v, table = _paged(device, lens, hkv, d, page)
            q = torch.randn(bsz, hq, d, device=device, dtype=torch.bfloat16)
            sets.append((q, k, v, torch.tensor(lens, dtype=torch.int32, device=device), table))
        it = [0]

        def step():
            q, k, v, ln, tb = sets[it[0] % len(sets)]
            it[0] += 1
            return op(q, k, v, ln, tb, max_total_seq_len=max(lens))
        t = _time(step, 60, 40)                                             # long warm-up: the first ~50 launches run through the clock transient
        nbytes = sum(lens) * hkv * d * 2 * 2 + 2 * bsz * hq * d * 2 + 4 * bsz * (sets[0][4].shape[1] + 1)
        t_eager = t
        if max(lens) <= 1024:            # a launch this short is within reach of the Python shim's per-call cost: quote the
            it[0] = 0                    # captured-graph figure (what a serving loop replays) and keep the eager one beside it
            t = min(t, _time_graph(step, reps=20, replays=10))
        res = _hbm(t, nbytes)
        res["us_eager"] = t_eager * 1e6
        res["tokens_per_s"] = bsz / t
        out[name] = res
        del sets
        torch.cuda.empty_cache()
    return out


def bench_decode_geometries(device):
    """MojoPagedDecodeGQA away from the headline head geometry: Llama-3-70B (64 q / 8 kv heads: groups of EIGHT query heads
    per kv head), its per-rank shape under TP 8 (8 / 1), and head_dim 64.  B = 64, ctx 4096, page 16, bf16, graph replay."""
    out = {}
    page, bsz, ctx = 16, 64, 4096
    for name, (hq, hkv, d) in {"G8_llama3_70b_64q_8kv_d128_ctx4096": (64, 8, 128), "G8_tp8_8q_1kv_d128_ctx4096": (8, 1, 128),
                               "G4_32q_8kv_d64_ctx4096": (32, 8, 64)}.items():
        if not _want(name):
            continue
        op = hip("MojoPagedDecodeGQA")(is_causal=True, gqa_layout="AABB")
        lens = [ctx] * bsz
        sets = []
        for _ in range(2):
            k, v, table = _paged(device, lens, hkv, d, page)
            q = torch.randn(bsz, hq, d, device=device, dtype=torch.bfloat16)
            sets.append((q, k, v, torch.tensor(lens, dtype=torch.int32, device=device), table))
        it = [0]

        def step():
            q, k, v, ln, tb = sets[it[0] % len(sets)]
            it[0] += 1
            return op(q, k, v, ln, tb, max_total_seq_len=ctx)
        t = _time_graph(step, reps=10, replays=10)
        nbytes = sum(lens) * hkv * d * 2 * 2 + 2 * bsz * hq * d * 2 + 4 * bsz * (sets[0][4].shape[1] + 1)
        res = _hbm(t, nbytes)
        res["tokens_per_s"] = bsz / t
        out[name] = res
        del sets
        torch.cuda.empty_cache()
    return out


def bench_prefill(device):
    out = {}
    hq, hkv, d, page = 32, 8, 128, 16
    op = hip("MojoPagedPrefillGQA")()
    g = torch.Generator().manual_seed(20260716)
    ragged = torch.randint(512, 1025, (16,), generator=g).tolist()
    for name, (q_lens, cached) in {"4x2048_nocache": ([2048] * 4, [0] * 4), "4x2048_cached2048": ([2048] * 4, [2048] * 4),
                                   "16_ragged_512_1024_nocache": (ragged, [0] * 16),
                                   "1x16384_nocache": ([16384], [0]),
                                   "chunked_1x512_cached16384": ([512], [16384])}.items():
        if not _want(name):
            continue
        kv = [a + b for a, b in zip(q_lens, cached)]
        k, v, table = _paged(device, kv, hkv, d, page)
        q = torch.randn(sum(q_lens), hq, d, device=device, dtype=torch.bfloat16)
        cu = lambda l: torch.tensor([0] + list(torch.tensor(l).cumsum(0).tolist()), dtype=torch.int32, device=device)  # noqa: E731
        cu_q, cu_kv = cu(q_lens), cu(kv)
        flops = sum(4.0 * hq * d * (a * b - a * a / 2.0) for a, b in zip(q_lens, kv))
        t = _time(lambda: op(q, k, v, cu_q, table, cu_total_seq_lens=cu_kv, max_q_len=max(q_lens), max_total_seq_len=max(kv)))
        out[name] = _mfma(t, flops)
        if len(set(q_lens)) == 1 and not any(cached):
            # calibration: torch's own scaled_dot_product_attention on the same box (the flash kernel of the PyTorch-ROCm wheel,
            # contiguous [B, H, S, D] tensors, enable_gqa) — what a user gets without this backend
            try:
                import torch.nn.functional as F
                b_, s_ = len(q_lens), q_lens[0]
                qs = torch.randn(b_, hq, s_, d, device=device, dtype=torch.bfloat16)
                ks = torch.randn(b_, hkv, s_, d, device=device, dtype=torch.bfloat16)
                vs = torch.randn(b_, hkv, s_, d, device=device, dtype=torch.bfloat16)
                t_lib = _time(lambda: F.scaled_dot_product_attention(qs, ks, vs, is_causal=True, enable_gqa=True), 5, 2)
                out[name].update({"torch_sdpa_same_box_us": t_lib * 1e6, "time_vs_torch_sdpa": t / t_lib})
                del qs, ks, vs
            except Exception as e:   # an SDPA build without this path must not cost the record
                out[name]["torch_sdpa_same_box_us"] = repr(e)[:80]
    return out


def bench_mla_decode(device):
    b, h, nope, rope, vd, r, page, ctx = 64, 128, 128, 64, 128, 512, 16, 4096
    op = hip("MojoPagedDecodeMLA")(h, nope, rope, vd, r).to(torch.bfloat16).to(device)
    with torch.no_grad():
        op.kv_b_proj.copy_(torch.randn_like(op.kv_b_proj) * 0.02)
    pages = ctx // page
    total = b * pages + 4
    ckv = torch.randn(total, 1, page, r, device=device, dtype=torch.bfloat16)
    kpe = torch.randn(total, 1, page, rope, device=device, dtype=torch.bfloat16)
    table = torch.randperm(total, dtype=torch.int32)[: b * pages].view(b, pages).to(device)
    if os.environ.get("MOJO_BENCH_MLA_SEQ"):          # diagnostic: physically sequential pages
        table = torch.arange(b * pages, dtype=torch.int32).view(b, pages).to(device)
    lens = torch.full((b,), ctx, dtype=torch.int32, device=device)
    q = torch.randn(b, h, nope + rope, device=device, dtype=torch.bfloat16)
    t = _time(lambda: op(q, ckv, kpe, lens, table))
    nbytes = b * ctx * (r + rope) * 2 + h * (nope + vd) * r * 2 + 2 * b * h * (nope + rope + vd) * 2
    res = _hbm(t, nbytes)
    flops = 2.0 * b * h * ctx * (2 * r + rope) + 2.0 * b * h * r * (nope + vd) * 2
    res.update({"tflops": flops / t / 1e12, "tokens_per_s": b / t})
    out = {"B64_H128_ctx4096_page16": res}
    if _want("golden_route"):
        # the PARITY route (the golden's own arithmetic: decompressed K / V for every cached token, sequences walked in slices
        # of the decompression budget) timed once beside the default: it is not a fast path and is not meant to be
        op.decode_route = "golden"
        tg = _time(lambda: op(q, ckv, kpe, lens, table, max_total_seq_len=ctx), iters=2, warmup=1, repeats=3)
        op.decode_route = None
        out["B64_H128_ctx4096_page16_golden_route"] = {"us": tg * 1e6, "tokens_per_s": b / tg, **_stats(tg),
                                                       "note": "parity route (MOJO_HIP_MLA_DECODE=golden), not the default"}
    return out


def bench_mla_prefill(device):
    """MojoPagedPrefillMLA, DeepSeek-V3 dims: 4 sequences x 512 new tokens, with and without 2048 cached tokens."""
    h, nope, rope, vd, r, page = 128, 128, 64, 128, 512, 16
    op = hip("MojoPagedPrefillMLA")(h, nope, rope, vd, r).to(torch.bfloat16).to(device)
    with torch.no_grad():
        op.kv_b_proj.copy_(torch.randn_like(op.kv_b_proj) * 0.02)
    out = {}
    for name, (q_lens, cached) in {"4x512_nocache": ([512] * 4, [0] * 4), "4x512_cached2048": ([512] * 4, [2048] * 4)}.items():
        if not _want(name):
            continue
        kv = [a + b for a, b in zip(q_lens, cached)]
        need = [(n + page - 1) // page for n in kv]
        total = sum(need) + 4
        ckv = torch.randn(total, 1, page, r, device=device, dtype=torch.bfloat16)
        kpe = torch.randn(total, 1, page, rope, device=device, dtype=torch.bfloat16)
        table = torch.randperm(total, dtype=torch.int32)[: sum(need)].view(len(kv), need[0]).to(device)
        cu = lambda l: torch.tensor([0] + torch.tensor(l).cumsum(0).tolist(), dtype=torch.int32, device=device)  # noqa: E731
        cu_q, cu_kv = cu(q_lens), cu(kv)
        q = torch.randn(sum(q_lens), h, nope + rope, device=device, dtype=torch.bfloat16)
        t = _time(lambda: op(q, ckv, kpe, cu_q, table, cu_total_seq_lens=cu_kv), 5, 1)
        vis = sum(a * b - a * (a - 1) / 2.0 for a, b in zip(q_lens, kv))          # visible (query, key) pairs
        # FLOPs of the formulation that runs (the golden's own: decompress every key once, then D_qk = 192 / D_v = 128
        # attention) and, for continuity with round 1, of the weight-absorbed formulation of the same result
        flops = 2.0 * sum(kv) * r * h * (nope + vd) + 2.0 * h * vis * (nope + rope + vd)
        flops_absorbed = 2.0 * h * vis * (2 * r + rope) + 2.0 * sum(q_lens) * h * r * (nope + vd)
        res = _mfma(t, flops)
        res["absorbed_form_equivalent_tflops"] = flops_absorbed / t / 1e12
        from mojo_opset_amd import switches
        os.environ["MOJO_HIP_MLA_PREFILL"] = "absorbed"
        switches.reload()                               # (switches are latched at first use)
        try:
            t_abs = _time(lambda: op(q, ckv, kpe, cu_q, table, cu_total_seq_lens=cu_kv), 3, 1)
        finally:
            os.environ.pop("MOJO_HIP_MLA_PREFILL", None)
            switches.reload()
        res["absorbed_route_us"] = t_abs * 1e6
        out[name] = res
    return out


def bench_streaming(device):
    out = {}
    rows, d = 2048, 4096
    for dtype, tag in ((torch.float32, "fp32"), (torch.bfloat16, "bf16")):
        x, r = torch.randn(rows, d, device=device, dtype=dtype), torch.randn(rows, d, device=device, dtype=dtype)
        norm = hip("MojoResidualAddRMSNorm")(d, 1e-5, "pre", dtype=dtype, device=device)
        with torch.no_grad():
            norm.weight.copy_(torch.randn(d))
        es = x.element_size()
        out[f"residual_add_rmsnorm_{tag}_2048x4096"] = _hbm(_time(lambda: norm(x, r), 50, 5), 4 * rows * d * es + d * es)
        act = hip("MojoSwiGLU")()
        out[f"swiglu_{tag}_2048x4096"] = _hbm(_time(lambda: act(x, r), 50, 5), 3 * rows * d * es)
    # larger streams (past the 256 MiB MALL) for the bandwidth-bound picture
    rows = 65536
    x, r = torch.randn(rows, d, device=device, dtype=torch.bfloat16), torch.randn(rows, d, device=device, dtype=torch.bfloat16)
    norm = hip("MojoResidualAddRMSNorm")(d, 1e-5, "pre", dtype=torch.bfloat16, device=device)
    with torch.no_grad():
        norm.weight.copy_(torch.randn(d))
    out["residual_add_rmsnorm_bf16_65536x4096"] = _hbm(_time(lambda: norm(x, r), 20, 3), 4 * rows * d * 2)
    out["swiglu_bf16_65536x4096"] = _hbm(_time(lambda: hip("MojoSwiGLU")()(x, r), 20, 3), 3 * rows * d * 2)
    del x, r
    # per-token activation quantisers (DeepSeek-V3 hidden 7168)
    rows, d = 8192, 7168
    x, r = torch.randn(rows, d, device=device, dtype=torch.bfloat16), torch.randn(rows, d, device=device, dtype=torch.bfloat16)
    dq = hip("MojoDynamicQuant")()
    out["dynamic_quant_bf16_8192x7168"] = _hbm(_time(lambda: dq(x), 20, 3), rows * d * 3 + rows * 4)
    nq = hip("MojoResidualAddRMSNormQuant")(norm_size=d).to(device)
    with torch.no_grad():
        nq.weight.copy_(torch.randn(d))
    out["residual_add_rmsnorm_quant_bf16_8192x7168"] = _hbm(_time(lambda: nq(x, r), 20, 3), rows * d * 7 + d * 4 + rows * 4)
    del x, r
    # RoPE: q [1,32,8192,128] + k [1,8,8192,128] head-first bf16 (the reference's published case)
    q = torch.randn(1, 32, 8192, 128, device=device, dtype=torch.bfloat16)
    k = torch.randn(1, 8, 8192, 128, device=device, dtype=torch.bfloat16)
    cos, sin = torch.randn(8192, 128, device=device), torch.randn(8192, 128, device=device)
    rope = hip("MojoApplyRoPE")()
    out["apply_rope_bf16_q32_k8_8192x128"] = _hbm(_time(lambda: rope(q, k, cos, sin, head_first=True), 50, 5),
                                                  2 * (q.numel() + k.numel()) * 2 + 2 * cos.numel() * 4)
    # StorePagedKVCache decode step: 64 sequences x 1 token, 8 kv heads x 128, page 16
    hkv, dd, page, bsz = 8, 128, 16, 64
    kc = torch.zeros(bsz * 256 + 4, hkv, page, dd, device=device, dtype=torch.bfloat16)
    vc = torch.zeros_like(kc)
    table = torch.randperm(bsz * 256, dtype=torch.int32).view(bsz, 256).to(device)
    ks, vs = torch.randn(bsz, hkv, dd, device=device, dtype=torch.bfloat16), torch.randn(bsz, hkv, dd, device=device, dtype=torch.bfloat16)
    ctx = torch.full((bsz,), 4000, dtype=torch.int32, device=device)
    store = hip("MojoStorePagedKVCache")()
    # (decode-sized: the eager figure is the Python shim, ~10 us per call; a serving loop replays graphs)
    out["store_paged_kv_decode_64x8x128"] = {"us_eager": _time(lambda: store(ks, vs, kc, vc, table, None, ctx), 50, 5) * 1e6,
                                             **_hbm(_time_graph(lambda: store(ks, vs, kc, vc, table, None, ctx)), 4 * bsz * hkv * dd * 2)}
    # prefill store: 8192 tokens
    ks, vs = torch.randn(8192, hkv, dd, device=device, dtype=torch.bfloat16), torch.randn(8192, hkv, dd, device=device, dtype=torch.bfloat16)
    cu = torch.arange(0, 8192 + 1, 2048, dtype=torch.int32, device=device)
    ctx4 = torch.zeros(4, dtype=torch.int32, device=device)
    out["store_paged_kv_prefill_8192x8x128"] = _hbm(_time(lambda: store(ks, vs, kc, vc, table[:4], cu, ctx4), 50, 5), 4 * 8192 * hkv * dd * 2)
    del kc, vc, ks, vs
    # MLA latent-cache store (DeepSeek-V3: 512 + 64 per token), prefill of 4 x 2048 tokens
    r_, rope_, page = 512, 64, 16
    pages = 4 * (2048 // page)
    ckv_c = torch.zeros(pages + 4, 1, page, r_, device=device, dtype=torch.bfloat16)
    kpe_c = torch.zeros(pages + 4, 1, page, rope_, device=device, dtype=torch.bfloat16)
    tbl = torch.randperm(pages, dtype=torch.int32).view(4, pages // 4).to(device)
    ckv_n, kpe_n = torch.randn(8192, r_, device=device, dtype=torch.bfloat16), torch.randn(8192, rope_, device=device, dtype=torch.bfloat16)
    smla = hip("MojoStorePagedMLAKVCache")()
    out["store_paged_mla_kv_prefill_8192x576"] = {"us_eager": _time(lambda: smla(ckv_n, kpe_n, ckv_c, kpe_c, tbl, cu, ctx4), 50, 5) * 1e6,
                                                  **_hbm(_time_graph(lambda: smla(ckv_n, kpe_n, ckv_c, kpe_c, tbl, cu, ctx4)), 2 * 8192 * (r_ + rope_) * 2)}
    # RotaryEmbedding: cached cos/sin gather for 8192 packed tokens (4 x 2048), rope_dim 128 -> two fp32 [8192, 128] outputs
    rot = hip("MojoRotaryEmbedding")(10000.0, 128, init_max_length=32768, device=device)
    xq = torch.empty(8192, 4096, device=device, dtype=torch.bfloat16)
    out["rotary_embedding_cached_8192x128"] = {"us_eager": _time(lambda: rot(xq, cu_q_lens=cu, total_seq_lens=None), 50, 5) * 1e6,
                                               **_hbm(_time_graph(lambda: rot(xq, cu_q_lens=cu, total_seq_lens=None)), 2 * 2 * 8192 * 128 * 4)}
    return out


def bench_compute_comm(device, world, rank):
    """Llama-3-70B row/column-parallel projections at tp = world: M in {1024, 4096, 8192};
    GemmAllReduce / GemmReduceScatter at (K, N) = (28672, 8192) and (8192, 8192), K split over the ranks; AllGatherGemm at
    N_total in {10240, 57344}, N split; GemmAll2All at one Ulysses-style shape.  Every rank runs every case (they contain
    collectives); rank 0's times are reported.  With world > 1 each reduce case is timed twice — the collective library's
    ring under the chunked pipeline ("rccl") and the ring-free peer exchange ("direct", MOJO_HIP_COMM_DIRECT=1) — next to
    the local GEMM alone and the full-K GEMM on one GPU (speedup_vs_tp1, target >= 6 at tp = 8)."""
    from mojo_opset_amd.backends.hip.operators.compute_with_comm import _ENGINE

    out = {}
    dt = torch.bfloat16
    link_peak = 153.0                                   # GB/s per xGMI link

    def timed(fn):
        return _time(fn, 5, 2, settle_n=12)             # a fixed call count: every rank must issue the same collectives

    from mojo_opset_amd import switches
    from mojo_opset_amd.backends.hip import lib as L

    def with_direct(flag, fn):
        old = os.environ.get("MOJO_HIP_COMM_DIRECT")
        if flag is None:                                # auto: what comm/select.py decided in warm() (self-test + timing per payload)
            os.environ.pop("MOJO_HIP_COMM_DIRECT", None)
        else:
            os.environ["MOJO_HIP_COMM_DIRECT"] = flag
        switches.reload()                               # (switches are latched at first use)
        try:
            return fn()
        finally:
            if old is None:
                os.environ.pop("MOJO_HIP_COMM_DIRECT", None)
            else:
                os.environ["MOJO_HIP_COMM_DIRECT"] = old
            switches.reload()

    # The direct exchange has run on 2 ranks of one GPU only.  Its flag waits are bounded; a short bound here keeps a broken
    # fabric path from eating the extras' deadline, and once ANY rank has seen it fail every rank stops timing it (the ranks
    # agree through a max-reduce, so they keep making the same collective calls).
    if "MOJO_HIP_PEER_TIMEOUT_MS" not in os.environ:
        L.load().mojo_hip_peer_set_timeout_ms(3000)
    direct_state = {"off": False}
    if world > 1:
        # what a serving engine does at start-up: decide the exchange for the shapes it will run, OUTSIDE any step
        # (comm/select.py warm(): self-test of the direct exchange once per group, both paths timed per payload)
        from mojo_opset_amd.comm import select
        import torch.distributed as dist
        try:
            with_direct(None, lambda: select.warm(dist.group.WORLD, [
                (op_, m_, k_ // world, n_, dt) for k_, n_ in ((28672, 8192), (8192, 8192)) for m_ in (1024, 4096, 8192)
                for op_ in ("gemm_all_reduce", "gemm_reduce_scatter")], device))
        except Exception as e:
            out["warm_error"] = repr(e)

    def direct_variant(fn):
        """(seconds | None, error | None) of one direct-exchange case, identical verdict on every rank."""
        import torch.distributed as dist
        if direct_state["off"]:
            return None, "skipped: the direct exchange failed in an earlier case"
        err, t = None, None
        try:
            t = with_direct("1", fn)
            from mojo_opset_amd.comm import peer
            peer.check_all()
        except Exception as e:
            err = repr(e)
        bad = torch.tensor([1 if err else 0], dtype=torch.int32, device=device)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if int(bad.item()):
            direct_state["off"] = True
            return None, err or "failed on another rank"
        return t, None

    for k_total, n in ((28672, 8192), (8192, 8192)):
        kl = k_total // world
        w = torch.randn(kl, n, device=device, dtype=dt) * 0.02
        wf = torch.randn(k_total, n, device=device, dtype=dt) * 0.02 if world > 1 else None
        for m in (1024, 4096, 8192):
            x = torch.randn(m, kl, device=device, dtype=dt)
            t_local = timed(lambda: _ENGINE(x, w, None, True))
            t1 = None
            if world > 1:
                xf = torch.randn(m, k_total, device=device, dtype=dt)
                t1 = timed(lambda: _ENGINE(xf, wf, None, True))
                del xf
            payload = m * n * 2
            bare = {}
            import torch.distributed as dist
            if world > 1 and dist.get_backend() == "nccl":  # the collective alone on the same payload: what overlap is measured against
                buf = torch.randn(m, n, device=device, dtype=dt)
                shard = torch.empty(m // world, n, device=device, dtype=dt)
                bare["gemm_allreduce"] = timed(lambda: dist.all_reduce(buf))
                bare["gemm_reducescatter"] = timed(lambda: dist.reduce_scatter_tensor(shard, buf))
                del buf, shard
            for name, cls, kw in (("gemm_allreduce", "MojoGemmAllReduce", {}), ("gemm_reducescatter", "MojoGemmReduceScatter", {"scatter_dim": 0})):
                op = hip(cls)(w, None, True, **kw)
                phases = 2 if name == "gemm_allreduce" else 1
                for variant in (("rccl", "direct", "auto") if world > 1 else ("rccl",)):
                    key = f"{name}_M{m}_K{k_total}_N{n}_tp{world}" + ("" if world == 1 else f"_{variant}")
                    algorithm = variant
                    if variant == "direct":
                        t, err = direct_variant(lambda: timed(lambda: op(x)))
                        if err:
                            out[key] = {"error": err}
                            continue
                    elif variant == "auto":            # what a user gets with no switch set: the selector's cached choice
                        from mojo_opset_amd.comm import select
                        if direct_state["off"]:
                            select.disable_direct(op._group(), "disabled: the direct exchange failed earlier in this run")
                        t = with_direct(None, lambda: timed(lambda: op(x)))
                        picks = [r for r in select.report() if r["op"] == {"gemm_allreduce": "gemm_all_reduce", "gemm_reducescatter": "gemm_reduce_scatter"}[name]
                                 and r["payload_bucket_MB"] == select.bucket(payload) / 2 ** 20]
                        algorithm = picks[-1]["algorithm"] if picks else "rccl"
                    else:
                        t = with_direct("0", lambda: timed(lambda: op(x)))
                    rec = {"us": t * 1e6, "aggregate_tflops": 2.0 * m * k_total * n / t / 1e12, "local_gemm_us": t_local * 1e6,
                           "exposed_exchange_us": max(t - t_local, 0.0) * 1e6, "payload_MB_per_rank": payload / 1e6,
                           "algorithm": algorithm}
                    if world > 1:
                        # bytes one rank moves over ONE link: ring = phases*(ws-1)/ws of the payload over its single ring link;
                        # direct = phases * payload/ws from each of its ws-1 peers
                        per_link = phases * payload * ((world - 1) / world if algorithm == "rccl" else 1.0 / world)
                        exposed = max(t - t_local, 0.0)
                        if name in bare:
                            rec.update({"bare_collective_us": bare[name] * 1e6,
                                        "overlap_frac": max(0.0, min(1.0, 1.0 - exposed / max(bare[name], 1e-9)))})
                        rec.update({"tp1_full_gemm_us": t1 * 1e6, "speedup_vs_tp1": t1 / t,
                                    "link_MB": per_link / 1e6, "link_GBps_over_whole_op": per_link / t / 1e9,
                                    "link_GBps_over_exposed_time": per_link / max(t - t_local, 1e-6) / 1e9,
                                    "link_peak_GBps": link_peak})
                    out[key] = rec
            del x
        del w, wf
        torch.cuda.empty_cache()
    # AllGatherGemm: x [M/tp, 8192] gathered along rows, column-parallel weight [8192, N_total/tp]
    k2 = 8192
    for n_total in (10240, 57344):
        nl = n_total // world
        w2 = torch.randn(k2, nl, device=device, dtype=dt) * 0.02
        for m in (1024, 4096, 8192):
            xs = torch.randn(m // world, k2, device=device, dtype=dt)
            op = hip("MojoAllGatherGemm")(w2, None, True, gather_dim=0)
            t_local = None
            if world > 1:
                xfull = torch.randn(m, k2, device=device, dtype=dt)
                t_local = timed(lambda: _ENGINE(xfull, w2, None, True))
                del xfull
            for variant in (("rccl", "direct") if world > 1 else ("rccl",)):
                key = f"allgather_gemm_M{m}_K{k2}_N{n_total}_tp{world}" + ("" if world == 1 else f"_{variant}")
                if variant == "direct":
                    t, err = direct_variant(lambda: timed(lambda: op(xs)))
                    if err:
                        out[key] = {"error": err}
                        continue
                else:
                    t = with_direct("0", lambda: timed(lambda: op(xs)))
                rec = {"us": t * 1e6, "aggregate_tflops": 2.0 * m * k2 * n_total / t / 1e12}
                if world > 1:
                    gathered = (world - 1) * (m // world) * k2 * 2
                    per_link = gathered if variant == "rccl" else gathered / (world - 1)     # ring: all of it over one link
                    rec.update({"local_gemm_us": t_local * 1e6, "exposed_exchange_us": max(t - t_local, 0.0) * 1e6,
                                "gathered_MB_per_rank": gathered / 1e6, "link_MB": per_link / 1e6,
                                "link_GBps_over_exposed_time": per_link / max(t - t_local, 1e-6) / 1e9, "link_peak_GBps": link_peak})
                out[key] = rec
            del xs
        del w2
        torch.cuda.empty_cache()
    # GemmAll2All (Ulysses-style): x [M/tp, 8192] @ W [8192, 10240] then all-to-all rows -> columns
    n_total, m = 10240, 4096
    xs2 = torch.randn(max(m // world, world), k2, device=device, dtype=dt)
    w3 = torch.randn(k2, n_total, device=device, dtype=dt) * 0.02
    op = hip("MojoGemmAll2All")(w3, None, True, scatter_dim=0, gather_dim=1)
    t = timed(lambda: op(xs2))
    out[f"gemm_all2all_M{xs2.shape[0]}_K8192_N10240_tp{world}"] = {"us": t * 1e6, "aggregate_tflops": 2.0 * xs2.shape[0] * world * k2 * n_total / t / 1e12}
    if world > 1:
        from mojo_opset_amd.comm import select
        out["selection"] = select.report()              # what comm/select.py decided, per (operator, payload bucket), and why
    return out


def bench_moe(device):
    """Mixtral routing: T = 8192 tokens, 8 experts, top-2, hidden 4096, inter 14336."""
    out = {}
    torch.manual_seed(20260716)                                   # the routing (rows per expert) decides the tile count: keep it fixed
    t_, e_, k_, h_, i_ = 8192, 8, 2, 4096, 14336
    x = torch.rand(t_, h_, device=device, dtype=torch.bfloat16)
    gating = hip("MojoMoEGating")(hidden_size=h_, num_experts=e_, top_k=k_).to(device)
    with torch.no_grad():
        gating.gate_weight.copy_(torch.randn(h_, e_) * 0.02)
    out["gating_T8192_E8_k2_H4096"] = _hbm(_time(lambda: gating(x), 20, 3), t_ * h_ * 2 + h_ * e_ * 4 + t_ * k_ * 8)
    # DeepSeek-V3-sized router: 256 experts, top-8, hidden 7168 (logits on MFMA, hi/lo split of the fp32 gate weight)
    xd = torch.rand(t_, 7168, device=device, dtype=torch.bfloat16)
    gd = hip("MojoMoEGating")(hidden_size=7168, num_experts=256, top_k=8).to(device)
    with torch.no_grad():
        gd.gate_weight.copy_(torch.randn(7168, 256) * 0.02)
    out["gating_T8192_E256_k8_H7168"] = _mfma(_time(lambda: gd(xd), 20, 3), 2.0 * t_ * 7168 * 256 * 2)
    del xd, gd
    # the router at decode: 64 tokens (graph replay; the few-token kernel: a grid over slices of H, fp32 FMAs, slabs + select)
    for name, (td, ed, kd, hd) in {"gating_decode_T64_E64_k8_H4096": (64, 64, 8, 4096), "gating_decode_T64_E256_k8_H7168": (64, 256, 8, 7168)}.items():
        xs = torch.rand(td, hd, device=device, dtype=torch.bfloat16)
        gs = hip("MojoMoEGating")(hidden_size=hd, num_experts=ed, top_k=kd).to(device)
        with torch.no_grad():
            gs.gate_weight.copy_(torch.randn(hd, ed) * 0.02)
        out[name] = _hbm(_time_graph(lambda: gs(xs), reps=10), hd * ed * 4 + td * hd * 2)
        del xs, gs
    idx, gates = gating(x)
    dispatch = hip("MojoMoEDispatch")(num_experts=e_)
    out["dispatch_T8192_E8_k2_H4096"] = _hbm(_time(lambda: dispatch(x, gates, idx), 20, 3), t_ * h_ * 2 + t_ * k_ * (h_ * 2 + 16))
    sh, counts, sg, tok = dispatch(x, gates, idx)
    combine = hip("MojoMoECombine")()
    buf = torch.empty_like(x)
    out["combine_T8192_k2_H4096"] = _hbm(_time(lambda: combine(buf, sh, sg, tok), 20, 3), t_ * k_ * (h_ * 2 + 8) + t_ * h_ * 2)
    experts = hip("MojoExperts")(num_experts=e_, hidden_size=h_, intermediate_size=i_).to(torch.bfloat16).to(device)
    with torch.no_grad():
        experts.up_proj_weight.normal_(std=0.02)
        experts.down_proj_weight.normal_(std=0.02)
    flops = 2.0 * t_ * k_ * h_ * (2 * i_) + 2.0 * t_ * k_ * i_ * h_
    out["experts_T8192x2_E8_H4096_I14336"] = _mfma(_time(lambda: experts(sh, counts), 5, 1), flops)

    def layer():
        i2, g2 = gating(x)
        a, c, b, d = dispatch(x, g2, i2)
        return combine(buf, experts(a, c), b, d)
    out["moe_layer_T8192_E8_k2"] = _mfma(_time(layer, 5, 1), flops)
    del experts
    # MoE at decode: 64 tokens, top-8 of 64 experts (hidden 4096, inter 2048): a few rows per expert, the weights of the experts
    # that received rows are a stream (50 MB each: 3 x 4096 x 2048 x 2 B); zero-mean inputs so that the routing spreads
    td, ed, kd, hd, idm = 64, 64, 8, 4096, 2048
    torch.manual_seed(20260717)
    xd = torch.randn(td, hd, device=device, dtype=torch.bfloat16)
    gd = hip("MojoMoEGating")(hidden_size=hd, num_experts=ed, top_k=kd).to(device)
    exd = hip("MojoExperts")(num_experts=ed, hidden_size=hd, intermediate_size=idm).to(torch.bfloat16).to(device)
    with torch.no_grad():
        gd.gate_weight.copy_(torch.randn(hd, ed) * 0.02)
        exd.up_proj_weight.normal_(std=0.02)
        exd.down_proj_weight.normal_(std=0.02)
    dd, cd = hip("MojoMoEDispatch")(num_experts=ed), hip("MojoMoECombine")()
    bufd = torch.empty_like(xd)
    i0, g0 = gd(xd)
    rows0, counts0, _, _ = dd(xd, g0, i0)
    used = int((counts0 > 0).sum())                        # (read once, outside the timed region)

    def decode_layer():
        i2, g2 = gd(xd)
        a, c, b, d = dd(xd, g2, i2)
        return cd(bufd, exd(a, c), b, d)
    w_bytes = used * 3 * hd * idm * 2
    res = _hbm(_time_graph(decode_layer, reps=5, replays=3), w_bytes)
    res["experts_with_rows"] = used
    res["experts_only"] = _hbm(_time_graph(lambda: exd(rows0, counts0), reps=5, replays=3), w_bytes)
    out["moe_layer_decode_T64_E64_k8_H4096_I2048"] = res
    del exd
    return out


def bench_decode_layer(device, bsz=64):
    """One whole Llama-3-8B decoder layer at decode (B = 64, ctx = 4096, bf16) as ONE captured graph of the operators of this
    package: residual-add RMSNorm -> QKV projection -> RoPE -> paged KV store -> paged decode attention -> output projection ->
    residual-add RMSNorm -> gate|up projection -> SwiGLU -> down projection.  Reported next to the sum of its memory traffic:
    what the per-operator numbers add up to once launch gaps and host work are out of the way."""
    from mojo_opset_amd.backends.hip.operators.gemm import dense_gemm as _ENGINE
    hq, hkv, d, page, ctx, hidden, inter = 32, 8, 128, 16, 4096, 4096, 14336
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(20260716)
    lens = [ctx] * bsz
    k_cache, v_cache, table = _paged(device, [ctx + page] * bsz, hkv, d, page)
    w_qkv = torch.randn((hq + 2 * hkv) * d, hidden, device=device, dtype=dt) * 0.02          # [N, K]
    w_o = torch.randn(hidden, hq * d, device=device, dtype=dt) * 0.02
    w_gu = torch.randn(2 * inter, hidden, device=device, dtype=dt) * 0.02
    w_dn = torch.randn(hidden, inter, device=device, dtype=dt) * 0.02
    x = torch.randn(bsz, hidden, device=device, dtype=dt)
    resid = torch.randn(bsz, hidden, device=device, dtype=dt)
    cos, sin = torch.randn(bsz, d, device=device), torch.randn(bsz, d, device=device)
    ctx_t = torch.tensor(lens, dtype=torch.int32, device=device)
    total_t = ctx_t + 1
    norm1 = hip("MojoResidualAddRMSNorm")(hidden, 1e-5, "pre", dtype=dt, device=device)
    norm2 = hip("MojoResidualAddRMSNorm")(hidden, 1e-5, "pre", dtype=dt, device=device)
    rope, store, attn, act = hip("MojoApplyRoPE")(), hip("MojoStorePagedKVCache")(), hip("MojoPagedDecodeGQA")(is_causal=True, gqa_layout="AABB"), hip("MojoSwiGLU")()

    def layer():
        h, r1 = norm1(x, resid)
        qkv = _ENGINE(h, w_qkv, None, False)
        q = qkv[:, : hq * d].reshape(bsz, hq, d)
        k = qkv[:, hq * d: (hq + hkv) * d].reshape(bsz, hkv, d)
        v = qkv[:, (hq + hkv) * d:].reshape(bsz, hkv, d).contiguous()
        q_r, k_r = rope(q.unsqueeze(0), k.unsqueeze(0), cos, sin, head_first=False)
        store(k_r.squeeze(0).contiguous(), v, k_cache, v_cache, table, None, ctx_t)
        o = attn(q_r.squeeze(0).contiguous(), k_cache, v_cache, total_t, table, max_total_seq_len=ctx + 1)
        a = _ENGINE(o.reshape(bsz, hq * d), w_o, None, False)
        h2, r2 = norm2(a, r1)
        gu = _ENGINE(h2, w_gu, None, False)
        m = act(gu[:, :inter], gu[:, inter:])               # (the halves of the fused projection are read in place)
        return _ENGINE(m, w_dn, None, False), r2

    t = _time_graph(layer, reps=4, replays=5)

    # The same layer period with the decode-sized fusions (round 4): QKV projection -> RoPE -> paged KV store with the K-slice
    # sums taken by the kernel that rotates and stores, output projection -> residual RMSNorm with the K-slice
    # sums feeding the norm kernel, gate|up projection -> SwiGLU in ONE launch, down projection -> the NEXT layer's residual
    # RMSNorm likewise.  Same operators' work (two norms, four projections, RoPE, store, attention); the period starts behind
    # the first norm and ends behind the next layer's, as it would inside a stack.
    from mojo_opset_amd.backends.hip.operators.gemm import dense_gemm_residual_rmsnorm as _GEMM_NORM, dense_gemm_swiglu as _GEMM_GLU
    h0, r0 = norm1(x, resid)

    from mojo_opset_amd.backends.hip.operators.gemm import qkv_rope_store as _QKV

    def layer_fused():
        q_r = _QKV(h0, w_qkv, None, cos, sin, k_cache, v_cache, table, ctx_t, hq, hkv)
        o = attn(q_r, k_cache, v_cache, total_t, table, max_total_seq_len=ctx + 1)
        h2, r2 = _GEMM_NORM(o.reshape(bsz, hq * d), w_o, None, r0, norm2.weight, 1e-5)
        m = _GEMM_GLU(h2, w_gu)
        return _GEMM_NORM(m, w_dn, None, r2, norm1.weight, 1e-5)

    t_fused = _time_graph(layer_fused, reps=4, replays=5)

    # The same layer written ONLY against the reference's operator API (round 5): `MojoGemm` for the projections and
    # `MojoSwiGLUMLP` for the gated MLP — what a model file of the reference gets with MOJO_BACKEND=hip and no code change.  The
    # MLP's gate|up projection + SwiGLU is one launch (mojo_hip_gemm_swiglu) behind that class; the other fusions need the chain
    # helpers above.
    g_qkv, g_o = hip("MojoGemm")(weight=w_qkv), hip("MojoGemm")(weight=w_o)
    mlp = hip("MojoSwiGLUMLP")(hidden, hidden, inter).to(dt).to(device)
    with torch.no_grad():
        mlp.fc1.weight.copy_(w_gu)
        mlp.fc2.weight.copy_(w_dn)

    def layer_api():
        h, r1 = norm1(x, resid)
        qkv = g_qkv(h)
        q = qkv[:, : hq * d].reshape(bsz, hq, d)
        k = qkv[:, hq * d: (hq + hkv) * d].reshape(bsz, hkv, d)
        v = qkv[:, (hq + hkv) * d:].reshape(bsz, hkv, d).contiguous()
        q_r, k_r = rope(q.unsqueeze(0), k.unsqueeze(0), cos, sin, head_first=False)
        store(k_r.squeeze(0).contiguous(), v, k_cache, v_cache, table, None, ctx_t)
        o = attn(q_r.squeeze(0).contiguous(), k_cache, v_cache, total_t, table, max_total_seq_len=ctx + 1)
        h2, r2 = norm2(g_o(o.reshape(bsz, hq * d)), r1)
        return mlp(h2), r2

    t_api = _time_graph(layer_api, reps=4, replays=5)
    # the same operators one at a time (each under graph replay on its own): their sum against the whole layer shows what
    # chaining costs (cold weights: each GEMM's 0.1 - 0.2 GB of weights is evicted from the 256 MB MALL by the next one)
    h, r1 = norm1(x, resid)
    qkv = _ENGINE(h, w_qkv, None, False)
    q = qkv[:, : hq * d].reshape(bsz, hq, d).contiguous()
    k = qkv[:, hq * d: (hq + hkv) * d].reshape(bsz, hkv, d).contiguous()
    v = qkv[:, (hq + hkv) * d:].reshape(bsz, hkv, d).contiguous()
    o = attn(q, k_cache, v_cache, total_t, table, max_total_seq_len=ctx + 1).reshape(bsz, hq * d)
    gu = _ENGINE(h, w_gu, None, False)
    gate, up = gu[:, :inter].contiguous(), gu[:, inter:].contiguous()
    m = act(gate, up)
    parts = {
        "norm_x2": 2 * _time_graph(lambda: norm1(x, resid)),
        "qkv_gemm": _time_graph(lambda: _ENGINE(h, w_qkv, None, False)),
        "rope": _time_graph(lambda: rope(q.unsqueeze(0), k.unsqueeze(0), cos, sin, head_first=False)),
        "kv_store": _time_graph(lambda: store(k, v, k_cache, v_cache, table, None, ctx_t)),
        "attention": _time_graph(lambda: attn(q, k_cache, v_cache, total_t, table, max_total_seq_len=ctx + 1), reps=4),
        "o_gemm": _time_graph(lambda: _ENGINE(o, w_o, None, False)),
        "gate_up_gemm": _time_graph(lambda: _ENGINE(h, w_gu, None, False), reps=4),
        "swiglu": _time_graph(lambda: act(gate, up)),
        "down_gemm": _time_graph(lambda: _ENGINE(m, w_dn, None, False), reps=4),
    }
    weights = (w_qkv.numel() + w_o.numel() + w_gu.numel() + w_dn.numel()) * 2
    kv = sum(lens) * hkv * d * 2 * 2
    res = _hbm(t, weights + kv)
    fused = _hbm(t_fused, weights + kv)
    fused.update({"tokens_per_s_one_layer": bsz / t_fused,
                  "per_op_us": {"qkv_gemm+rope+store": _time_graph(lambda: _QKV(h, w_qkv, None, cos, sin, k_cache, v_cache, table, ctx_t, hq, hkv)) * 1e6,
                                "o_gemm+norm": _time_graph(lambda: _GEMM_NORM(o, w_o, None, r1, norm2.weight, 1e-5)) * 1e6,
                                "gate_up_gemm+swiglu": _time_graph(lambda: _GEMM_GLU(h, w_gu), reps=4) * 1e6,
                                "down_gemm+norm": _time_graph(lambda: _GEMM_NORM(m, w_dn, None, r1, norm1.weight, 1e-5), reps=4) * 1e6},
                  "note": "graph replay; one period of the layer stack with mojo_hip_qkv_rope_store, mojo_hip_gemm_residual_rmsnorm (o-proj and "
                          "down-proj, the latter feeding the next layer's norm) and mojo_hip_gemm_swiglu: 6 launches + the attention's; "
                          "bit-identical to the separate calls"})
    res.update({"weights_MB": weights / 1e6, "kv_MB": kv / 1e6, "tokens_per_s_one_layer": bsz / t,
                "per_op_us": {n: v * 1e6 for n, v in parts.items()}, "sum_of_ops_us": sum(parts.values()) * 1e6,
                "note": "graph replay; bytes = the layer's weights + the K/V the attention reads (activations are noise at B = 64)"})
    api = _hbm(t_api, weights + kv)
    api.update({"tokens_per_s_one_layer": bsz / t_api,
                "note": "graph replay; the layer written only against the reference's operator API: MojoResidualAddRMSNorm, MojoGemm (QKV, o), "
                        "MojoApplyRoPE, MojoStorePagedKVCache, MojoPagedDecodeGQA, MojoSwiGLUMLP (gate|up + SwiGLU in one launch, then down)"})
    tag = f"llama3_8b_layer_B{bsz}_ctx4096"
    return {tag: res, tag + "_fused": fused, tag + "_mojo_api": api}


def bench_dense_mid_m(device):
    """Dense products at mid-size M (a prefill chunk of 256-2048 tokens; config 4 at M 1024): this backend's `mojo_hip_gemm`
    (kernel form recorded: the 128-row tiles of round 5, or the 256 x 256 kernel) next to hipBLASLt on the same box in the same
    run (`F.linear` / `x @ w`), and MojoQuantGemm ([N,K] weights) at the same sizes.  Device times under graph replay."""
    from mojo_opset_amd.backends.hip import lib as _L
    from mojo_opset_amd.backends.hip.operators.gemm import dense_gemm
    out = {}
    dt = torch.bfloat16
    for m, k, n in ((512, 4096, 4096), (1024, 4096, 4096), (2048, 4096, 4096), (1024, 4096, 6144), (512, 3584, 8192), (1024, 14336, 4096)):
        x = torch.randn(m, k, device=device, dtype=dt)
        for layout in ("NK", "KN"):
            if not _want(f"dense_{m}x{k}x{n}_{layout}"):
                continue
            w = torch.randn(n, k, device=device, dtype=dt) * 0.02
            w = w if layout == "NK" else w.t().contiguous()
            trans = layout == "KN"
            t = _time_graph(lambda: dense_gemm(x, w, None, trans), reps=10)
            form = _L.last_launch()
            t_lib = _time_graph((lambda: x @ w) if trans else (lambda: torch.nn.functional.linear(x, w)), reps=10)
            rec = _mfma(t, 2.0 * m * k * n)
            rec.update({"kernel_form": form, "hipblaslt_same_box_us": t_lib * 1e6, "time_vs_hipblaslt": t / t_lib})
            out[f"dense_{m}x{k}x{n}_{layout}"] = rec
    # the fall-offs the round-5 sweep against the vendor library found (scripts/probes/gemm_sweep_vs_lib.py): decode-sized rows with
    # [K,N] weights (`x @ w`: the GEMM + collective operators' trans_weight), 65-128 rows (a 128-row lm_head), a bias on a
    # chip-filling product
    for m, k, n, layout, bias in ((32, 4096, 4096, "KN", False), (64, 8192, 1024, "KN", False), (128, 4096, 128256, "NK", False),
                                  (96, 4096, 14336, "NK", False), (8192, 4096, 6144, "NK", True)):
        name = f"dense_{m}x{k}x{n}_{layout}" + ("_bias" if bias else "")
        if not _want(name):
            continue
        x = torch.randn(m, k, device=device, dtype=dt)
        w = torch.randn(n, k, device=device, dtype=dt) * 0.02
        w = w if layout == "NK" else w.t().contiguous()
        b = torch.randn(n, device=device, dtype=dt) if bias else None
        trans = layout == "KN"
        reps = 10 if m * k * n < 2 ** 36 else 3
        t = _time_graph(lambda: dense_gemm(x, w, b, trans), reps=reps)
        form = _L.last_launch()
        if trans:
            t_lib = _time_graph((lambda: x @ w) if b is None else (lambda: torch.addmm(b, x, w)), reps=reps)
        else:
            t_lib = _time_graph(lambda: torch.nn.functional.linear(x, w, b), reps=reps)
        rec = _mfma(t, 2.0 * m * k * n)
        rec.update({"kernel_form": form, "hipblaslt_same_box_us": t_lib * 1e6, "time_vs_hipblaslt": t / t_lib})
        out[name] = rec
        del x, w
    # COLD weights: every call of the graph on another copy of the weight (copies x bytes >= 768 MB, three times the last-level
    # cache) — what a model's layers see; the one-weight rows above re-read the weight from the cache.  gemm_api.hip's kernel
    # choice is fitted on this regime (scripts/probes/gemm_cold_weights_ab.py).
    for m, k, n in ((256, 8192, 1024), (256, 4096, 4096), (512, 4096, 4096), (1024, 4096, 4096), (1024, 14336, 4096), (128, 14336, 4096)):
        name = f"dense_{m}x{k}x{n}_NK_cold_weights"
        if not _want(name):
            continue
        copies = max(2, -(-768 * 2 ** 20 // (k * n * 2)))
        ws = [torch.randn(n, k, device=device, dtype=dt) * 0.02 for _ in range(copies)]
        x = torch.randn(m, k, device=device, dtype=dt)
        turn = [0, 0]

        def ours():
            turn[0] += 1
            return dense_gemm(x, ws[turn[0] % copies], None, False)

        def lib():
            turn[1] += 1
            return torch.nn.functional.linear(x, ws[turn[1] % copies])
        t = _time_graph(ours, reps=max(10, copies))
        form = _L.last_launch()
        t_lib = _time_graph(lib, reps=max(10, copies))
        rec = _mfma(t, 2.0 * m * k * n)
        rec.update({"kernel_form": form, "weight_copies": copies, "hipblaslt_same_box_us": t_lib * 1e6, "time_vs_hipblaslt": t / t_lib})
        out[name] = rec
        del ws, x
    # few small experts (8 x 2048 -> 1408, 128 rows each): the grouped product on the 128-row tiles
    if _want("group_8x128_2048x1408_KN"):
        gw = torch.randn(8, 2048, 1408, device=device, dtype=dt) * 0.02
        gx = torch.randn(8 * 128, 2048, device=device, dtype=dt)
        gl = torch.full((8,), 128, dtype=torch.int32, device=device)
        gop = hip("MojoGroupGemm")(gw, False)
        t = _time_graph(lambda: gop(gx, gl), reps=10)
        rec = _mfma(t, 2.0 * 1024 * 2048 * 1408)
        rec["kernel_form"] = _L.last_launch()
        out["group_8x128_2048x1408_KN"] = rec
    for qd, tag in ((torch.int8, "int8"), (torch.float8_e4m3fn, "fp8_e4m3")):
        for m, k, n in ((512, 4096, 4096), (1024, 4096, 4096), (1024, 7168, 4096), (2048, 7168, 1536)):
            if not _want(f"quant_{tag}_{m}x{k}x{n}_NK"):
                continue
            op = hip("MojoQuantGemm")(k, n, output_dtype=dt, trans_weight=True, quant_dtype=qd, weight_dtype=qd, device=device)
            if qd == torch.int8:
                op.weight.copy_(torch.randint(-127, 128, (n, k), dtype=torch.int8, device=device))
                xq = torch.randint(-127, 128, (m, k), dtype=torch.int8, device=device)
            else:
                op.weight.copy_(torch.randn(n, k, device=device).to(qd))
                xq = torch.randn(m, k, device=device).to(qd)
            op.weight_scale.fill_(0.01)
            sc = torch.rand(m, device=device)
            t = _time_graph(lambda: op(xq, sc), reps=10)
            rec = _mfma(t, 2.0 * m * k * n, peak=2 * MFMA_BF16_PEAK_TFLOPS)
            rec["kernel_form"] = _L.last_launch()
            out[f"quant_{tag}_{m}x{k}x{n}_NK"] = rec
            del op
    return out


def bench_prefill_layer(device):
    """One whole Llama-3-8B decoder layer at PREFILL, written only against the reference's operator API (what a model file gets
    with MOJO_BACKEND=hip): MojoResidualAddRMSNorm -> MojoGemm (QKV) -> MojoApplyRoPE -> MojoStorePagedKVCache ->
    MojoPagedPrefillGQA -> MojoGemm (o) -> MojoResidualAddRMSNorm -> MojoSwiGLUMLP.  Two batches: 4 x 2048 new tokens (the
    projections are chip-filling 256 x 256-tile GEMMs, gate|up with the fused SwiGLU epilogue) and a CHUNK of 1024 tokens behind
    4096 cached ones (mid-size M: the 128-row-tile kernel of round 5).  One captured graph each; MFMA fraction of the whole layer
    against the dense bf16 peak."""
    hq, hkv, d, page, hidden, inter = 32, 8, 128, 16, 4096, 14336
    dt = torch.bfloat16
    w_qkv = torch.randn((hq + 2 * hkv) * d, hidden, device=device, dtype=dt) * 0.02
    w_o = torch.randn(hidden, hq * d, device=device, dtype=dt) * 0.02
    norm1 = hip("MojoResidualAddRMSNorm")(hidden, 1e-5, "pre", dtype=dt, device=device)
    norm2 = hip("MojoResidualAddRMSNorm")(hidden, 1e-5, "pre", dtype=dt, device=device)
    rope, store, attn = hip("MojoApplyRoPE")(), hip("MojoStorePagedKVCache")(), hip("MojoPagedPrefillGQA")()
    g_qkv, g_o = hip("MojoGemm")(weight=w_qkv), hip("MojoGemm")(weight=w_o)
    mlp = hip("MojoSwiGLUMLP")(hidden, hidden, inter).to(dt).to(device)
    with torch.no_grad():
        mlp.fc1.weight.normal_(std=0.02)
        mlp.fc2.weight.normal_(std=0.02)
    out = {}
    for name, (q_lens, cached) in {"llama3_8b_prefill_layer_4x2048": ([2048] * 4, [0] * 4),
                                   "llama3_8b_prefill_layer_chunk1024_cached4096": ([1024], [4096])}.items():
        if not _want(name):
            continue
        tokens = sum(q_lens)
        kv = [a + b for a, b in zip(q_lens, cached)]
        k_cache, v_cache, table = _paged(device, kv, hkv, d, page)
        cu = lambda l: torch.tensor([0] + list(torch.tensor(l).cumsum(0).tolist()), dtype=torch.int32, device=device)  # noqa: E731
        cu_q, cu_kv = cu(q_lens), cu(kv)
        ctx_t = torch.tensor(cached, dtype=torch.int32, device=device)
        x = torch.randn(tokens, hidden, device=device, dtype=dt)
        resid = torch.randn(tokens, hidden, device=device, dtype=dt)
        cos, sin = torch.randn(tokens, d, device=device), torch.randn(tokens, d, device=device)

        def layer():
            h, r1 = norm1(x, resid)
            qkv = g_qkv(h)
            q = qkv[:, : hq * d].reshape(tokens, hq, d)
            k = qkv[:, hq * d: (hq + hkv) * d].reshape(tokens, hkv, d)
            v = qkv[:, (hq + hkv) * d:].reshape(tokens, hkv, d).contiguous()
            q_r, k_r = rope(q.unsqueeze(0), k.unsqueeze(0), cos, sin, head_first=False)
            store(k_r.squeeze(0).contiguous(), v, k_cache, v_cache, table, cu_q, ctx_t)
            o = attn(q_r.squeeze(0).contiguous(), k_cache, v_cache, cu_q, table, cu_total_seq_lens=cu_kv, max_q_len=max(q_lens),
                     max_total_seq_len=max(kv))
            h2, r2 = norm2(g_o(o.reshape(tokens, hq * d)), r1)
            return mlp(h2), r2

        from mojo_opset_amd.backends.hip import lib as _L
        _L.launch_history(clear=True)
        layer()
        forms = _L.launch_history()
        t = _time_graph(layer, reps=2, replays=5)
        h, r1 = norm1(x, resid)
        qkv = g_qkv(h)
        q = qkv[:, : hq * d].reshape(tokens, hq, d).contiguous()
        k = qkv[:, hq * d: (hq + hkv) * d].reshape(tokens, hkv, d).contiguous()
        v = qkv[:, (hq + hkv) * d:].reshape(tokens, hkv, d).contiguous()
        o = attn(q, k_cache, v_cache, cu_q, table, cu_total_seq_lens=cu_kv, max_q_len=max(q_lens), max_total_seq_len=max(kv)).reshape(tokens, hq * d)
        parts = {
            "norm_x2": 2 * _time_graph(lambda: norm1(x, resid), reps=4),
            "qkv_gemm": _time_graph(lambda: g_qkv(h), reps=4),
            "rope": _time_graph(lambda: rope(q.unsqueeze(0), k.unsqueeze(0), cos, sin, head_first=False), reps=4),
            "kv_store": _time_graph(lambda: store(k, v, k_cache, v_cache, table, cu_q, ctx_t), reps=4),
            "attention": _time_graph(lambda: attn(q, k_cache, v_cache, cu_q, table, cu_total_seq_lens=cu_kv, max_q_len=max(q_lens), max_total_seq_len=max(kv)), reps=2),
            "o_gemm": _time_graph(lambda: g_o(o), reps=4),
            "swiglu_mlp": _time_graph(lambda: mlp(h), reps=2),
        }
        gemm_flops = 2.0 * tokens * hidden * ((hq + 2 * hkv) * d + hq * d + 3 * inter)
        attn_flops = sum(4.0 * hq * d * (a * b - a * a / 2.0) for a, b in zip(q_lens, kv))
        rec = _mfma(t, gemm_flops + attn_flops)
        rec.update({"tokens": tokens, "tokens_per_s_one_layer": tokens / t, "gemm_tflop": gemm_flops / 1e12, "attention_tflop": attn_flops / 1e12,
                    "per_op_us": {n: v_ * 1e6 for n, v_ in parts.items()}, "sum_of_ops_us": sum(parts.values()) * 1e6,
                    "kernel_forms": forms,
                    "note": "graph replay; the layer written only against the reference's operator API (MojoGemm, MojoSwiGLUMLP, ...)"})
        out[name] = rec
        del k_cache, v_cache
    return out


def bench_dense_decode(device):
    """Decode-sized dense bf16 GEMMs with K-major ([N,K], `F.linear`) weights — the GEMM half of the GEMM+collective ops at
    decode batch sizes; timed under HIP-graph replay (the Python shim's launch overhead would hide the kernel)."""
    from mojo_opset_amd.backends.hip.operators.gemm import dense_gemm

    out = {}
    for m, k, n in ((1, 8192, 8192), (64, 8192, 8192), (64, 14336, 4096), (128, 8192, 8192)):
        x = torch.randn(m, k, device=device, dtype=torch.bfloat16)
        w = torch.randn(n, k, device=device, dtype=torch.bfloat16)
        t = _time_graph(lambda: dense_gemm(x, w, None, False))
        out[f"bf16_{m}x{k}x{n}_NK"] = {**_hbm(t, n * k * 2 + m * k * 2 + m * n * 2), "tflops": 2.0 * m * k * n / t / 1e12}
        del x, w
    return out


def _host_cost(fn, device_s, calls=1000, batch=200):
    """Host-side cost of an eager call: `eager_host_us` = time the Python shim + ctypes call + HIP launch take to ENQUEUE one
    call (batches of `batch` calls timed on the host clock with a sync BETWEEN batches, so the launch queue never fills and
    the device never throttles the host), `eager_wall_us` = wall per call of the same loop including the device work,
    `device_us` = the graph-replay figure.  An op whose `eager_host_us` exceeds its `device_us` needs graph capture to reach
    its device number."""
    import time

    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    host, wall, done = 0.0, 0.0, 0
    while done < calls:
        t0 = time.perf_counter()
        for _ in range(batch):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        host += t1 - t0
        wall += t2 - t0
        done += batch
    return {"eager_host_us": host / done * 1e6, "eager_wall_us": wall / done * 1e6, "device_us": device_s * 1e6,
            "needs_graph_capture": host / done > device_s, "calls": done}


def bench_host_overhead(device):
    """what the drop-in costs on the HOST.  The shims are Python + ctypes; for the operators of a decode step
    (Llama-3-8B, B 64, ctx 4096) the enqueue cost per eager call next to the device time under graph replay."""
    hq, hkv, d, page, bsz, ctx, hidden, inter = 32, 8, 128, 16, 64, 4096, 4096, 14336
    dt = torch.bfloat16
    k_cache, v_cache, table = _paged(device, [ctx + page] * bsz, hkv, d, page)
    x = torch.randn(bsz, hidden, device=device, dtype=dt)
    resid = torch.randn(bsz, hidden, device=device, dtype=dt)
    cos, sin = torch.randn(bsz, d, device=device), torch.randn(bsz, d, device=device)
    ctx_t = torch.full((bsz,), ctx, dtype=torch.int32, device=device)
    total_t = ctx_t + 1
    q = torch.randn(bsz, hq, d, device=device, dtype=dt)
    k = torch.randn(bsz, hkv, d, device=device, dtype=dt)
    v = torch.randn(bsz, hkv, d, device=device, dtype=dt)
    gate, up = torch.randn(bsz, inter, device=device, dtype=dt), torch.randn(bsz, inter, device=device, dtype=dt)
    norm = hip("MojoResidualAddRMSNorm")(hidden, 1e-5, "pre", dtype=dt, device=device)
    rope, store, attn, act = hip("MojoApplyRoPE")(), hip("MojoStorePagedKVCache")(), hip("MojoPagedDecodeGQA")(is_causal=True, gqa_layout="AABB"), hip("MojoSwiGLU")()
    gemm = hip("MojoGemm")(weight=torch.randn(hidden, hq * d, device=device, dtype=dt) * 0.02)
    mlp = hip("MojoSwiGLUMLP")(hidden, hidden, inter).to(dt).to(device)
    o = torch.randn(bsz, hq * d, device=device, dtype=dt)
    ops = {
        "MojoResidualAddRMSNorm_64x4096": lambda: norm(x, resid),
        "MojoApplyRoPE_64x(32+8)x128": lambda: rope(q.unsqueeze(0), k.unsqueeze(0), cos, sin, head_first=False),
        "MojoStorePagedKVCache_decode_64": lambda: store(k, v, k_cache, v_cache, table, None, ctx_t),
        "MojoPagedDecodeGQA_B64_ctx4096": lambda: attn(q, k_cache, v_cache, total_t, table, max_total_seq_len=ctx + 1),
        "MojoSwiGLU_64x14336": lambda: act(gate, up),
        "MojoGemm_64x4096x4096": lambda: gemm(o),
        "MojoSwiGLUMLP_64x4096x14336": lambda: mlp(x),
    }
    out = {}
    for name, fn in ops.items():
        if not _want(name):
            continue
        dev_s = _time_graph(fn, reps=4 if "Decode" in name or "MLP" in name else 20)
        out[name] = _host_cost(fn, dev_s)
    return out


def run_extras(device, world, rank=0):
    out = {}
    # With more than one rank the single-GPU cases would only repeat the N = 1 record on every GPU and delay the one thing a
    # multi-GPU run adds — the GEMM + collective cases — so they are skipped (the second headline stays: the line's
    # `roofline_group_gemm` is built from it).  MOJO_BENCH_FULL_EXTRAS=1 runs everything on every rank.
    multi_short = world > 1 and os.environ.get("MOJO_BENCH_FULL_EXTRAS", "0") != "1"
    if multi_short:
        out["note"] = "N > 1: single-GPU cases are on the N = 1 record (MOJO_BENCH_FULL_EXTRAS=1 repeats them on every rank)"
    for name, fn in (("MojoPagedDecodeGQA_bf16_other_contexts", bench_decode_variants),
                     ("MojoPagedDecodeGQA_bf16_other_geometries", bench_decode_geometries), ("MojoGroupGemm_bf16", bench_group_gemm), ("MojoQuantGemm", bench_quant_gemm),
                     ("MojoPagedPrefillGQA_bf16", bench_prefill), ("MojoPagedDecodeMLA_bf16", bench_mla_decode),
                     ("MojoPagedPrefillMLA_bf16", bench_mla_prefill),
                     ("streaming_ops", bench_streaming), ("MoE_bf16", bench_moe),
                     ("dense_gemm_decode_bf16", bench_dense_decode), ("dense_gemm_mid_m", bench_dense_mid_m),
                     ("decode_layer_bf16", bench_decode_layer), ("prefill_layer_bf16", bench_prefill_layer),
                     ("host_overhead_decode_step", bench_host_overhead)):
        if multi_short and name != "MojoGroupGemm_bf16":
            continue
        try:
            out[name] = fn(device)
        except Exception as e:  # one failing extra must not hide the others
            out[name] = {"error": repr(e)}
        torch.cuda.empty_cache()
    try:
        if world > 1 and os.environ.get("MOJO_BENCH_COMM_INPROC", "0") != "1":
            out["compute_comm_bf16"] = _comm_in_children(device, world, rank)
        else:
            out["compute_comm_bf16"] = bench_compute_comm(device, world, rank)
    except Exception as e:
        out["compute_comm_bf16"] = {"error": repr(e)}
    return out


def _comm_in_children(device, world, rank, deadline_s=None):
    """Run `bench_compute_comm` in one CHILD process per rank (benchmarks/comm_child.py) with a process group of their own:
    a crash or a hang of the never-on-real-xGMI direct exchange then costs this block only — the parent rank stays alive to
    print the result line.  Rank 0 returns the children's result (or an error record); the others return a stub."""
    import json
    import socket
    import subprocess
    import sys
    import tempfile

    import torch.distributed as dist

    deadline_s = float(os.environ.get("MOJO_BENCH_COMM_DEADLINE_S", "420")) if deadline_s is None else deadline_s
    box = [None, None]
    if rank == 0:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        box = [s.getsockname()[1], os.path.join(tempfile.gettempdir(), f"mojo_bench_comm_{os.getpid()}.json")]
        s.close()
    dist.broadcast_object_list(box, src=0)
    port, path = box
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS", "GROUP_RANK", "ROLE_RANK",
              "TORCHELASTIC_USE_AGENT_STORE"):
        env.pop(k, None)                                  # the child is a plain env:// rank, not an elastic worker
    here = os.path.dirname(os.path.abspath(__file__))
    proc = subprocess.Popen([sys.executable, os.path.join(here, "comm_child.py"), path], env=env,
                            stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    try:
        _, err = proc.communicate(timeout=deadline_s)
        status = proc.returncode
    except subprocess.TimeoutExpired:
        proc.kill()
        _, err = proc.communicate()
        status = "deadline"
    statuses = [None] * world
    dist.all_gather_object(statuses, (status, (err or "")[-600:] if status != 0 else ""))
    if rank != 0:
        return {"note": "measured in child processes; rank 0 holds the result"}
    res = None
    if os.path.exists(path):
        try:
            res = json.load(open(path))
        finally:
            os.remove(path)
    bad = {r: st for r, st in enumerate(statuses) if st[0] != 0}
    if res is None:
        return {"error": f"the GEMM + collective children left no result (per rank: {bad})"}
    if bad:
        res["child_failures"] = bad
    res["isolation"] = f"measured in {world} child processes (one per rank, own process group), deadline {deadline_s:.0f} s"
    return res
