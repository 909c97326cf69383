"""A/B in one process: MojoPagedPrefillMLA (decompressed route) with the decompression and the attention back to back
(MOJO_HIP_MLA_PREFILL_GROUPS=1) against head groups whose decompression runs on a side stream beside the previous group's
attention (2, 4); DeepSeek-V3 dims, the two bench cases; eager and under HIP-graph replay; the group count must not change a bit."""
import json, os, sys, torch
sys.path.insert(0, ".")
from benchmarks.extras import hip, _time, _time_graph
dev = torch.device("cuda:0")
h, nope, rope, vd, r, page = 128, 128, 64, 128, 512, 16
op = hip("MojoPagedPrefillMLA")(h, nope, rope, vd, r).to(torch.bfloat16).to(dev)
with torch.no_grad():
    op.kv_b_proj.copy_(torch.randn_like(op.kv_b_proj) * 0.02)
rec = {}
for name, (q_lens, cached) in {"4x512_nocache": ([512] * 4, [0] * 4), "4x512_cached2048": ([512] * 4, [2048] * 4)}.items():
    kv = [a + b for a, b in zip(q_lens, cached)]
    need = [(n + page - 1) // page for n in kv]
    total = sum(need) + 4
    ckv = torch.randn(total, 1, page, r, device=dev, dtype=torch.bfloat16)
    kpe = torch.randn(total, 1, page, rope, device=dev, dtype=torch.bfloat16)
    table = torch.randperm(total, dtype=torch.int32)[: sum(need)].view(len(kv), need[0]).to(dev)
    cu = lambda l: torch.tensor([0] + torch.tensor(l).cumsum(0).tolist(), dtype=torch.int32, device=dev)  # noqa: E731
    cu_q, cu_kv = cu(q_lens), cu(kv)
    q = torch.randn(sum(q_lens), h, nope + rope, device=dev, dtype=torch.bfloat16)
    call = lambda: op(q, ckv, kpe, cu_q, table, cu_total_seq_lens=cu_kv, max_total_seq_len=max(kv))  # noqa: E731
    outs, rec[name] = {}, {}
    for rnd in range(2):
        for g in ("1", "2", "4"):
            os.environ["MOJO_HIP_MLA_PREFILL_GROUPS"] = g
            outs[g] = call().clone()
            t = _time(call, 5, 2)
            tg = _time_graph(call, reps=4)
            rec[name].setdefault("groups_" + g, []).append({"eager_us": round(t * 1e6, 1), "graph_us": round(tg * 1e6, 1)})
    rec[name]["bit_identical_across_group_counts"] = bool(torch.equal(outs["1"], outs["2"]) and torch.equal(outs["1"], outs["4"]))
    print(name, json.dumps(rec[name]), flush=True)
json.dump(rec, open("gpurun_out/mla_prefill_groups_ab.json", "w"), indent=1)
