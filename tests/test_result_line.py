"""The driver parses the LAST stdout line of bench.py; round 3's ~24 KB line was not parsed.  These tests
feed a canned full record — round 3's real line, profiles/r3_bench_line.json, with a multi-rank `comm` block added —
through the printer and check what the driver will see."""
import io
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchmarks import result_line as rl  # noqa: E402


def _canned(world=1):
    full = json.load(open(os.path.join(ROOT, "profiles", "r3_bench_line.json")))
    assert len(json.dumps(full)) > 8192                  # the record that broke the parser
    if world > 1:
        full["n_gpus"] = world
        full["comm"] = {"backend": "nccl", "world_observed": world, "ranks": [f"AMD Instinct MI355X (device {i})" for i in range(world)],
                        "rccl_version": "2.26.6", "xgmi_link_peak_GBps": 153.0, "per_case": "x" * 300}
        full.setdefault("extras", {}).setdefault("compute_comm_bf16", {})[f"gemm_allreduce_M4096_K28672_N8192_tp{world}_auto"] = {
            "us": 400.123456, "aggregate_tflops": 9621.5, "speedup_vs_tp1": 6.7, "algorithm": "direct", "exposed_exchange_us": 31.0,
            "payload_MB_per_rank": 67.1}
    return full


@pytest.mark.parametrize("world", [1, 8])
def test_last_line_is_compact_parseable_and_complete(tmp_path, world):
    buf = io.StringIO()
    rl.emit(_canned(world), out=buf, extras_path=str(tmp_path / "bench_extras.json"))
    lines = buf.getvalue().splitlines()
    assert lines[-2].startswith("EXTRAS {")
    last = lines[-1]
    assert len(last.encode()) <= rl.MAX_LINE_BYTES
    rec = json.loads(last)
    for k in rl.CONTRACT_KEYS:
        assert k in rec, k
    assert rec["config"]["workload"].startswith("MojoPagedDecodeGQA")
    for k in rl.ROOFLINE_KEYS:
        assert k in rec["roofline"], k
    assert rec["roofline"]["frac"] == pytest.approx(rec["roofline"]["achieved"] / rec["roofline"]["peak"], rel=1e-4)
    for k in rl.CPU_KEYS:
        assert k in rec["cpu_baseline"], k
    assert set(rec["roofline_group_gemm"]) <= set(rl.GROUP_GEMM_KEYS) and len(rec["roofline_group_gemm"]) <= 6
    assert "extras" not in rec and "cpu_baseline_per_op" not in rec
    if world > 1:
        assert rec["comm"]["world_observed"] == world
        assert rec["comm"]["gemm_allreduce_M4096_K28672_N8192"]["speedup_vs_tp1"] == 6.7
        assert "ranks" not in rec["comm"]
    # everything else is in the file and on the EXTRAS line
    side = json.load(open(tmp_path / "bench_extras.json"))
    assert "extras" in side and "cpu_baseline_per_op" in side
    assert json.loads(lines[-2][len("EXTRAS "):])["extras"].keys() == side["extras"].keys()
    # the 8 KB tail the driver keeps holds the whole last line
    assert buf.getvalue()[-8192:].splitlines()[-1] == last


def test_oversized_or_incomplete_records_degrade_to_a_parseable_line(tmp_path):
    """a missing field or an oversized block must not cost the result line (rank 0 used to raise before printing
    anything): optional blocks are dropped, long strings clipped, missing contract fields named — the line always parses and
    always fits.  `strict=True` keeps the assertions for these unit tests."""
    full = _canned(8)
    full["config"] = {f"k{i}": "v" * 190 for i in range(40)}
    with pytest.raises(ValueError, match="bytes"):
        rl.compact_line(full, strict=True)
    buf = io.StringIO()
    rl.emit(full, out=buf, extras_path=str(tmp_path / "x.json"))
    last = buf.getvalue().splitlines()[-1]
    rec = json.loads(last)
    assert len(last.encode()) <= rl.MAX_LINE_BYTES and rec["dropped"]
    assert rec["metric"] == full["metric"] and rec["value"] == pytest.approx(full["value"], rel=1e-5) and rec["n_gpus"] == 8
    # a long error string in the comm block: the block goes, the contract stays
    full = _canned(8)
    full["comm"]["error"] = "E" * 100000
    full["roofline"]["kernel"] = "k" * 5000
    rec = json.loads(rl.compact_line(full))
    assert all(k in rec for k in rl.CONTRACT_KEYS) and "roofline" in rec
    # a missing contract field: named, not fatal
    full = _canned()
    del full["roofline"]
    with pytest.raises(ValueError, match="roofline"):
        rl.compact_line(full, strict=True)
    rec = json.loads(rl.compact_line(full))
    assert rec["roofline"] is None and rec["incomplete"] == ["roofline"] and rec["value"] > 0


@pytest.mark.gpu
def test_bench_last_line_parses_on_the_gpu_box():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--no-extras"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    last = p.stdout.strip().splitlines()[-1]
    assert len(last.encode()) <= rl.MAX_LINE_BYTES
    rec = json.loads(last)
    assert rec["steps"] == 5 and rec["warmup"] == 2 and rec["n_gpus"] == 1
    assert rec["value"] > 0 and 0 < rec["roofline"]["frac"] < 1
    assert rec["cpu_baseline"]["value"] > 0 and rec["cpu_baseline"]["kind"] == "port"


@pytest.mark.parametrize("world,full", [(1, False), (2, False), (8, False), (8, True)])
def test_multi_rank_runs_skip_the_single_gpu_extras(monkeypatch, world, full):
    """bench.py at N > 1 measures the second headline (GroupGemm: the line's `roofline_group_gemm`) and the GEMM + collective
    cases only; the single-GPU cases are on the N = 1 record.  MOJO_BENCH_FULL_EXTRAS=1 repeats them on every rank."""
    import torch

    from benchmarks import extras

    ran = []
    for name in [n for n in dir(extras) if n.startswith("bench_")]:
        monkeypatch.setattr(extras, name, (lambda n: lambda *a, **k: ran.append(n) or {"stub": n})(name))
    monkeypatch.setattr(torch.cuda, "empty_cache", lambda: None)
    monkeypatch.setenv("MOJO_BENCH_COMM_INPROC", "1")
    if full:
        monkeypatch.setenv("MOJO_BENCH_FULL_EXTRAS", "1")
    else:
        monkeypatch.delenv("MOJO_BENCH_FULL_EXTRAS", raising=False)
    out = extras.run_extras("cpu", world, 0)
    assert out["MojoGroupGemm_bf16"] == {"stub": "bench_group_gemm"} and out["compute_comm_bf16"] == {"stub": "bench_compute_comm"}
    if world == 1 or full:
        assert "MojoPagedDecodeMLA_bf16" in out and "decode_layer_bf16" in out and len(ran) >= 14
    else:
        assert ran == ["bench_group_gemm", "bench_compute_comm"] and "note" in out
