"""The result line of ``bench.py``: a compact last stdout line the driver parses, everything else beside it.

The driver keeps an 8 KB tail of stdout and parses the LAST line as JSON (round 3's ~24 KB line was not parsed).  So:

* ``compact_line(full)`` keeps exactly the contract fields (metric … config, ``roofline``, ``cpu_baseline``, a six-key
  ``roofline_group_gemm``, and on a multi-rank run a small ``comm`` block) and is asserted to stay below ``MAX_LINE_BYTES``;
* ``emit(full)`` writes the whole record to ``bench_extras.json`` next to ``bench.py``, prints it on an EARLIER stdout line
  prefixed ``EXTRAS `` and prints the compact line LAST.

Per-op accounting follows the reference's own benchmark records (latency + bytes / FLOPs per call,
``/root/reference/mojo_opset/benchmark/api.py:119-145``); the full record keeps those per case.
"""
import json
import os
import sys

MAX_LINE_BYTES = 4096

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config")
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "algorithmic_bytes_per_launch",
                 "device_us_per_launch", "kernel")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample")
GROUP_GEMM_KEYS = ("bound", "achieved", "peak", "unit", "frac", "device_us_per_launch")
REQUIRED = CONTRACT_KEYS + ("roofline",)


def _clip(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[: n - 3] + "..."


def _num(v, digits=6):
    """Floats rounded to a few significant digits: the line is for reading and parsing, not for bit-exact storage."""
    if isinstance(v, float):
        return float(f"{v:.{digits}g}")
    return v


def _pick(d, keys, clip=160):
    return {k: _clip(_num(d.get(k)), clip) for k in keys if k in d}


def _comm_summary(full):
    """≤ 8 keys describing the fabric and the GemmAllReduce case BASELINE.json names (M 4096, K 28672, N 8192)."""
    comm = full.get("comm")
    if not isinstance(comm, dict):
        return None
    out = {k: comm.get(k) for k in ("backend", "world_observed", "rccl_version") if k in comm}
    if "error" in comm:
        out["error"] = _clip(comm["error"], 160)
    cc = (full.get("extras") or {}).get("compute_comm_bf16") if isinstance(full.get("extras"), dict) else None
    if isinstance(cc, dict):
        world = full.get("n_gpus", 1)
        base = f"gemm_allreduce_M4096_K28672_N8192_tp{world}"
        case = next((cc[k] for k in (base + "_auto", base, base + "_rccl") if isinstance(cc.get(k), dict) and "us" in cc[k]), None)
        if isinstance(case, dict):
            out["gemm_allreduce_M4096_K28672_N8192"] = {
                k: _num(case[k]) for k in ("us", "aggregate_tflops", "speedup_vs_tp1", "algorithm", "exposed_exchange_us")
                if k in case}
    return out


def compact_line(full, strict=False):
    """The contract fields of ``full`` and nothing else, ALWAYS as a line the driver can parse (a missing field
    or an oversized block used to raise before anything was printed, i.e. rank 0 died without a result line).  A missing
    contract field is filled with ``None`` and named under ``"incomplete"``; a line above ``MAX_LINE_BYTES`` drops its
    optional blocks one by one (``comm``, ``roofline_group_gemm``, ``cpu_baseline``, then long strings) and says so under
    ``"dropped"``.  ``strict=True`` (the unit tests) raises instead."""
    line = {k: _num(full[k]) for k in CONTRACT_KEYS if k in full}
    if isinstance(line.get("config"), dict):
        line["config"] = {k: _clip(v, 200) for k, v in line["config"].items()}
    if isinstance(full.get("roofline"), dict):
        line["roofline"] = _pick(full["roofline"], ROOFLINE_KEYS)
    if isinstance(full.get("cpu_baseline"), dict):
        line["cpu_baseline"] = _pick(full["cpu_baseline"], CPU_KEYS)
    if isinstance(full.get("roofline_group_gemm"), dict):
        line["roofline_group_gemm"] = _pick(full["roofline_group_gemm"], GROUP_GEMM_KEYS)
    comm = _comm_summary(full)
    if comm:
        line["comm"] = comm
    if "extras_file" in full:
        line["extras_file"] = full["extras_file"]
    missing = [k for k in REQUIRED if k not in line]
    if missing:
        if strict:
            raise ValueError(f"result line lacks contract fields: {missing}")
        for k in missing:
            line[k] = None
        line["incomplete"] = missing
    text = json.dumps(line, separators=(",", ":"))
    if len(text.encode()) > MAX_LINE_BYTES:
        if strict:
            raise ValueError(f"result line is {len(text.encode())} bytes; the driver parses at most {MAX_LINE_BYTES}")
        dropped = []
        for victim in ("comm", "roofline_group_gemm", "cpu_baseline"):
            if victim in line and len(text.encode()) > MAX_LINE_BYTES:
                del line[victim]
                dropped.append(victim)
                line["dropped"] = dropped
                text = json.dumps(line, separators=(",", ":"))
        clip = 120
        while len(text.encode()) > MAX_LINE_BYTES and clip >= 15:          # long strings, shorter and shorter
            keep = ("metric", "unit", "dtype", "data", "scaling")            # what the driver matches on is never clipped
            line = {k: (v if k in keep else {kk: _clip(vv, clip) for kk, vv in v.items()} if isinstance(v, dict) else _clip(v, clip))
                    for k, v in line.items()}
            line["dropped"] = dropped + [f"strings clipped to {clip}"]
            text = json.dumps(line, separators=(",", ":"))
            clip //= 2
        if len(text.encode()) > MAX_LINE_BYTES:                             # last resort: the bare contract scalars
            line = {k: (line.get(k) if k in ("metric", "unit", "dtype", "data", "scaling") or not isinstance(line.get(k), (dict, list, str))
                        else _clip(str(line.get(k)), 60)) for k in CONTRACT_KEYS}
            line["dropped"] = ["everything but the contract scalars"]
            text = json.dumps(line, separators=(",", ":"))
    return text


def emit(full, out=None, extras_path=None):
    """Write ``bench_extras.json``, print ``EXTRAS <everything>`` and then the compact line as the LAST stdout line."""
    out = out or sys.stdout
    if extras_path is None:
        extras_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench_extras.json")
    full = dict(full)
    try:
        with open(extras_path, "w") as f:
            json.dump(full, f, indent=1)
        full["extras_file"] = os.path.basename(extras_path)
    except OSError as e:                     # a read-only tree must not cost the result line
        full["extras_file"] = f"not written: {e!r}"
    text = compact_line(full)                # (never raises: degrades, see compact_line)
    out.write("EXTRAS " + json.dumps(full, separators=(",", ":")) + "\n")
    out.write(text + "\n")
    out.flush()
    return text
