"""Secondary measurements reported under ``extras`` on bench.py's JSON line: every other op of the
hot path at the BASELINE shapes, as absolute rate and fraction of its roofline (HBM 8 TB/s or bf16
MFMA 2.5 PFLOP/s dense).  Timing: HIP events on the launch stream, >= 20 launches after warm-up."""
import torch

import mojo_opset_amd as mo

HBM_PEAK_GBS = 8000.0
MFMA_BF16_PEAK_TFLOPS = 2500.0


def _time(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def hip(name):
    return getattr(mo, name).get_backend_impl("hip", strict=True)


def group_gemm_case(device, m, k, n, groups, trans, split="balanced", dtype=torch.bfloat16):
    x = torch.randn(m, k, device=device, dtype=dtype)
    w = torch.randn(groups, n, k, device=device, dtype=dtype) if trans else torch.randn(groups, k, n, device=device, dtype=dtype)
    if split == "balanced":
        counts = torch.full((groups,), m // groups, dtype=torch.int32)
    else:  # one expert takes half of the rows
        counts = torch.full((groups,), (m // 2) // (groups - 1), dtype=torch.int32)
        counts[0] = m - int(counts[1:].sum())
    counts[-1] += m - int(counts.sum())
    counts = counts.to(device)
    op = hip("MojoGroupGemm")(w, trans)
    t = _time(lambda: op(x, counts))
    tf = 2.0 * m * k * n / t / 1e12
    return {"us": t * 1e6, "tflops": tf, "frac_of_mfma_peak": tf / MFMA_BF16_PEAK_TFLOPS}


def run_extras(device, world):
    out = {}
    gg = {}
    for name, (m, k, n, g, trans, split) in {
        "ref_case_20480x4096x4096_G8_KN": (20480, 4096, 4096, 8, False, "balanced"),
        "mixtral_up_16384x4096x28672_G8_KN": (16384, 4096, 28672, 8, False, "balanced"),
        "mixtral_up_16384x4096x28672_G8_NK": (16384, 4096, 28672, 8, True, "balanced"),
        "mixtral_down_16384x14336x4096_G8_KN": (16384, 14336, 4096, 8, False, "balanced"),
        "mixtral_up_16384_skewed_KN": (16384, 4096, 28672, 8, False, "skewed"),
        "mixtral_up_4096x4096x28672_G8_KN": (4096, 4096, 28672, 8, False, "balanced"),
    }.items():
        try:
            gg[name] = group_gemm_case(device, m, k, n, g, trans, split)
        except Exception as e:
            gg[name] = {"error": repr(e)}
        torch.cuda.empty_cache()
    out["MojoGroupGemm_bf16"] = gg
    return out
