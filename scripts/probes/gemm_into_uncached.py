"""Round 5: does the GEMM of a GEMM + collective operator pay for writing its tile into the UNCACHED peer buffer of the direct
exchange?  Local GEMM of GemmAllReduce at tp 8 (M 4096 / 2048, K 3584, N 8192, bf16) into (a) an ordinary tensor, (b) an
uncached allocation (mojo_hip_peer_alloc(uncached=1), what comm/peer.py shares over HIP-IPC), (c) a plain hipMalloc one."""
import ctypes, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.extras import _time
from mojo_opset_amd.backends.hip import lib as L
from mojo_opset_amd.backends.hip.operators.compute_with_comm import _ENGINE
from mojo_opset_amd.comm.peer import _DeviceBytes
dev = torch.device("cuda", 0)
lib = L.load()
out = {}
for m, k, n in ((4096, 3584, 8192), (2048, 3584, 8192), (4096, 1024, 8192), (1024, 3584, 8192), (512, 3584, 8192), (512, 1024, 8192), (256, 3584, 8192)):   # (the last rows: the 128-row tiles of gemm_tile128_core.h, 8-byte stores)
    x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    w = torch.randn(k, n, device=dev, dtype=torch.bfloat16) * 0.02
    nbytes = m * n * 2
    rec = {}
    bufs = {"tensor": torch.empty(m, n, device=dev, dtype=torch.bfloat16)}
    for kind, flag in (("uncached", 1), ("hipMalloc", 0)):
        p = ctypes.c_void_p()
        L.check(lib.mojo_hip_peer_alloc(ctypes.byref(p), nbytes, flag), "alloc")
        bufs[kind] = torch.as_tensor(_DeviceBytes(p.value, nbytes), device=dev).view(torch.bfloat16).view(m, n)
    ref = None
    for rep in range(2):
        for kind, dst in bufs.items():
            _ENGINE(x, w, None, True, out=dst)
            if ref is None:
                ref = dst.clone()
            assert torch.equal(dst, ref)
            t = _time(lambda: _ENGINE(x, w, None, True, out=dst), 20, 5)
            form = L.last_launch()
            rec.setdefault(kind, []).append(round(t * 1e6, 1))
    out[f"{m}x{k}x{n}"] = {a: {"us": min(v), "tflops": round(2.0 * m * k * n / (min(v) * 1e-6) / 1e12)} for a, v in rec.items()}
    out[f"{m}x{k}x{n}"]["form"] = form
    print(json.dumps({f"{m}x{k}x{n}": out[f"{m}x{k}x{n}"]}), flush=True)
