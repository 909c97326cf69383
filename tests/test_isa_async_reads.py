"""The compiled kernels never touch a register that an LDS read issued from inline asm is still going to write.

Several kernels issue `ds_read*` from inline asm (hipcc waits `vmcnt(0)` in front of the transposed-read builtins when LDS-DMA
requests are in flight, and the tile loops count their own `lgkmcnt`).  To the compiler such a destination holds its value
when the asm ends; the hardware writes it later.  If the compiler believes the value dead it hands the register to something
else — round 5: the last K step of `gemm_tile128.hip` fetched fragments nobody multiplies, hipcc put the epilogue's output
address into one of their registers, the LDS data landed on it, and the store went wild (a GPU fault at K = 64).
`scripts/check_async_lds_reads.py` walks the control-flow graph of the compiled assembly from every such read to the wait
that retires it; here it runs on every source file that uses the pattern, compiled with the library's flags.
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mojo_opset_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "scripts"))

FILES = ["gemm_tile128.hip", "gemm_mfma256.hip", "quant_gemm.hip", "mla_attn.hip", "mla_prefill.hip", "paged_prefill_gqa.hip",
         "paged_decode_gqa.hip"]


def _hipcc():
    return shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else None)


def _compile(args):
    src, out = args
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fno-gpu-rdc", "-ffp-contract=on", "-w",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, src)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    return src, out, r.returncode, r.stderr[-2000:]


@pytest.mark.skipif(_hipcc() is None, reason="no hipcc")
def test_no_register_of_an_in_flight_asm_lds_read_is_touched(tmp_path):
    import check_async_lds_reads as chk

    every_source = {f for f in os.listdir(CSRC) if f.endswith(".hip")}
    assert set(FILES) <= every_source
    with concurrent.futures.ThreadPoolExecutor(max_workers=4) as pool:
        results = list(pool.map(_compile, [(f, str(tmp_path / (f + ".s"))) for f in FILES]))
    total = 0
    for src, out, rc, err in results:
        assert rc == 0, (src, err)
        reads, report = chk.check_file(out)
        assert not report, (src, report[:5])
        total += reads
    assert total > 1000                                     # the pattern is really there (and the parser still sees it)


def test_the_checker_sees_the_hazard_it_was_written_for(tmp_path):
    """A reduced listing of the round-5 fault: the asm's destination v[4:5] becomes an address before the wait."""
    import check_async_lds_reads as chk

    bad = """
k_bad:
\ts_waitcnt lgkmcnt(0)
\t;;#ASMSTART
\tds_read_b64_tr_b16 v[2:3], v103 offset:0
\tds_read_b64_tr_b16 v[4:5], v102 offset:0
\t;;#ASMEND
\tv_mfma_f32_16x16x32_bf16 v[92:95], v[10:13], v[70:73], v[92:95]
\ts_cbranch_scc1 .LBB0_2
\tv_mad_u64_u32 v[4:5], s[6:7], s2, v104, v[156:157]
.LBB0_2:
\ts_waitcnt vmcnt(0) lgkmcnt(0)
\tglobal_store_dwordx2 v[4:5], v[92:93], off
\ts_endpgm
.Lfunc_end0:
"""
    good = bad.replace("\tv_mad_u64_u32 v[4:5], s[6:7], s2, v104, v[156:157]\n", "")
    counted = bad.replace("\tv_mad_u64_u32 v[4:5], s[6:7], s2, v104, v[156:157]\n",
                          "\tds_read_b128 v[20:23], v9\n\ts_waitcnt lgkmcnt(1)\n\tv_mad_u64_u32 v[4:5], s[6:7], s2, v104, v[156:157]\n")
    for name, text, want in (("bad", bad, 1), ("good", good, 0), ("counted", counted, 0)):
        p = tmp_path / (name + ".s")
        p.write_text(text)
        reads, report = chk.check_file(str(p))
        assert reads == 2 and len(report) == want, (name, report)
