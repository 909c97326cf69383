"""The C-ABI library must load on a CPU-only host and export exactly what include/mojo_hip.h declares; the
ctypes table must mirror the header.  No compute call is made here."""
import ctypes
import os
import re

from conftest import ROOT
from mojo_opset_amd.backends.hip import lib as L

HEADER = os.path.join(ROOT, "include", "mojo_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mojo_hip_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    names = declared_symbols()
    assert len(names) >= 20
    for must in ("mojo_hip_paged_decode_gqa", "mojo_hip_paged_prefill_gqa", "mojo_hip_mla_latent_attn",
                 "mojo_hip_residual_add_rmsnorm", "mojo_hip_swiglu", "mojo_hip_apply_rope",
                 "mojo_hip_rotary_embedding", "mojo_hip_store_paged_kv_plan", "mojo_hip_store_paged_kv_layout",
                 "mojo_hip_group_gemm", "mojo_hip_quant_gemm", "mojo_hip_gemm", "mojo_hip_gemm_rowmap"):
        assert must in names


def test_library_exports_every_declared_symbol():
    assert os.path.exists(L.lib_path()), "build the library first: python -m mojo_opset_amd.csrc.build"
    handle = ctypes.CDLL(L.lib_path())
    for name in declared_symbols():
        assert hasattr(handle, name), f"{name} is declared in mojo_hip.h but not exported"


def test_ctypes_table_mirrors_the_header():
    assert sorted(L.SIGNATURES) == declared_symbols()
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, (_, argtypes) in L.SIGNATURES.items():
        m = re.search(r"\b" + name + r"\s*\((.*?)\)\s*;", text, flags=re.S)
        assert m, name
        params = m.group(1).strip()
        n_params = 0 if params in ("", "void") else params.count(",") + 1
        assert n_params == len(argtypes), f"{name}: header has {n_params} parameters, ctypes table {len(argtypes)}"


def test_loader_reports_version_and_error_text():
    h = L.load()
    assert b"gfx950" in h.mojo_hip_version()
    assert isinstance(h.mojo_hip_last_error(), bytes)
