"""The C-ABI library must load on a CPU-only host and export exactly what include/mojo_hip.h declares; the
ctypes table must mirror the header.  No compute call is made here."""
import ctypes
import os
import re

import pytest

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


def _split_top_level(args: str):
    out, depth, cur = [], 0, ""
    for ch in args:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def test_integration_md_stub_passes_what_the_header_declares():
    """The ctypes stub a maintainer copies from INTEGRATION.md must pass exactly the parameters of the header
    (a stale stub once passed the stream where `dtype` belongs)."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    header = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    blocks = re.findall(r"```python\n(.*?)```", doc, flags=re.S)
    calls = 0
    for block in blocks:
        code = re.sub(r"#[^\n]*", "", block)
        for m in re.finditer(r"lib\.(mojo_hip_[a-z0-9_]+)\(", code):
            name, start = m.group(1), m.end()
            depth, i = 1, start
            while depth and i < len(code):
                depth += code[i] in "([{"
                depth -= code[i] in ")]}"
                i += 1
            passed = len(_split_top_level(code[start:i - 1]))
            decl = re.search(r"\b" + name + r"\s*\((.*?)\)\s*;", header, flags=re.S)
            assert decl, f"INTEGRATION.md calls {name}, which mojo_hip.h does not declare"
            params = decl.group(1).strip()
            declared = 0 if params in ("", "void") else params.count(",") + 1
            assert passed == declared, f"INTEGRATION.md passes {passed} arguments to {name}; the header declares {declared}"
            calls += 1
    assert calls >= 3


def test_every_environment_switch_of_the_package_is_documented():
    """INTEGRATION.md §5 lists every MOJO_HIP_* switch the library or the host side reads (a switch a maintainer cannot find
    is a behaviour they cannot reproduce)."""
    import glob
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = set()
    for pat in ("mojo_opset_amd/csrc/*.hip", "mojo_opset_amd/csrc/*.h", "mojo_opset_amd/csrc/experiments/*.h", "mojo_opset_amd/**/*.py"):
        for path in glob.glob(os.path.join(root, pat), recursive=True):
            names.update(re.findall(r"MOJO_HIP_[A-Z0-9_]+", open(path).read()))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    missing = sorted(n for n in names if n not in doc)
    assert len(names) > 30 and not missing, missing


def test_library_is_stamped_with_the_hash_of_the_tree_and_a_stale_one_is_refused(monkeypatch):
    """`mojo_hip_version()` ends in `src=<hash of csrc/ + include/>`; `lib.load()` recomputes the hash from the tree the
    library sits in and refuses a library built from other sources (the prebuilt .so travels with the snapshot)."""
    from mojo_opset_amd.backends.hip import lib
    from mojo_opset_amd.csrc import build as B

    version = lib.load().mojo_hip_version().decode()
    assert version.endswith("src=" + B.source_hash()), version
    assert all(f.endswith((".hip", ".h")) for f in B.hashed_files()) and "include/mojo_hip.h" in B.hashed_files()

    class _Stale:
        @staticmethod
        def mojo_hip_version():
            return b"mojo_hip 0.2.0 (gfx950) src=0000000000000000"

    with pytest.raises(lib.MojoHipError, match="built from other sources"):
        lib._check_source_hash(_Stale, lib.DEFAULT_LIB)
    monkeypatch.setenv("MOJO_HIP_ALLOW_STALE", "1")
    lib._check_source_hash(_Stale, lib.DEFAULT_LIB)              # explicit override
