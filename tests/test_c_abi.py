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


def _switch_table(section):
    """Names in the first column of the table under ``### <section>`` of INTEGRATION.md."""
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    body = doc.split(f"### {section}", 1)[1].split("\n### ", 1)[0].split("\nNot environment switches", 1)[0]
    names = set()
    for line in body.splitlines():
        if line.startswith("| `"):
            names.update(re.findall(r"`(MOJO_[A-Z0-9_]+)`", line.split("|")[1]))
    return names


def test_switches_in_the_binary_are_exactly_the_documented_table():
    """INTEGRATION.md 5.1 == the MOJO_HIP_* names compiled into libmojo_hip.so (59 switches with two
    semantics became <= 20 with one).  A switch a maintainer cannot find is a behaviour they cannot reproduce; a documented
    switch the binary does not read is a lie."""
    import re

    from mojo_opset_amd.backends.hip import lib

    blob = open(lib.lib_path(), "rb").read()
    in_binary = {m.decode() for m in re.findall(rb"MOJO_HIP_[A-Z0-9_]+", blob)}
    table = _switch_table("5.1")
    experiments = _switch_table("5.3")
    if lib.built_with_experiments():
        assert table <= in_binary and in_binary - table <= experiments, (sorted(in_binary - table - experiments), sorted(table - in_binary))
    else:
        assert in_binary == table, (sorted(in_binary - table), sorted(table - in_binary))
        assert len(table) <= 20


def test_every_environment_switch_of_the_package_is_documented():
    """Every MOJO_HIP_* name in the sources is in one of the three tables of INTEGRATION.md section 5: 5.1 (read by the
    library), 5.2 (read by the Python layer; exactly the names `mojo_opset_amd/**/*.py` reads through `switches.get`), 5.3
    (build-time and experiments-only)."""
    import glob
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names, py_reads = set(), set()
    for pat in ("mojo_opset_amd/csrc/*.hip", "mojo_opset_amd/csrc/*.h", "mojo_opset_amd/csrc/experiments/*.h", "mojo_opset_amd/**/*.py"):
        for path in glob.glob(os.path.join(root, pat), recursive=True):
            text = open(path).read()
            names.update(re.findall(r"MOJO_HIP_[A-Z0-9_]+", text))
            if path.endswith(".py"):
                py_reads.update(re.findall(r"(?:switches\.get(?:_int)?|os\.environ\.get)\(\s*\"(MOJO_HIP_[A-Z0-9_]+)\"", text))
    t1, t2, t3 = _switch_table("5.1"), _switch_table("5.2"), _switch_table("5.3")
    missing = sorted(n for n in names if n not in t1 | t2 | t3)
    assert len(names) > 30 and not missing, missing
    assert py_reads - {"MOJO_HIP_BUILD_EXPERIMENTS", "MOJO_HIP_EXTRA_CXXFLAGS"} == t2, (sorted(py_reads - t2), sorted(t2 - py_reads))


def test_switches_are_latched_and_reloaded(monkeypatch):
    """The one semantics of section 5: a value is latched at first use, `switches.reload()` re-reads both layers."""
    import ctypes as C

    from mojo_opset_amd import switches
    from mojo_opset_amd.backends.hip import lib

    h = lib.load()
    monkeypatch.delenv("MOJO_HIP_COMM_CHUNKS", raising=False)
    assert switches.get_int("MOJO_HIP_COMM_CHUNKS", 0) == 0
    os.environ["MOJO_HIP_COMM_CHUNKS"] = "3"                   # behind the fixture's back: no reload
    assert switches.get_int("MOJO_HIP_COMM_CHUNKS", 0) == 0    # still latched
    switches.reload()
    assert switches.get_int("MOJO_HIP_COMM_CHUNKS", 0) == 3
    monkeypatch.delenv("MOJO_HIP_COMM_CHUNKS")
    assert switches.get_int("MOJO_HIP_COMM_CHUNKS", 0) == 0
    # the library's side: mojo_hip_paged_decode_gqa_workspace_bytes reads MOJO_HIP_DECODE_CHUNK (no GPU needed)
    args = (8, 32, 8, 128, 16, 512, 8192)
    base = h.mojo_hip_paged_decode_gqa_workspace_bytes(*args)
    os.environ["MOJO_HIP_DECODE_CHUNK"] = "64"
    try:
        assert h.mojo_hip_paged_decode_gqa_workspace_bytes(*args) == base          # latched
        h.mojo_hip_reload_env()
        assert h.mojo_hip_paged_decode_gqa_workspace_bytes(*args) > base           # 128 chunks of 64 tokens: more partials
        assert lib.switches().get("MOJO_HIP_DECODE_CHUNK") == 64
    finally:
        os.environ.pop("MOJO_HIP_DECODE_CHUNK", None)
        h.mojo_hip_reload_env()
    assert h.mojo_hip_paged_decode_gqa_workspace_bytes(*args) == base
    assert lib.switches().get("MOJO_HIP_DECODE_CHUNK", 0) is None
    assert isinstance(lib.last_launch(), str) and isinstance(lib.launch_history(clear=True), str)
    assert h.mojo_hip_peer_set_timeout_ms(1234) == 0 and h.mojo_hip_peer_set_timeout_ms(0) == 1234


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
