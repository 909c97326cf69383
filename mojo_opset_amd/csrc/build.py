"""Build libmojo_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m mojo_opset_amd.csrc.build [--force] [-j N]

Every ``*.hip`` file in this directory is compiled to an object (skipped when up to date) and
linked into ``mojo_opset_amd/lib/libmojo_hip.so`` — in-tree, so it travels with the repo
snapshot to the GPU box.
"""
import argparse
import concurrent.futures
import glob
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OBJ_DIR = os.path.join(HERE, "build")
LIB_DIR = os.path.join(PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libmojo_hip.so")
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH, /opt/rocm/bin/hipcc)")


def hashed_files():
    """The files whose contents the library is stamped with (sorted, relative to the repository root)."""
    root = os.path.dirname(PKG)
    files = (glob.glob(os.path.join(HERE, "*.hip")) + glob.glob(os.path.join(HERE, "*.h")) +
             glob.glob(os.path.join(HERE, "experiments", "*.h")) + glob.glob(os.path.join(root, "include", "*.h")))
    return sorted(os.path.relpath(f, root) for f in files)


def source_hash() -> str:
    """sha256 over names and contents of the kernel sources and the C-ABI header: what `mojo_hip_version()` reports after
    ``src=`` and what ``lib.load()`` checks against the tree it finds the library in."""
    root = os.path.dirname(PKG)
    h = hashlib.sha256()
    for rel in hashed_files():
        h.update(rel.encode() + b"\0")
        with open(os.path.join(root, rel), "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, jobs: int = 4, verbose: bool = True) -> str:
    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    sources = sorted(glob.glob(os.path.join(HERE, "*.hip")))
    headers = (sorted(glob.glob(os.path.join(HERE, "*.h"))) + sorted(glob.glob(os.path.join(HERE, "experiments", "*.h"))) +
               [os.path.join(os.path.dirname(PKG), "include", "mojo_hip.h")])
    flags = ["-O3", f"--offload-arch={ARCH}", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
             "-fno-gpu-rdc", "-ffp-contract=on"]
    if os.environ.get("MOJO_HIP_BUILD_EXPERIMENTS", "") == "1":     # the measured-and-dropped kernels of csrc/experiments/
        flags.append("-DMOJO_HIP_BUILD_EXPERIMENTS")
    flags += os.environ.get("MOJO_HIP_EXTRA_CXXFLAGS", "").split()   # e.g. -DMLA_DBG_TIMERS for kernel phase timers
    # a change of flags rebuilds everything (object staleness is otherwise by mtime)
    flags_file = os.path.join(OBJ_DIR, "flags.txt")
    if not os.path.exists(flags_file) or open(flags_file).read() != " ".join(flags):
        force = True
    src_hash = source_hash()
    stamp_file = os.path.join(OBJ_DIR, "source_hash.txt")
    stamp_changed = not os.path.exists(stamp_file) or open(stamp_file).read() != src_hash

    def compile_one(src):
        obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
        stamped = os.path.basename(src) == "api_common.hip"          # the one file that carries the source hash
        if force or _stale(obj, [src] + headers) or (stamped and stamp_changed):
            cmd = [hipcc, *flags, *([f'-DMOJO_SRC_HASH="{src_hash}"'] if stamped else []), "-c", src, "-o", obj]
            if verbose:
                print("[mojo_hip] hipcc", os.path.basename(src), flush=True)
            res = subprocess.run(cmd, capture_output=True, text=True)
            if res.returncode != 0:
                raise RuntimeError(f"hipcc failed for {src}:\n{res.stdout}\n{res.stderr}")
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, jobs)) as pool:
        objs = list(pool.map(compile_one, sources))
    with open(flags_file, "w") as f:
        f.write(" ".join(flags))
    with open(stamp_file, "w") as f:
        f.write(src_hash)
    if force or _stale(LIB_PATH, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH, *objs]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
        if verbose:
            print("[mojo_hip] linked", LIB_PATH, flush=True)
    return LIB_PATH


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("-j", type=int, default=4)
    ns = ap.parse_args()
    print(build(force=ns.force, jobs=ns.j))
    sys.exit(0)
