"""bench.py on a host without a GPU: it must fail loudly — no CPU fallback, no result line — both as a single process and
as the parent that starts its own ranks for --gpus N (the parent never touches the GPU runtime; it relays the ranks' exit)."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(torch.cuda.is_available(), reason="checks the behaviour on a host WITHOUT a GPU")


def _run(*args):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    return p.returncode, p.stdout, p.stderr


def test_bench_fails_loudly_without_a_gpu():
    rc, out, err = _run("--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline")
    assert rc != 0
    assert '"metric"' not in out
    assert "GPU" in err or "HIP" in err or "cuda" in err.lower()


def test_bench_parent_of_ranks_relays_their_failure():
    rc, out, err = _run("--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline")
    assert rc != 0                                       # the ranks cannot start without GPUs; the parent says so
    assert '"metric"' not in out
