"""ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT.

CPU restatement (plain PyTorch, fp32 softmax / reference rounding points) of the reference's
torch-native golden backend for every op on the hot path.  Importing this
package registers a ``Torch<Op>`` class — backend name ``"torch"`` — into the registry of each
`mojo_opset_amd.core.Mojo<Op>`, which is how the parity tests obtain
``Mojo<Op>._registry.get("torch")`` exactly as the reference's tests do.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  Nothing under ``mojo_opset_amd/`` imports it: the product path never computes on the CPU.

Parity pin: every function here is checked bit-for-bit (`torch.equal`) against outputs captured
from the imported reference (`/root/reference`, torch 2.10.0+rocm7.0 CPU) — see
``oracle/make_golden.py`` (the generator, run in the authoring container) and
``tests/golden/*.pt`` (the captured vectors) — plus the reference's own known-answer tests
re-stated in ``tests/test_oracle_kat.py``.  One exception: fp8 `MojoQuantGemm` has no reference
implementation — **parity unpinned** for that dtype (its restatement is checked against an
independent float64 formula only).
"""
from .torch_golden import *  # noqa: F401,F403
from .torch_golden import __all__  # noqa: F401
