"""Tolerance semantics of the reference's parity helper.

Restates `mojo_opset/utils/acc.py:12-61` (`check_tol_diff`): nested tuple/list results
are walked pairwise with per-index tolerances (:28-38); ``mixed_tol`` splits on
``|ref| < 1`` and applies atol 2**-6 below / rtol 2**-6 above (:40-44); ``ptol < 1``
accepts when the fraction of `isclose` elements reaches ptol (:46-59); otherwise a
plain fp32 `assert_close` (:61).
"""
import torch

_MIXED = 2.0 ** -6


def _nth(tol, i):
    if isinstance(tol, (tuple, list)):
        if i >= len(tol):
            raise IndexError(f"Tolerance tuple/list index {i} out of range for value {tol}.")
        return tol[i]
    return tol


def check_tol_diff(norm, ref, atol=1e-2, rtol=1e-2, ptol=1.0, mixed_tol=False):
    if isinstance(norm, (tuple, list)):
        for i, (n_i, r_i) in enumerate(zip(norm, ref)):
            check_tol_diff(n_i, r_i, _nth(atol, i), _nth(rtol, i), _nth(ptol, i), _nth(mixed_tol, i))
        return

    if mixed_tol:
        small = ref.abs() < 1.0
        torch.testing.assert_close(norm[small], ref[small], atol=_MIXED, rtol=0)
        torch.testing.assert_close(norm[~small], ref[~small], atol=0, rtol=_MIXED)
        return

    if ptol != 1.0:
        assert ptol < 1.0, f"{ptol=} should <= 1.0"
        ok = torch.isclose(norm, ref, rtol=rtol, atol=atol)
        total = ok.numel()
        match = int(ok.sum())
        match_ratio = match / total
        assert match_ratio >= ptol, (
            f"{match_ratio=:.5%} ({match=} / mismatch={total - match} / {total=}) is under {ptol=:%}, Please Check!"
        )
        return

    torch.testing.assert_close(norm.to(torch.float32), ref.to(torch.float32), atol=atol, rtol=rtol)
