"""Minimal fully-grown string iteration (the step that consumes the batched E/F of all images).

This is the *caller* side of the hot path, restated only as far as the
headline metric needs it: one iteration = E+F of every image + projection + step + equal-arc
reparametrisation (pysisyphus GrowingString/StringOptimizer semantics, SURVEY.md Appendix B:
perpendicular force F - (F.t)t along spline tangents, step scaled to ``max_step``, ``param="equi"``).
All tensor math is device-agnostic torch so every rank replays the identical update.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch


def tangents(x: torch.Tensor) -> torch.Tensor:
    """Unit tangents of a (K, D) string: central differences, one-sided at the ends."""
    t = torch.empty_like(x)
    t[1:-1] = x[2:] - x[:-2]
    t[0] = x[1] - x[0]
    t[-1] = x[-1] - x[-2]
    return t / t.norm(dim=1, keepdim=True).clamp_min(1e-30)


def perpendicular(f: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    return f - (f * t).sum(dim=1, keepdim=True) * t


def reparametrize_equal(x: torch.Tensor) -> torch.Tensor:
    """Redistribute images to equal arc length along the piecewise-linear string (endpoints kept)."""
    k = x.shape[0]
    seg = (x[1:] - x[:-1]).norm(dim=1)
    s = torch.cat([seg.new_zeros(1), seg.cumsum(0)])
    target = torch.linspace(0.0, float(s[-1]), k, dtype=x.dtype, device=x.device)
    idx = torch.searchsorted(s, target[1:-1].contiguous(), right=True).clamp(1, k - 1)
    s0, s1 = s[idx - 1], s[idx]
    w = ((target[1:-1] - s0) / (s1 - s0).clamp_min(1e-30)).unsqueeze(1)
    out = x.clone()
    out[1:-1] = x[idx - 1] * (1.0 - w) + x[idx] * w
    return out


def string_step(x: torch.Tensor, f: torch.Tensor, max_step: float = 0.1, alpha: float = 0.5,
                fix_ends: bool = False) -> torch.Tensor:
    """One steepest-descent string update.  x, f: (K, D) coordinates / forces (same units)."""
    t = tangents(x)
    fp = perpendicular(f, t)
    if fix_ends:
        fp[0] = 0.0
        fp[-1] = 0.0
    step = alpha * fp
    big = step.abs().max()
    step = step * torch.clamp(max_step / big.clamp_min(1e-300), max=1.0)       # scaled on the device: no host synchronisation
    return reparametrize_equal(x + step)


def select_hei_index(energies: Sequence[float]) -> int:
    """Highest-energy image preferring internal local maxima (reference ``path_opt.py:259-273``)."""
    e = [float(v) for v in energies]
    n = len(e)
    if n >= 3:
        cand = [i for i in range(1, n - 1) if e[i] > e[i - 1] and e[i] > e[i + 1]]
        if cand:
            return max(cand, key=lambda i: e[i])
        return 1 + max(range(n - 2), key=lambda i: e[1 + i])
    return max(range(n), key=lambda i: e[i])
