"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's bond-change masks (never imported by the product).

Follows ``pdb2reaction/bond_changes.py:142-187`` (compare_structures): T = bond_factor*(r_i+r_j); eps = margin_fraction*T;
A = (D <= T - eps) on the upper triangle; need = |D2 - D1| >= delta_fraction*T; formed = ~A1 & A2 & need;
broken = A1 & ~A2 & need.  Distances by direct differences in float64 (torch.cdist is not available bit-for-bit; the
reference has no fixtures for this function -- PARITY UNPINNED, tolerance 1e-12 relative on distances).
"""
import numpy as np


def compare(r1, r2, cov, bond_factor=1.20, margin_fraction=0.05, delta_fraction=0.05):
    r1 = np.asarray(r1, np.float64).reshape(-1, 3)
    r2 = np.asarray(r2, np.float64).reshape(-1, 3)
    cov = np.asarray(cov, np.float64)
    n = len(r1)
    d1 = np.sqrt(((r1[:, None, :] - r1[None, :, :]) ** 2).sum(-1))
    d2 = np.sqrt(((r2[:, None, :] - r2[None, :, :]) ** 2).sum(-1))
    T = bond_factor * (cov[:, None] + cov[None, :])
    eps = margin_fraction * T
    up = np.triu(np.ones((n, n), bool), 1)
    a1 = (d1 <= (T - eps)) & up
    a2 = (d2 <= (T - eps)) & up
    need = (np.abs(d2 - d1) >= delta_fraction * T) & up
    formed = (~a1) & a2 & need
    broken = a1 & (~a2) & need
    code = np.zeros((n, n), np.uint8)
    code[formed] = 1
    code[broken] = 2
    return d1, d2, code
