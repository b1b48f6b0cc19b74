"""Synthetic active-site clusters and reaction-string images.

``make_cluster(N, seed)``: jittered simple-cubic lattice at 0.10 atoms/A^3 (a = 2.154 A, uniform
jitter +-0.3a per axis), the N points nearest the origin; elements drawn i.i.d. with
p(H,C,N,O,S) = (0.50, 0.30, 0.08, 0.11, 0.01).  ``make_images`` interpolates reactant -> product
(Gaussian-bump displacement, sigma 3 A, amplitude 1.5 A) and adds N(0, 0.02 A) noise per image.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

DEFAULT_SEED = 20260130
DENSITY = 0.10
ELEMENTS = ("H", "C", "N", "O", "S")
ELEMENT_Z = (1, 6, 7, 8, 16)
ELEMENT_P = (0.50, 0.30, 0.08, 0.11, 0.01)

SYMBOLS = [
    "X", "H", "He", "Li", "Be", "B", "C", "N", "O", "F", "Ne", "Na", "Mg", "Al", "Si", "P", "S", "Cl", "Ar",
    "K", "Ca", "Sc", "Ti", "V", "Cr", "Mn", "Fe", "Co", "Ni", "Cu", "Zn", "Ga", "Ge", "As", "Se", "Br", "Kr",
    "Rb", "Sr", "Y", "Zr", "Nb", "Mo", "Tc", "Ru", "Rh", "Pd", "Ag", "Cd", "In", "Sn", "Sb", "Te", "I", "Xe",
    "Cs", "Ba", "La", "Ce", "Pr", "Nd", "Pm", "Sm", "Eu", "Gd", "Tb", "Dy", "Ho", "Er", "Tm", "Yb", "Lu",
    "Hf", "Ta", "W", "Re", "Os", "Ir", "Pt", "Au", "Hg", "Tl", "Pb", "Bi", "Po", "At", "Rn", "Fr", "Ra", "Ac",
    "Th", "Pa", "U", "Np", "Pu", "Am", "Cm", "Bk", "Cf", "Es",
]
Z_OF_SYMBOL = {s: i for i, s in enumerate(SYMBOLS)}


def symbols_to_z(elem) -> np.ndarray:
    """Element symbols (any case) -> atomic numbers; mirrors ``e.capitalize()`` of uma_pysis.py:266."""
    try:
        return np.array([Z_OF_SYMBOL[str(e).capitalize()] for e in elem], dtype=np.int32)
    except KeyError as exc:
        raise ValueError(f"unknown element symbol {exc}") from None


def make_cluster(n_atoms: int, seed: int = DEFAULT_SEED) -> Tuple[np.ndarray, np.ndarray]:
    """Return (Z int32 [N], pos float64 [N,3] in Angstrom)."""
    rng = np.random.default_rng(seed)
    a = DENSITY ** (-1.0 / 3.0)
    radius = (3.0 * n_atoms / (4.0 * np.pi * DENSITY)) ** (1.0 / 3.0)
    m = int(np.ceil(radius / a)) + 3
    g = np.arange(-m, m + 1, dtype=np.float64)
    lat = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3) * a
    lat = lat + rng.uniform(-0.3 * a, 0.3 * a, size=lat.shape)
    order = np.argsort(np.einsum("ij,ij->i", lat, lat), kind="stable")[:n_atoms]
    pos = lat[order]
    z = rng.choice(np.array(ELEMENT_Z, dtype=np.int32), size=n_atoms, p=np.array(ELEMENT_P))
    return z.astype(np.int32), pos


def make_product(pos: np.ndarray, seed: int = DEFAULT_SEED + 1, sigma: float = 3.0, amp: float = 1.5) -> np.ndarray:
    rng = np.random.default_rng(seed)
    d = rng.standard_normal(pos.shape)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    w = amp * np.exp(-0.5 * np.einsum("ij,ij->i", pos, pos) / sigma ** 2)
    return pos + d * w[:, None]


def make_images(n_atoms: int, n_images: int, seed: int = DEFAULT_SEED, noise: float = 0.02):
    """Return (Z [N], images float64 [K,N,3] Angstrom, frozen atom indices list)."""
    z, r = make_cluster(n_atoms, seed)
    p = make_product(r, seed + 1)
    imgs = []
    for k in range(n_images):
        t = k / max(n_images - 1, 1)
        rng = np.random.default_rng(seed + 2 + k)
        imgs.append((1.0 - t) * r + t * p + noise * rng.standard_normal(r.shape))
    imgs = np.stack(imgs)
    n_frozen = min(20, max(n_atoms // 10, 0))
    frozen = np.argsort(-np.einsum("ij,ij->i", r, r), kind="stable")[:n_frozen]
    return z, imgs, sorted(int(i) for i in frozen)


def count_edges(pos: np.ndarray, cutoff: float = 6.0) -> int:
    """Directed edge count (O(N^2) in blocks); diagnostic only."""
    n = len(pos)
    tot = 0
    for s in range(0, n, 1024):
        d2 = ((pos[s:s + 1024, None, :] - pos[None, :, :]) ** 2).sum(-1)
        tot += int(((d2 <= cutoff * cutoff) & (d2 > 0)).sum())
    return tot
