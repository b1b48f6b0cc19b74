"""Wire formats and config plumbing either side of the hot path.

Restated from the reference (file:line relative to the reference root); pure Python/numpy, no third-party IO:

* XYZ ``.trj``: concatenated XYZ frames whose comment line holds the energy in Hartree as ``f"{E:.12f}"`` and whose
  coordinates are ``"{sym} {x:.15f} {y:.15f} {z:.15f}"`` in Angstrom -- ``path_opt.py:276-290`` (ASE images),
  ``path_opt.py:983-1004`` / ``path_search.py:407-423`` (pysisyphus images).
* reader ``read_energies_xyz``: first decimal number on the comment line, exponents not parsed -- ``trj2fig.py:86-109``.
* ``deep_update`` / ``apply_yaml_overrides`` / ``load_yaml_dict``: defaults <- CLI <- YAML precedence -- ``utils.py:243-313``.
* HEI rule lives in :func:`pdb2reaction_amd.string.select_hei_index` (``path_opt.py:259-273``).
"""
from __future__ import annotations

import re
from pathlib import Path
from typing import Any, Dict, List, Mapping, Optional, Sequence, Tuple, Union

import numpy as np

PathLike = Union[str, Path]


# ---- XYZ / .trj ------------------------------------------------------------------------------------
def xyz_block(symbols: Sequence[str], coords_ang: np.ndarray, comment: str = "") -> str:
    c = np.asarray(coords_ang, dtype=float).reshape(-1, 3)
    if len(symbols) != len(c):
        raise ValueError(f"{len(symbols)} symbols for {len(c)} coordinates")
    lines = [str(len(symbols)), comment]
    lines.extend(f"{sym} {x:.15f} {y:.15f} {z:.15f}" for sym, (x, y, z) in zip(symbols, c))
    return "\n".join(lines) + "\n"


def write_trj_with_energy(symbols: Sequence[str], images_ang: Sequence[np.ndarray], energies_hartree: Sequence[float],
                          path: PathLike) -> None:
    """Write an XYZ ``.trj`` with the energy on line 2 of every frame (reference ``path_opt.py:276-290``)."""
    e = np.array(energies_hartree, dtype=float)
    if len(e) != len(images_ang):
        raise ValueError("one energy per image is required")
    with open(path, "w") as f:
        f.write("".join(xyz_block(symbols, img, f"{ei:.12f}") for img, ei in zip(images_ang, e)))


def write_xyz(symbols: Sequence[str], coords_ang: np.ndarray, path: PathLike, energy_hartree: Optional[float] = None) -> None:
    """Single frame, e.g. ``hei.xyz`` (comment = energy when given)."""
    with open(path, "w") as f:
        f.write(xyz_block(symbols, coords_ang, "" if energy_hartree is None else f"{energy_hartree:.12f}"))


def read_energies_xyz(fname: PathLike) -> List[float]:
    """Hartree energies from the comment line of each frame (reference ``trj2fig.py:86-109``)."""
    energies: List[float] = []
    with open(fname, encoding="utf-8") as fh:
        while (hdr := fh.readline()):
            try:
                nat = int(hdr.strip())
            except ValueError:
                break
            comment = fh.readline().strip()
            m = re.search(r"(-?\d+(?:\.\d+)?)", comment)
            if not m:
                raise RuntimeError(f"Energy not found in comment: {comment}")
            energies.append(float(m.group(1)))
            for _ in range(nat):
                fh.readline()
    if not energies:
        raise RuntimeError(f"No energy data in {fname}")
    return energies


def read_trj(fname: PathLike) -> Tuple[List[str], np.ndarray, List[str]]:
    """All frames of an XYZ ``.trj``: (symbols, coords [K,N,3] Angstrom, comment lines)."""
    symbols: List[str] = []
    frames, comments = [], []
    with open(fname, encoding="utf-8") as fh:
        while (hdr := fh.readline()):
            if not hdr.strip():
                continue
            nat = int(hdr.strip())
            comments.append(fh.readline().rstrip("\n"))
            syms, xyz = [], []
            for _ in range(nat):
                parts = fh.readline().split()
                syms.append(parts[0])
                xyz.append([float(v) for v in parts[1:4]])
            if symbols and syms != symbols:
                raise ValueError("atom order changes between frames")
            symbols = syms
            frames.append(xyz)
    if not frames:
        raise RuntimeError(f"No frames in {fname}")
    return symbols, np.asarray(frames, dtype=float), comments


# ---- YAML precedence (defaults <- CLI <- YAML) ---------------------------------------------------------
def deep_update(dst: Dict[str, Any], src: Optional[Mapping[str, Any]]) -> Dict[str, Any]:
    """Recursively update mapping *dst* with *src*, returning *dst* (reference ``utils.py:243-252``)."""
    for k, v in (src or {}).items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            deep_update(dst[k], v)
        else:
            dst[k] = v
    return dst


def _get_mapping_section(cfg: Mapping[str, Any], path: Sequence[str]) -> Optional[Dict[str, Any]]:
    cur: Any = cfg
    for key in path:
        if not isinstance(cur, Mapping):
            return None
        cur = cur.get(key)
        if cur is None:
            return None
    return cur if isinstance(cur, dict) else None


def apply_yaml_overrides(yaml_cfg: Mapping[str, Any],
                         overrides: Sequence[Tuple[Dict[str, Any], Sequence[Sequence[str]]]]) -> None:
    """For every (target, candidate paths): deep-merge the FIRST existing YAML section (reference ``utils.py:266-297``)."""
    for target, paths in overrides:
        for path in paths:
            section = _get_mapping_section(yaml_cfg, tuple(path))
            if section is not None:
                deep_update(target, section)
                break


def load_yaml_dict(path: Optional[PathLike]) -> Dict[str, Any]:
    """YAML file whose root must be a mapping; ``{}`` when *path* is falsy (reference ``utils.py:300-313``)."""
    if not path:
        return {}
    import yaml

    with open(path, "r") as f:
        data = yaml.safe_load(f) or {}
    if not isinstance(data, dict):
        raise ValueError(f"YAML root must be a mapping, got: {type(data)}")
    return data
