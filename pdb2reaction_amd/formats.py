"""Wire formats and config plumbing either side of the hot path.

Written from the behaviour of the reference (file:line relative to the reference root; exact-text tests in
tests/test_formats.py pin it); pure Python/numpy, no third-party IO:

* XYZ ``.trj``: concatenated XYZ frames whose comment line holds the energy in Hartree as ``f"{E:.12f}"`` and whose
  coordinates are ``"{sym} {x:.15f} {y:.15f} {z:.15f}"`` in Angstrom -- ``path_opt.py:276-290`` (ASE images),
  ``path_opt.py:983-1004`` / ``path_search.py:407-423`` (pysisyphus images).
* reader ``read_energies_xyz``: first decimal number on the comment line, exponents not parsed -- ``trj2fig.py:86-109``.
* ``deep_update`` / ``apply_yaml_overrides`` / ``load_yaml_dict``: defaults <- CLI <- YAML precedence -- ``utils.py:243-313``.
* ``summary.yaml`` of a path search: ``out_dir / n_images / n_segments / segments[index, tag, kind, barrier_kcal, delta_kcal,
  bond_changes]`` (+ optional ``energy_diagrams``), dumped with ``yaml.safe_dump(sort_keys=False, allow_unicode=True)``; the
  bond-change report text becomes a list of one-key mappings (``path_search.py:245-293,2762-2786``).
* energy series of a trajectory: ``recompute_energies`` (all frames in ONE batched engine call), ``transform_series``,
  ``write_energy_csv`` -- ``trj2fig.py:112-205,287-303`` (plotting itself is out of scope).
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


_FIRST_DECIMAL = re.compile(r"-?\d+(?:\.\d+)?")       # sign, digits, optional fraction; an exponent is NOT part of the match


def _frames(lines: Sequence[str], strict_header: bool):
    """Walk concatenated XYZ frames in a list of lines: yields (comment, atom_lines).

    A header that is not an integer ends the walk (``strict_header=False``: what the reference's energy reader does with
    trailing junk) or raises (``True``); blank lines between frames are skipped only in strict mode."""
    i, n = 0, len(lines)
    while i < n:
        head = lines[i].strip()
        if strict_header and not head:
            i += 1
            continue
        try:
            nat = int(head)
        except ValueError:
            if strict_header:
                raise ValueError(f"line {i + 1}: expected an atom count, got {lines[i]!r}")
            return
        comment = lines[i + 1] if i + 1 < n else ""
        yield comment, lines[i + 2: i + 2 + nat]
        i += 2 + nat


def read_energies_xyz(fname: PathLike) -> List[float]:
    """One energy (Hartree) per frame of an XYZ/.trj file: the FIRST plain decimal number on the comment line -- so
    ``-1.25e-3`` reads as ``-1.25`` exactly as in the reference's plotting reader (``trj2fig.py:86-109``).
    RuntimeError when a comment holds no number or the file holds no frame."""
    text = Path(fname).read_text(encoding="utf-8")
    out: List[float] = []
    for comment, _ in _frames(text.split("\n"), strict_header=False):
        hit = _FIRST_DECIMAL.search(comment)
        if hit is None:
            raise RuntimeError(f"Energy not found in comment: {comment.strip()}")
        out.append(float(hit.group(0)))
    if not out:
        raise RuntimeError(f"No energy data in {fname}")
    return out


def read_trj(fname: PathLike) -> Tuple[List[str], np.ndarray, List[str]]:
    """All frames of an XYZ ``.trj``: (symbols, coords [K,N,3] Angstrom, comment lines)."""
    text = Path(fname).read_text(encoding="utf-8")
    symbols: Optional[List[str]] = None
    coords, comments = [], []
    for comment, atoms in _frames(text.split("\n"), strict_header=True):
        cols = [a.split() for a in atoms]
        syms = [c[0] for c in cols]
        if symbols is not None and syms != symbols:
            raise ValueError("atom order changes between frames")
        symbols = syms
        coords.append([[float(v) for v in c[1:4]] for c in cols])
        comments.append(comment.rstrip("\n"))
    if symbols is None:
        raise RuntimeError(f"No frames in {fname}")
    return symbols, np.asarray(coords, dtype=float), comments


# ---- YAML precedence (defaults <- CLI <- YAML) ---------------------------------------------------------
# Behaviour of the reference's helpers (``utils.py:243-313``): a YAML section overrides what defaults and CLI flags have
# put into a config dict; nested dicts merge key by key, everything else (lists included) is replaced wholesale; of
# several candidate section paths only the first one that exists AND is a mapping is used.
def deep_update(dst: Dict[str, Any], src: Optional[Mapping[str, Any]]) -> Dict[str, Any]:
    """Merge *src* into *dst* in place and return *dst*; dict-into-dict merges recurse, any other value overwrites."""
    todo = [(dst, src)] if src else []
    while todo:
        into, frm = todo.pop()
        for key in frm:
            val = frm[key]
            if isinstance(val, dict) and isinstance(into.get(key), dict):
                todo.append((into[key], val))
            else:
                into[key] = val
    return dst


def _section(cfg: Any, path: Sequence[str]) -> Optional[Dict[str, Any]]:
    """The mapping found by walking *path* through nested mappings, else None (missing key, null, or a non-mapping)."""
    node = cfg
    for key in path:
        node = node.get(key) if isinstance(node, Mapping) else None
        if node is None:
            return None
    return node if isinstance(node, dict) else None


def apply_yaml_overrides(yaml_cfg: Mapping[str, Any],
                         overrides: Sequence[Tuple[Dict[str, Any], Sequence[Sequence[str]]]]) -> None:
    """``overrides`` = [(target dict, (path, path, ...)), ...]: each target receives the first of its candidate sections."""
    for target, candidates in overrides:
        found = next((sec for sec in (_section(yaml_cfg, tuple(c)) for c in candidates) if sec is not None), None)
        if found is not None:
            deep_update(target, found)


def load_yaml_dict(path: Optional[PathLike]) -> Dict[str, Any]:
    """Parsed YAML file; no path -> ``{}``, empty file -> ``{}``, a root that is not a mapping -> ValueError."""
    if not path:
        return {}
    import yaml

    data = yaml.safe_load(Path(path).read_text())
    if data is None:
        return {}
    if not isinstance(data, dict):
        raise ValueError(f"YAML root must be a mapping, got: {type(data)}")
    return data


# ---- summary.yaml (run-level summary of a path search) ------------------------------------------------------
try:  # Hartree -> kcal/mol exactly as pysisyphus.constants builds it: E_h * N_A / 1000 / 4.184
    from scipy import constants as _sc

    AU2KCALPERMOL = _sc.value("Hartree energy") * _sc.N_A / 1000.0 / 4.184
except Exception:  # CODATA 2022
    AU2KCALPERMOL = 627.5094740630558


# ---- energy series of a trajectory (the consumer of the .trj files: ``trj2fig.py:112-205,287-303``) ------------------------------
def recompute_energies(traj_path: PathLike, charge: Optional[int], multiplicity: Optional[int], *, calc=None,
                       max_batch: Optional[int] = None) -> List[float]:
    """Hartree energy of every frame of an XYZ trajectory, re-scored by the calculator (``trj2fig.py:112-134``: one
    ``uma_pysis(charge=charge or 0, spin=multiplicity or 1)``, ``get_energy(symbols, positions * ANG2BOHR)`` per frame).

    The reference walks the frames one by one; here ALL frames go to the engine in one batched call
    (``uma_pysis.get_energy_batch``; ``max_batch`` frames at a time if given) -- per frame the same number ``get_energy`` returns.
    ``calc``: an existing calculator to use (it is not closed); by default one is created as in the reference and closed again.
    Frames must hold the same atoms in the same order (the reference binds the calculator to the first frame's elements)."""
    symbols, coords_ang, _ = read_trj(traj_path)
    if len(coords_ang) == 0:
        raise RuntimeError(f"No frames found in {traj_path}")
    from ._calculator_base import ANG2BOHR

    own = calc is None
    if own:
        from .uma_pysis import uma_pysis

        calc = uma_pysis(charge=charge or 0, spin=multiplicity or 1)
    try:
        coords_bohr = np.asarray(coords_ang, dtype=np.float64) * ANG2BOHR
        step = len(coords_bohr) if not max_batch else max(1, int(max_batch))
        energies: List[float] = []
        for k0 in range(0, len(coords_bohr), step):
            energies += [float(e) for e in calc.get_energy_batch(symbols, coords_bohr[k0:k0 + step])["energy"]]
        return energies
    finally:
        if own and hasattr(calc, "close"):
            calc.close()


def _parse_reference_spec(spec: Optional[str]):
    """``-r/--reference``: None -> "init"; "none"/"null" -> None (absolute energies); "init"; else an integer frame index
    (``trj2fig.py:137-157``)."""
    if spec is None:
        return "init"
    s = str(spec).strip()
    if s.lower() in ("none", "null"):
        return None
    if s.lower() == "init":
        return "init"
    try:
        return int(s)
    except ValueError:
        raise ValueError(f"Invalid -r/--reference: {spec!r}. Use 'init', 'None', or an integer index.")


def transform_series(energies_hartree: Sequence[float], ref_spec_raw: Optional[str], unit: str,
                     reverse_x: bool) -> Tuple[List[float], str, bool]:
    """(values, y-axis label, is_delta) of an energy profile (``trj2fig.py:160-205``): relative to the reference frame ("init" = first
    frame, or the last one when the x axis is reversed; an explicit index must lie in 0..n-1 -> IndexError) or absolute when the
    reference is none; ``unit`` "kcal" scales by AU2KCALPERMOL, anything else stays in Hartree."""
    ref = _parse_reference_spec(ref_spec_raw)
    n = len(energies_hartree)
    if ref is None:
        idx = None
    elif ref == "init":
        idx = n - 1 if reverse_x else 0
    else:
        idx = int(ref)
        if idx < 0 or idx >= n:
            raise IndexError(f"Reference index {idx} out of range (0..{n-1}).")
    scale = AU2KCALPERMOL if unit == "kcal" else 1.0
    name = "kcal/mol" if unit == "kcal" else "hartree"
    if idx is None:
        return [float(e * scale) for e in energies_hartree], f"E ({name})", False
    base = energies_hartree[idx]
    return [float((e - base) * scale) for e in energies_hartree], f"\u0394E ({name})", True


def write_energy_csv(out: PathLike, energies_hartree: Sequence[float], series: Sequence[float], unit: str, is_delta: bool) -> None:
    """``frame,energy_hartree,<delta|energy>_<unit>`` with ``%.8f`` / ``%.6f`` values, csv-module line ends (``trj2fig.py:287-303``)."""
    import csv

    with Path(out).open("w", newline="", encoding="utf-8") as fh:
        w = csv.writer(fh)
        w.writerow(["frame", "energy_hartree", f"delta_{unit}" if is_delta else f"energy_{unit}"])
        for i, (eh, y) in enumerate(zip(energies_hartree, series)):
            w.writerow([i, f"{eh:.8f}", f"{y:.6f}"])


def barrier_and_delta_kcal(energies_hartree: Sequence[float]) -> Tuple[float, float]:
    """(max(E) - E[0], E[-1] - E[0]) in kcal/mol -- the two numbers reported per MEP segment (``path_search.py:1206-1207``)."""
    e = [float(x) for x in energies_hartree]
    return (max(e) - e[0]) * AU2KCALPERMOL, (e[-1] - e[0]) * AU2KCALPERMOL


def bond_changes_block(text: Optional[str]):
    """The ``bond_changes`` value of a summary segment.

    ``summarize_changes`` text such as ``"Bond formed (1):\n  - C1-O2 : 1.500 Å --> 1.360 Å\nBond broken: None"`` becomes
    ``[{"Bond formed (1)": ["C1-O2 : 1.500 Å --> 1.360 Å"]}, {"Bond broken": ["None"]}]``; empty / None -> ``""``; text
    without any ``Bond ...`` heading is kept as is (behaviour of the reference's ``_bond_changes_block``,
    ``path_search.py:245-293``; multi-line fallback text is dumped in YAML literal style by :func:`write_summary_yaml`)."""
    body = "" if text is None else str(text).strip()
    if not body:
        return ""
    out: List[Dict[str, List[str]]] = []
    open_title: Optional[str] = None
    items: List[str] = []

    def close():
        nonlocal open_title, items
        if open_title is not None:
            out.append({open_title: items or ["None"]})
        open_title, items = None, []

    for raw in body.splitlines():
        line = raw.strip()
        if line.startswith("Bond "):
            close()
            if line.endswith(": None"):
                out.append({line[: -len(": None")]: ["None"]})
            else:
                open_title = line.rstrip(":")
        elif line.startswith("- "):
            items.append(line[2:])
    close()
    return out if out else body


def summary_dict(out_dir: PathLike, n_images: int, segments: Sequence[Mapping[str, Any]],
                 energy_diagram: Optional[Mapping[str, Any]] = None) -> Dict[str, Any]:
    """Key order and types of the reference's summary (``path_search.py:2762-2782``).  Each segment mapping carries
    ``index, tag, kind ("seg" | "bridge"), barrier_kcal, delta_kcal`` and the ``summarize_changes`` text under ``summary``
    (bridges report no bond changes)."""
    segs = []
    for sg in segments:
        kind = str(sg.get("kind", "seg"))
        segs.append({"index": int(sg["index"]), "tag": sg["tag"], "kind": kind, "barrier_kcal": float(sg["barrier_kcal"]),
                     "delta_kcal": float(sg["delta_kcal"]), "bond_changes": bond_changes_block(sg.get("summary")) if kind != "bridge" else ""})
    d: Dict[str, Any] = {"out_dir": str(out_dir), "n_images": int(n_images), "n_segments": len(segs), "segments": segs}
    if energy_diagram is not None:
        d["energy_diagrams"] = [dict(energy_diagram)]
    return d


def write_summary_yaml(path: PathLike, summary: Mapping[str, Any]) -> str:
    """Dump `summary` as the reference does (block style, insertion order, unicode kept; multi-line strings literal)."""
    import yaml

    class _Dumper(yaml.SafeDumper):
        pass

    def _str(dumper, data):
        return dumper.represent_scalar("tag:yaml.org,2002:str", data, style="|" if "\n" in data else None)

    _Dumper.add_representer(str, _str)
    text = yaml.dump(dict(summary), Dumper=_Dumper, sort_keys=False, allow_unicode=True)
    Path(path).write_text(text, encoding="utf-8")
    return text
