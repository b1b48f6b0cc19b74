"""The GSM branch of the reference's ``path-opt`` flow as ONE function on plain arrays: the caller of the hot path.

Reference ``path_opt.cli`` (``path_opt.py:669-1091``), GSM mode, does, with one shared calculator (``:823``):
optional endpoint pre-optimisation (``_optimize_single``, ``:826-868``) -> rigid alignment + staged anchor scan of the second endpoint
onto the first (``align_and_refine_sequence_inplace``, ``:871-886``) -> ``GrowingString`` + ``StringOptimizer.run()`` (``:949-977``) ->
``final_geometries.trj`` with the energies on the comment lines (``:983-1004``) -> highest-energy image ``hei.xyz`` (``:1033-1048``).
:func:`optimize_path_gsm` strings together this repository's counterparts of exactly those steps -- ``rfo.optimize_single``,
``prestep.align_and_refine_sequence``, ``gsm.GrowingStringDriver.from_calculator`` (device resident on the engine, every cycle ONE batched
evaluation, images sharded over ranks when torch.distributed is initialised), ``formats.write_trj_with_energy`` / ``write_xyz``,
``string.select_hei_index`` -- so that a ``path-opt`` run is one call.  Click options, YAML merging, PDB / GJF conversion and exit codes are
the reference's CLI plumbing and stay out of scope; failures raise.

Keyword defaults are the CLI's: ``max_nodes = 10`` (``GS_KW``), ``fix_ends = False`` (``--fix-ends`` default, ``:663-668,735-736``),
``max_cycles = 300`` for both ``opt.max_cycles`` and ``stop_in_when_full`` (``:731-732``), ``climb = True``, ``preopt = False``.
"""
from __future__ import annotations

import os
from typing import Any, Dict, Optional, Sequence

import numpy as np

from ._calculator_base import ANG2BOHR, BOHR2ANG
from . import formats
from .gsm import GS_KW, STOPT_KW, GrowingStringDriver
from .string import select_hei_index


def optimize_path_gsm(elem: Sequence[str], reactant_ang: np.ndarray, product_ang: np.ndarray, *, calc=None, calc_kw: Optional[Dict[str, Any]] = None,
                      freeze_atoms: Sequence[int] = (), max_nodes: int = GS_KW["max_nodes"], max_cycles: int = 300, climb: bool = True,
                      fix_ends: bool = False, thresh: Optional[str] = None, preopt: bool = False, preopt_max_cycles: int = 10000,
                      sopt_kind: str = "lbfgs", sopt_cfg: Optional[Dict[str, Any]] = None, align: bool = True, align_kw: Optional[Dict[str, Any]] = None,
                      gs_kw: Optional[Dict[str, Any]] = None, stopt_kw: Optional[Dict[str, Any]] = None, out_dir: Optional[str] = None,
                      group=None, log=None) -> Dict[str, Any]:
    """Minimum-energy path between two endpoint geometries (Angstrom) by the growing-string method on the HIP engine.

    calc: a ``uma_pysis`` instance shared by every step (created from ``calc_kw`` + ``freeze_atoms`` when None, as ``path_opt.py:823``).
    Returns ``{"images_ang" (K, N, 3), "energies" (K,) Hartree, "hei_index", "converged", "cycles", "force_evaluations", "fully_grown",
    "preopt": [...], "align": [...], "files": {...}}``; with ``out_dir`` the reference's ``final_geometries.trj`` and ``hei.xyz`` are written."""
    log = log or (lambda s: None)
    elem = [str(e).capitalize() for e in elem]
    n = len(elem)
    freeze = sorted({int(i) for i in freeze_atoms})
    if calc is None:
        from .uma_pysis import uma_pysis

        calc = uma_pysis(**{**(calc_kw or {}), "freeze_atoms": freeze})
    geoms = [np.asarray(reactant_ang, dtype=np.float64).reshape(n, 3) * ANG2BOHR, np.asarray(product_ang, dtype=np.float64).reshape(n, 3) * ANG2BOHR]

    # optional endpoint pre-optimisation (path_opt.py:826-868): a failure keeps the input geometry, as the reference does
    pre = []
    if preopt:
        from .rfo import optimize_single

        cfg = {"max_cycles": int(preopt_max_cycles), **(sopt_cfg or {})}
        for i in range(2):
            try:
                r = optimize_single(calc, elem, geoms[i], sopt_kind, cfg, freeze=freeze)
                geoms[i] = np.asarray(r["coords"], dtype=np.float64).reshape(n, 3)
                pre.append({"converged": bool(r["converged"]), "cycles": int(r["cycles"]), "energy": float(r["energy"])})
                log(f"[preopt] endpoint {i}: {sopt_kind} {'converged' if r['converged'] else 'stopped'} after {r['cycles']} cycles")
            except Exception as exc:     # noqa: BLE001 -- reference: "[preopt] WARNING: Failed to preoptimize endpoint" and carry on
                pre.append({"error": f"{type(exc).__name__}: {exc}"})
                log(f"[preopt] WARNING: endpoint {i} not pre-optimised: {exc}")

    # rigid alignment + staged anchor scan of the product onto the reactant (path_opt.py:871-886); skipped with a note on failure
    align_info = []
    if align:
        from .prestep import align_and_refine_sequence

        try:
            geoms, align_info = align_and_refine_sequence(calc, elem, geoms, [freeze, freeze], **(align_kw or {}))
        except Exception as exc:         # noqa: BLE001 -- reference: "[align] WARNING: alignment skipped"
            align_info = [{"error": f"{type(exc).__name__}: {exc}"}]
            log(f"[align] WARNING: alignment skipped: {exc}")

    g = {"max_nodes": int(max_nodes), "climb": bool(climb), "climb_lanczos": bool(climb) and GS_KW["climb_lanczos"], "fix_first": bool(fix_ends),
         "fix_last": bool(fix_ends), **(gs_kw or {})}
    o = {"max_cycles": int(max_cycles), "stop_in_when_full": int(max_cycles), **({"thresh": thresh} if thresh else {}), **(stopt_kw or {})}
    drv = GrowingStringDriver.from_calculator(elem, geoms[0].reshape(-1), geoms[1].reshape(-1), calc, group=group, gs_kw=g, stopt_kw=o, log=log)
    res = drv.run()
    images_ang = res.coords.reshape(len(res.coords), n, 3) * BOHR2ANG
    hei = select_hei_index(res.energies)                         # path_opt.py:259-273
    files: Dict[str, str] = {}
    if out_dir is not None:
        os.makedirs(out_dir, exist_ok=True)
        files["final_geometries"] = os.path.join(out_dir, "final_geometries.trj")
        formats.write_trj_with_energy(elem, images_ang, res.energies, files["final_geometries"])
        files["hei"] = os.path.join(out_dir, "hei.xyz")
        formats.write_xyz(elem, images_ang[hei], files["hei"], energy_hartree=float(res.energies[hei]))
    return {"images_ang": images_ang, "energies": res.energies, "hei_index": int(hei), "converged": bool(res.converged), "cycles": int(res.cycles),
            "force_evaluations": int(res.force_evaluations), "fully_grown": bool(res.fully_grown), "history": res.history, "timing": res.timing,
            "device": str(drv.device), "preopt": pre, "align": align_info, "files": files, "defaults": {"GS_KW": dict(GS_KW), "STOPT_KW": dict(STOPT_KW)}}
