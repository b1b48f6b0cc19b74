#!/usr/bin/env python3
"""Generate tests/golden/ref_host_functions.json: input -> output vectors of the REFERENCE's own pure helper functions.

BUILD-CONTAINER ONLY (needs /root/reference; the GPU box never has it and never runs this).  The reference package cannot
be imported here (pysisyphus / ase / fairchem are absent), but a handful of its host-side helpers are self-contained
numpy / stdlib code.  This script parses the reference files with ``ast``, compiles ONLY the named function definitions
(nothing else of the module is executed, no module of the reference is imported, no third-party stand-ins are written)
and records what they return on seeded inputs.  The fixture is DATA (inputs and outputs); no reference source text is
stored.  tests/test_reference_fixtures.py then holds this repo's restatements to those outputs:

  pdb2reaction/path_opt.py            _select_hei_index                         -> pdb2reaction_amd.string.select_hei_index
  pdb2reaction/align_freeze_atoms.py  kabsch_R_t, _rodrigues,
                                      _rotation_align_vectors, _rmsd            -> pdb2reaction_amd.prestep.*
  pdb2reaction/opt.py                 HarmonicBiasCalculator._bias_energy_forces_bohr -> prestep.HarmonicBias._bias
  pdb2reaction/bond_changes.py        _bond_str, summarize_changes              -> pdb2reaction_amd.bond_changes.*
  pdb2reaction/trj2fig.py             read_energies_xyz                         -> pdb2reaction_amd.formats.read_energies_xyz
  pdb2reaction/path_search.py         _bond_changes_block (+ the summary.yaml
                                      dump call as written at :2784-2785)       -> pdb2reaction_amd.formats.bond_changes_block,
                                                                                   summary_dict, write_summary_yaml
  pdb2reaction/utils.py               deep_update, apply_yaml_overrides,
                                      load_yaml_dict                            -> pdb2reaction_amd.formats.*

and, into tests/golden/ref_uma_pysis_methods.json (round 3: the file that IS the drop-in boundary):

  pdb2reaction/uma_pysis.py           uma_pysis._au_energy, _au_forces, _au_hessian, _active_and_frozen_dof_idx,
                                      _zero_frozen_forces_ev, _apply_analytical_active_trim, _build_fd_hessian_gpu,
                                      get_energy, get_forces, get_hessian (:502-780), run on tests/toy_core.ToyPairCore
                                                                                -> pdb2reaction_amd.uma_pysis.uma_pysis, hessian.*
  pdb2reaction/bond_changes.py        compare_structures (:142-187; CR injected)-> pdb2reaction_amd.bond_changes.compare_structures
                                                                                   (HIP kernel k_bond_changes), oracle/bond_changes_oracle
  pdb2reaction/path_opt.py            _write_ase_trj_with_energy (:276-290)     -> pdb2reaction_amd.formats.write_trj_with_energy

and, into tests/golden/ref_energy_series.json (the consumer of the .trj files):

  pdb2reaction/trj2fig.py             recompute_energies, transform_series (+ _parse_reference_spec,
                                      _resolve_reference_index), write_csv (:112-205,287-303)
                                                                                -> pdb2reaction_amd.formats.recompute_energies (ONE batched
                                                                                   engine call for all frames), transform_series, write_energy_csv

Unit constants the reference takes from pysisyphus.constants are passed in from scipy and recorded
in the fixture.
"""
from __future__ import annotations

import ast
import copy
import json
import os
import re
import sys
import tempfile
import types
from dataclasses import dataclass
from pathlib import Path
from typing import Any, Dict, Iterable, List, Mapping, Optional, Sequence, Set, Tuple

import numpy as np

REF = Path(os.environ.get("REFERENCE_ROOT", "/root/reference")) / "pdb2reaction"
ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "tests" / "golden" / "ref_host_functions.json"
sys.path.insert(0, str(ROOT))
from pdb2reaction_amd._calculator_base import ANG2BOHR, AU2EV, BOHR2ANG  # noqa: E402  (scipy CODATA, as pysisyphus)

H_EVAA_2_AU = 1.0 / AU2EV / ANG2BOHR / ANG2BOHR


def grab(path: Path, names: Sequence[str], cls: Optional[str] = None, extra: Optional[Dict[str, Any]] = None) -> Dict[str, Any]:
    """Compile the named module-level functions / dataclasses (or methods of class `cls`) of one reference file."""
    src = path.read_text()
    tree = ast.parse(src)
    body = tree.body
    if cls is not None:
        body = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls).body
    ns: Dict[str, Any] = {"np": np, "re": re, "Any": Any, "Dict": Dict, "List": List, "Tuple": Tuple, "Optional": Optional,
                          "Sequence": Sequence, "_Sequence": Sequence, "Mapping": Mapping, "Set": Set, "Iterable": Iterable,
                          "Path": Path, "dataclass": dataclass, "ANG2BOHR": ANG2BOHR, "BOHR2ANG": BOHR2ANG,
                          "H_EVAA_2_AU": H_EVAA_2_AU}
    ns.update(extra or {})
    for node in body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, f"<{path.name}:{node.name}>", "exec"), ns)
    missing = [n for n in names if n not in ns]
    if missing:
        raise RuntimeError(f"{path}: {missing} not found")
    return ns


def main():
    rng = np.random.default_rng(20261004)
    fx: Dict[str, Any] = {"constants": {"ANG2BOHR": ANG2BOHR, "BOHR2ANG": BOHR2ANG, "H_EVAA_2_AU": H_EVAA_2_AU},
                          "generated_by": "tools/make_reference_fixtures.py", "reference": "t-0hmura/pdb2reaction (local checkout)"}

    # ---- HEI rule (path_opt.py:259-273)
    hei = grab(REF / "path_opt.py", ["_select_hei_index"])["_select_hei_index"]
    cases = [[0.0, 1.0], [3.0], [0.0, 2.0, 1.0], [0.0, 1.0, 3.0, 2.0, 5.0, 4.0, 0.5], [5.0, 1.0, 2.0, 3.0, 9.0],
             [9.0, 1.0, 0.5, 0.2, 0.1], [1.0, 1.0, 1.0, 1.0], [0.0, 2.0, 2.0, 1.0], [0.0, 3.0, 1.0, 3.0, 0.0]]
    cases += [list(rng.normal(size=int(n))) for n in rng.integers(2, 30, size=40)]
    fx["select_hei_index"] = [{"energies": c, "index": int(hei(c))} for c in cases]

    # ---- alignment maths (align_freeze_atoms.py:128-225)
    al = grab(REF / "align_freeze_atoms.py", ["kabsch_R_t", "_rodrigues", "_rotation_align_vectors", "_orth_proj_perp", "_rmsd"])
    kab = []
    for n in (3, 4, 7, 25):
        P = rng.normal(size=(n, 3)) * 3.0
        Q = rng.normal(size=(n, 3)) * 3.0
        if n == 7:                                # mobile = rotated + reflected copy: exercises the det < 0 branch
            Q = P @ np.diag([1.0, 1.0, -1.0]) + 0.01 * rng.normal(size=(n, 3))
        R, t = al["kabsch_R_t"](P.copy(), Q.copy())
        kab.append({"P": P.tolist(), "Q": Q.tolist(), "R": R.tolist(), "t": t.tolist()})
    fx["kabsch_R_t"] = kab
    rod = []
    for _ in range(6):
        ax, th = rng.normal(size=3), float(rng.uniform(-np.pi, np.pi))
        rod.append({"axis": ax.tolist(), "theta": th, "R": al["_rodrigues"](ax.copy(), th).tolist()})
    rod.append({"axis": [0.0, 0.0, 0.0], "theta": 1.0, "R": al["_rodrigues"](np.zeros(3), 1.0).tolist()})
    fx["rodrigues"] = rod
    rav = []
    pairs = [(rng.normal(size=3), rng.normal(size=3)) for _ in range(6)]
    pairs += [(np.array([1.0, 2.0, 3.0]), np.array([2.0, 4.0, 6.0])), (np.array([1.0, 0.0, 0.0]), np.array([-2.0, 0.0, 0.0])),
              (np.array([0.0, 1.0, 0.0]), np.array([0.0, -1.0, 0.0])), (np.zeros(3), np.array([1.0, 0.0, 0.0]))]
    for a, b in pairs:
        rav.append({"a": a.tolist(), "b": b.tolist(), "R": al["_rotation_align_vectors"](a.copy(), b.copy()).tolist()})
    fx["rotation_align_vectors"] = rav
    fx["rmsd"] = []
    for n in (1, 5, 40):
        A, B = rng.normal(size=(n, 3)), rng.normal(size=(n, 3))
        fx["rmsd"].append({"A": A.tolist(), "B": B.tolist(), "rmsd_ang": al["_rmsd"](A, B)})

    # ---- harmonic bias (opt.py:298-322): the method compiled on its own, `self` = a plain namespace with the two attributes it reads
    bias_fn = grab(REF / "opt.py", ["_bias_energy_forces_bohr"], cls="HarmonicBiasCalculator")["_bias_energy_forces_bohr"]
    fx["harmonic_bias"] = []
    for n, k_ev in ((4, 10.0), (9, 3.5), (3, 100.0)):
        x = rng.normal(size=(n, 3)) * 2.5
        prs = [(int(i), int(j), float(t)) for i, j, t in zip(rng.integers(0, n, 6), rng.integers(0, n, 6), rng.uniform(0.8, 3.0, 6))]
        prs += [(0, n + 3, 1.0), (-1, 0, 1.0), (1, 1, 1.5)]                 # out of range / self pair: skipped by the reference
        me = types.SimpleNamespace(k_au_bohr2=float(k_ev) * H_EVAA_2_AU, _pairs=list(prs))
        e, f = bias_fn(me, x.reshape(-1).copy())
        fx["harmonic_bias"].append({"k_ev_ang2": k_ev, "coords_bohr": x.tolist(), "pairs": [list(p) for p in prs],
                                    "energy": float(e), "forces": np.asarray(f).tolist()})

    # ---- bond-change report text (bond_changes.py:96-232)
    bc = grab(REF / "bond_changes.py", ["BondChangeResult", "_bond_str", "summarize_changes"], extra={"Pair": Tuple[int, int]})
    fx["summarize_changes"] = []
    for formed, broken, with_d in (({(0, 2), (1, 3)}, {(2, 4)}, True), (set(), {(0, 1)}, True), (set(), set(), True), ({(3, 4)}, set(), False)):
        n = 5
        d1, d2 = np.abs(rng.normal(size=(n, n))) * 4 + 1, np.abs(rng.normal(size=(n, n))) * 4 + 1
        res = bc["BondChangeResult"](formed_covalent=formed, broken_covalent=broken, distances_1=d1 if with_d else None,
                                     distances_2=d2 if with_d else None)
        geom = types.SimpleNamespace(atoms=["c", "H", "o", "N", "cl"])
        for one_based in (True, False):
            fx["summarize_changes"].append({"atoms": geom.atoms, "formed": sorted(map(list, formed)), "broken": sorted(map(list, broken)),
                                            "d1": d1.tolist() if with_d else None, "d2": d2.tolist() if with_d else None,
                                            "one_based": one_based, "text": bc["summarize_changes"](geom, res, one_based)})

    # ---- energy reader (trj2fig.py:86-109)
    rd = grab(REF / "trj2fig.py", ["read_energies_xyz"])["read_energies_xyz"]
    fx["read_energies_xyz"] = []
    texts = ["2\n-1.5\nH 0 0 0\nH 0 0 1\n2\n   3.25 extra 7\nH 0 0 0\nH 0 0 1\n",
             "1\nE = -1.25e-3 Hartree step 7\nH 0 0 0\n1\nenergy: 42\nH 0 0 1\n",
             "1\n-0.000000000001\nH 0 0 0\nnot a header\n1\n5.0\nH 0 0 0\n",
             "1\nno number here\nH 0 0 0\n", "not an xyz\n", "", "1\n.5\nH 0 0 0\n", "1\n-228.123456789012\nC 1 2 3"]
    with tempfile.TemporaryDirectory() as td:
        for t in texts:
            p = Path(td) / "a.trj"
            p.write_text(t)
            try:
                fx["read_energies_xyz"].append({"text": t, "energies": rd(p)})
            except Exception as exc:
                fx["read_energies_xyz"].append({"text": t, "raises": type(exc).__name__, "message": str(exc).replace(str(p), "<path>")})

        # ---- YAML precedence (utils.py:243-313)
        import yaml
        ut = grab(REF / "utils.py", ["deep_update", "_get_mapping_section", "apply_yaml_overrides", "load_yaml_dict"], extra={"yaml": yaml})
        fx["deep_update"] = []
        for dst, src in (({"a": 1, "n": {"x": 1, "y": {"z": 2}}}, {"n": {"y": {"w": 3}, "x": [1, 2]}, "b": None}),
                         ({"a": {"b": 1}}, {"a": 5}), ({"a": 5}, {"a": {"b": 1}}), ({"k": {"l": [1]}}, None), ({}, {"q": {"r": {}}})):
            d = copy.deepcopy(dst)
            fx["deep_update"].append({"dst": dst, "src": src, "result": ut["deep_update"](d, copy.deepcopy(src))})
        fx["apply_yaml_overrides"] = []
        ycfg = {"calc": {"charge": 2, "spin": 3}, "gs": {"max_nodes": 14, "nested": {"b": 5}}, "sopt": {"lbfgs": {"thresh": "gau"}},
                "lbfgs": {"thresh": "baker"}, "opt": None, "scalar": 7, "deep": {"x": {"y": 4}}}
        targets = [({"charge": -1, "model": "uma-s-1p1"}, [["calc"]]), ({"max_nodes": 10, "nested": {"a": 1, "b": 2}}, [["gs"]]),
                   ({"thresh": "gau_loose", "max_cycles": 100}, [["sopt", "lbfgs"], ["lbfgs"]]), ({"thresh": "x"}, [["missing"], ["lbfgs"]]),
                   ({"a": 1}, [["opt"], ["scalar"], ["deep", "x"]]), ({"keep": True}, [["nope"], ["scalar", "deeper"]])]
        tg = [(copy.deepcopy(t), tuple(tuple(p) for p in ps)) for t, ps in targets]
        ut["apply_yaml_overrides"](ycfg, tg)
        fx["apply_yaml_overrides"].append({"yaml": ycfg, "targets": [{"before": t, "paths": ps, "after": a[0]} for (t, ps), a in zip(targets, tg)]})
        fx["load_yaml_dict"] = []
        for t in ("calc:\n  charge: 2\n", "", "- 1\n- 2\n", "a: {b: [1, 2], c: null}\n", "42\n"):
            p = Path(td) / "c.yaml"
            p.write_text(t)
            try:
                fx["load_yaml_dict"].append({"text": t, "data": ut["load_yaml_dict"](p)})
            except Exception as exc:
                fx["load_yaml_dict"].append({"text": t, "raises": type(exc).__name__, "message": str(exc)})
        fx["load_yaml_dict"].append({"text": None, "data": ut["load_yaml_dict"](None)})

    # ---- summary.yaml (path_search.py:234-293, 2762-2786): the block builder is the reference's; the dict literal and the dump
    #      call mirror :2762-2785 (safe_dump, sort_keys=False, allow_unicode=True)
    import yaml as _yaml
    ps = grab(REF / "path_search.py", ["_LiteralStr", "_literal_str_representer", "_bond_changes_block"], extra={"yaml": _yaml})
    _yaml.add_representer(ps["_LiteralStr"], ps["_literal_str_representer"], Dumper=_yaml.SafeDumper)
    texts_bc = ["Bond formed (1):\n  - C1-O2 : 1.500 Å --> 1.360 Å\nBond broken: None", "Bond formed: None\nBond broken (2):\n  - H3-O4 : 0.980 Å --> 2.310 Å\n  - C1-H9 : 1.090 Å --> 1.900 Å",
                "", None, "   ", "(no covalent changes detected)", "two lines\nof free text", "Bond formed (1):", "Bond formed: None\nBond broken: None"]
    fx["bond_changes_block"] = []
    for t in texts_bc:
        r = ps["_bond_changes_block"](t)
        fx["bond_changes_block"].append({"text": t, "result": r if not isinstance(r, str) else str(r), "literal": isinstance(r, ps["_LiteralStr"])})
    segs_in = [{"index": 1, "tag": "seg_000", "kind": "seg", "barrier_kcal": 23.456789012345, "delta_kcal": -5.5, "summary": texts_bc[0]},
               {"index": 2, "tag": "bridge_000_001", "kind": "bridge", "barrier_kcal": float("nan"), "delta_kcal": 1e-7, "summary": ""},
               {"index": 3, "tag": "seg_001", "kind": "seg", "barrier_kcal": 0.0, "delta_kcal": 12.0, "summary": "two lines\nof free text"},
               {"index": 4, "tag": "seg_002", "kind": "seg", "barrier_kcal": 3.25, "delta_kcal": -0.125, "summary": "(no covalent changes detected)"}]
    summary = {"out_dir": "result_path_search", "n_images": 37, "n_segments": len(segs_in),
               "segments": [{"index": int(sg["index"]), "tag": sg["tag"], "kind": sg["kind"], "barrier_kcal": float(sg["barrier_kcal"]),
                             "delta_kcal": float(sg["delta_kcal"]),
                             "bond_changes": (ps["_bond_changes_block"](sg["summary"]) if (sg["kind"] != "bridge") else "")} for sg in segs_in]}
    diagram = {"name": "energy_diagram_MEP", "labels": ["R", "TS1", "IM1", "P"], "energies_kcal": [0.0, 23.5, -2.0, -5.5], "ylabel": "ΔE (kcal/mol)"}
    fx["summary_yaml"] = []
    for dg in (None, diagram):
        sm = dict(summary)
        if dg is not None:
            sm["energy_diagrams"] = [dg]
        fx["summary_yaml"].append({"out_dir": "result_path_search", "n_images": 37, "segments": segs_in, "energy_diagram": dg,
                                   "text": _yaml.safe_dump(sm, sort_keys=False, allow_unicode=True)})

    # ---- rigid pre-alignment of a pair (align_freeze_atoms.py:253-387): 0 / 1 / 2 / degenerate-2 / many anchors.  The geometry objects
    # are data holders with the three members the function touches (coords3d, freeze_atoms, set_coords)
    class _Geom:
        def __init__(self, c, fz):
            self.coords3d = np.array(c, float)
            self.freeze_atoms = np.array(fz, int)
            self.seen_freeze = None

        def set_coords(self, flat, cartesian=True):
            assert cartesian is True
            self.seen_freeze = list(self.freeze_atoms)                 # must be empty while the coordinates are written
            self.coords3d = np.array(flat, float).reshape(-1, 3)

    import io
    import contextlib
    al2 = grab(REF / "align_freeze_atoms.py", ["kabsch_R_t", "_rodrigues", "_rotation_align_vectors", "_orth_proj_perp", "_rmsd", "_coords3d",
                                                "_set_all_coords_disabling_freeze", "_freeze_union", "align_second_to_first_kabsch_inplace"])
    rng2 = np.random.default_rng(20261007)
    fx["align_pair"] = []
    for n, fz_ref, fz_mob in ((9, [], []), (9, [4], []), (9, [], [4, 4]), (9, [2], [7]), (9, [7, 2], [2]), (9, [1, 3], [5, 8, 1]),
                              (6, [0, 1, 2, 3, 4, 5], []), (7, [3, 40, -2], []), (7, [50], [1, 6]), (5, [0, 4], [])):
        P = rng2.normal(size=(n, 3)) * 4.0
        ang = rng2.normal(size=3)
        rot = al2["_rodrigues"](ang / np.linalg.norm(ang), float(rng2.uniform(0.3, 2.5)))
        Q = P @ rot.T + rng2.normal(size=3) * 2.0 + rng2.normal(size=(n, 3)) * 0.15
        degenerate = (n == 5)
        if degenerate:                                                # both anchors on one point in the reference: falls through to Kabsch
            P[4] = P[0]
        g_ref, g_mob = _Geom(P, fz_ref), _Geom(Q, fz_mob)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            rep = al2["align_second_to_first_kabsch_inplace"](g_ref, g_mob, verbose=True)
        fx["align_pair"].append({"ref_bohr": P.tolist(), "mob_bohr": Q.tolist(), "freeze_ref": fz_ref, "freeze_mob": fz_mob,
                                 "union": al2["_freeze_union"](g_ref, g_mob, n_atoms=n), "union_unbounded": al2["_freeze_union"](g_ref, g_mob),
                                 "aligned_bohr": g_mob.coords3d.tolist(), "report": rep, "freeze_during_write": g_mob.seen_freeze,
                                 "freeze_after": [int(i) for i in g_mob.freeze_atoms], "printed": buf.getvalue()})
    try:
        al2["align_second_to_first_kabsch_inplace"](_Geom(np.zeros((3, 3)), []), _Geom(np.zeros((4, 3)), []), verbose=False)
    except Exception as exc:
        fx["align_pair_mismatch"] = {"raises": type(exc).__name__, "message": str(exc)}

    # ---- default settings (uma_pysis.py:132-165,432-452; path_opt.py:168-200; opt.py:171-246): the VALUES of the module-level
    # dict assignments (only those Assign nodes are evaluated, `**OPT_BASE_KW` resolving to the one just evaluated) and the
    # keyword defaults of the calculator's constructor
    def literal_dicts(path, names, given=None):
        env: Dict[str, Any] = dict(given or {})
        for node in ast.parse(path.read_text()).body:
            tgt = node.targets[0] if isinstance(node, ast.Assign) else node.target if isinstance(node, ast.AnnAssign) else None
            if isinstance(tgt, ast.Name) and tgt.id in names and getattr(node, "value", None) is not None:
                env[tgt.id] = eval(compile(ast.Expression(node.value), f"<{path.name}:{tgt.id}>", "eval"), {"__builtins__": {}, "dict": dict}, env)
        return {k: env[k] for k in names}

    dflt = literal_dicts(REF / "uma_pysis.py", ["GEOM_KW_DEFAULT", "CALC_KW"])
    dflt.update(literal_dicts(REF / "path_opt.py", ["GS_KW", "STOPT_KW"]))
    dflt.update(literal_dicts(REF / "opt.py", ["OPT_BASE_KW", "LBFGS_KW", "RFO_KW"]))
    cls = next(n for n in ast.parse((REF / "uma_pysis.py").read_text()).body if isinstance(n, ast.ClassDef) and n.name == "uma_pysis")
    init = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "__init__")
    dflt["uma_pysis.__init__"] = {"positional": [a.arg for a in init.args.args], "var_keyword": init.args.kwarg.arg if init.args.kwarg else None,
                                  "keyword_only": {a.arg: eval(compile(ast.Expression(d), "<uma_pysis.__init__>", "eval"), {"__builtins__": {}}, {"CALC_KW": dflt["CALC_KW"]})
                                                   for a, d in zip(init.args.kwonlyargs, init.args.kw_defaults)}}
    fx["defaults"] = dflt

    OUT.parent.mkdir(parents=True, exist_ok=True)
    OUT.write_text(json.dumps(fx, indent=1) + "\n")          # insertion order matters (YAML key order of dict-valued inputs)
    print(f"wrote {OUT} ({OUT.stat().st_size} bytes): " + ", ".join(f"{k}={len(v)}" for k, v in fx.items() if isinstance(v, list)))
    boundary_fixtures()
    energy_series_fixtures()

OUT_BOUNDARY = ROOT / "tests" / "golden" / "ref_uma_pysis_methods.json"
OUT_SERIES = ROOT / "tests" / "golden" / "ref_energy_series.json"
UMA_METHODS = ["_ensure_core", "_au_energy", "_au_forces", "_au_hessian", "_active_and_frozen_dof_idx", "_zero_frozen_forces_ev",
               "_apply_analytical_active_trim", "_build_fd_hessian_gpu", "get_energy", "get_forces", "get_hessian"]


def _hess_record(h) -> Dict[str, Any]:
    import torch

    if isinstance(h, torch.Tensor):
        return {"container": "torch", "dtype": str(h.dtype).replace("torch.", ""), "shape": list(h.shape),
                "values": h.detach().cpu().to(torch.float64).numpy().tolist()}
    return {"container": "numpy", "dtype": str(h.dtype), "shape": list(h.shape), "values": np.asarray(h, dtype=np.float64).tolist()}


def boundary_fixtures():
    """uma_pysis.py:502-780 (the calculator boundary itself), bond_changes.compare_structures, the ASE .trj writer."""
    import torch

    sys.path.insert(0, str(ROOT / "tests"))
    from toy_core import ToyPairCore, toy_geometry  # the committed toy analytic core both sides run on

    EV2AU = 1.0 / AU2EV
    F_EVAA_2_AU = EV2AU / ANG2BOHR
    ns = grab(REF / "uma_pysis.py", UMA_METHODS, cls="uma_pysis",
              extra={"torch": torch, "EV2AU": EV2AU, "F_EVAA_2_AU": F_EVAA_2_AU, "H_EVAA_2_AU": H_EVAA_2_AU})
    # the reference class body without its constructor: the methods exactly as compiled from the reference file; the instance
    # attributes are set the way uma_pysis.__init__ sets them (:482-499), `_core` is the toy core (so _ensure_core never loads UMA)
    RefCalc = type("RefCalc", (), {m: ns[m] for m in UMA_METHODS})

    def ref_calc(core, *, freeze_atoms=None, out_hess_torch=True, hessian_calc_mode="FiniteDifference",
                 return_partial_hessian=False, hessian_double=True):
        me = RefCalc()
        me._core = core
        me.out_hess_torch = out_hess_torch
        me.hessian_calc_mode = hessian_calc_mode
        me.freeze_atoms = sorted(set(int(i) for i in (freeze_atoms or [])))          # reference :497
        me.return_partial_hessian = bool(return_partial_hessian)
        me.hessian_double = bool(hessian_double)
        return me

    rng = np.random.default_rng(20261005)
    fx: Dict[str, Any] = {"generated_by": "tools/make_reference_fixtures.py:boundary_fixtures",
                          "reference": "t-0hmura/pdb2reaction (local checkout): pdb2reaction/uma_pysis.py:502-780, bond_changes.py:142-187, path_opt.py:276-290",
                          "core": "tests/toy_core.ToyPairCore (float32 positions and forces, scalar IEEE arithmetic)",
                          "constants": {"ANG2BOHR": ANG2BOHR, "BOHR2ANG": BOHR2ANG, "EV2AU": EV2AU, "F_EVAA_2_AU": F_EVAA_2_AU,
                                        "H_EVAA_2_AU": H_EVAA_2_AU}}

    # ---- helpers on their own
    me = ref_calc(None, freeze_atoms=[4, 1, 4])
    fx["au_energy"] = [{"e_ev": e, "e_au": me._au_energy(e)} for e in (-1234.56789, 0.0, 3.25e-7)]
    fx["au_forces"] = []
    for dt in (np.float32, np.float64):
        f = (rng.normal(size=(5, 3)) * 2.0).astype(dt)
        r = me._au_forces(f)
        fx["au_forces"].append({"dtype": np.dtype(dt).name, "f_ev": f.astype(np.float64).tolist(), "f_au": r.tolist(),
                                "out_dtype": str(r.dtype), "out_shape": list(r.shape)})
    fx["dof_idx"] = []
    for n, fz in ((6, [4, 1, 4]), (3, []), (4, [0, 1, 2, 3]), (5, [2])):
        m2 = ref_calc(None, freeze_atoms=fz)
        a, ad, fd = m2._active_and_frozen_dof_idx(n)
        fx["dof_idx"].append({"n_atoms": n, "freeze_atoms": fz, "normalised": m2.freeze_atoms, "active_atoms": a, "active_dof": ad, "frozen_dof": fd})
    fx["zero_frozen"] = []
    for fz in ([], [3, 0]):
        m2 = ref_calc(None, freeze_atoms=fz)
        f = rng.normal(size=(5, 3)).astype(np.float32)
        out = m2._zero_frozen_forces_ev(f)
        fx["zero_frozen"].append({"freeze_atoms": fz, "f": f.astype(np.float64).tolist(), "out": out.astype(np.float64).tolist(),
                                  "same_object": out is f, "none_passthrough": m2._zero_frozen_forces_ev(None) is None})
    fx["au_hessian"] = []
    for dt, dbl, as_torch in ((torch.float32, True, True), (torch.float32, False, True), (torch.float64, True, False), (torch.float32, False, False)):
        h = torch.as_tensor(rng.normal(size=(4, 3, 4, 3)), dtype=dt)
        m2 = ref_calc(None, hessian_double=dbl, out_hess_torch=as_torch)
        fx["au_hessian"].append({"in_dtype": str(dt).replace("torch.", ""), "hessian_double": dbl, "out_hess_torch": as_torch,
                                 "h": h.to(torch.float64).numpy().tolist(), "out": _hess_record(m2._au_hessian(h.clone()))})
    fx["analytical_trim"] = []
    for fz, part in (([], False), ([], True), ([2, 0], False), ([2, 0], True), ([1], False)):
        h = torch.as_tensor(rng.normal(size=(4, 3, 4, 3)), dtype=torch.float32)
        m2 = ref_calc(None, freeze_atoms=fz, return_partial_hessian=part)
        fx["analytical_trim"].append({"freeze_atoms": fz, "return_partial_hessian": part, "h": h.to(torch.float64).numpy().tolist(),
                                      "out": _hess_record(m2._apply_analytical_active_trim(h.clone()))})

    # ---- the three API methods end to end on the toy core
    fx["api"] = []
    n = 6
    elem = ["c", "H", "o", "N", "h", "S"]
    variants = []
    for fz in ([], [4, 1, 4]):
        for part in (False, True):
            for dbl in (True, False):
                for as_torch in (True, False):
                    variants.append(dict(freeze_atoms=fz, return_partial_hessian=part, hessian_double=dbl, out_hess_torch=as_torch,
                                         hessian_calc_mode="FiniteDifference", workers=1, has_torch_model=False))
    for mode in (None, "", "bogus", " finitedifference "):
        variants.append(dict(freeze_atoms=[2], return_partial_hessian=False, hessian_double=True, out_hess_torch=False,
                             hessian_calc_mode=mode, workers=1, has_torch_model=True))
    for mode, workers, has_model in (("Analytical", 1, True), (" analytic ", 1, True), ("Analytical", 4, True), ("Analytical", 1, False)):
        for fz, part in (([], False), ([3, 0], False), ([3, 0], True)):
            variants.append(dict(freeze_atoms=fz, return_partial_hessian=part, hessian_double=True, out_hess_torch=False,
                                 hessian_calc_mode=mode, workers=workers, has_torch_model=has_model))
    for vi, v in enumerate(variants):
        x_ang = toy_geometry(n, seed=vi % 3)
        flat = (vi % 2 == 0)
        coords_bohr = (x_ang * ANG2BOHR).reshape(-1) if flat else (x_ang * ANG2BOHR)
        core = ToyPairCore(n, seed=7, parallel_predict=v["workers"] > 1, has_torch_model=v["has_torch_model"])
        kw = {k: v[k] for k in ("freeze_atoms", "return_partial_hessian", "hessian_double", "out_hess_torch", "hessian_calc_mode")}
        me = ref_calc(core, **kw)
        r_e = me.get_energy(elem, coords_bohr.tolist())
        r_f = me.get_forces(elem, coords_bohr)
        calls0 = core.calls
        r_h = me.get_hessian(elem, coords_bohr)
        fx["api"].append({**v, "n_atoms": n, "elem": elem, "core_seed": 7, "geometry_seed": vi % 3, "coords_flat": flat,
                          "coords_bohr": np.asarray(coords_bohr).tolist(),
                          "get_energy": {"energy": r_e["energy"], "keys": sorted(r_e)},
                          "get_forces": {"energy": r_f["energy"], "forces": r_f["forces"].tolist(), "dtype": str(r_f["forces"].dtype),
                                         "shape": list(r_f["forces"].shape), "keys": sorted(r_f)},
                          "get_hessian": {"energy": r_h["energy"], "forces": r_h["forces"].tolist(), "hessian": _hess_record(r_h["hessian"]),
                                          "keys": sorted(r_h), "reference_core_calls": core.calls - calls0}})

    # ---- compare_structures (bond_changes.py:142-187); CR = injected radii in the unit of the coordinates
    radii = {"h": 0.6, "c": 1.45, "n": 1.35, "o": 1.25, "s": 2.0}
    bc = grab(REF / "bond_changes.py", ["BondChangeResult", "_upper_pairs_from_mask", "_element_arrays", "_resolve_device", "compare_structures"],
              extra={"Pair": Tuple[int, int], "torch": torch, "CR": radii, "warnings": __import__("warnings")})
    fx["compare_structures"] = []
    for n_at, kw in ((6, {}), (12, {"bond_factor": 1.3, "margin_fraction": 0.02, "delta_fraction": 0.1}), (40, {}), (2, {}), (1, {})):
        atoms = [str(a) for a in rng.choice(["H", "c", "N", "o", "S"], size=n_at)]
        g = max(2, int(np.ceil(n_at ** (1.0 / 3.0))))
        grid = np.array([(i, j, k) for i in range(g) for j in range(g) for k in range(g)], dtype=np.float64)[:n_at] * 2.6
        r1 = grid + rng.uniform(-0.45, 0.45, size=(n_at, 3))
        r2 = r1 + rng.normal(scale=0.45, size=(n_at, 3))
        g1, g2 = types.SimpleNamespace(atoms=atoms, coords3d=r1), types.SimpleNamespace(atoms=list(atoms), coords3d=r2)
        res = bc["compare_structures"](g1, g2, device="cpu", **kw)
        fx["compare_structures"].append({"atoms": atoms, "radii": radii, "r1": r1.tolist(), "r2": r2.tolist(), "kwargs": kw,
                                         "formed": sorted(map(list, ((int(i), int(j)) for i, j in res.formed_covalent))),
                                         "broken": sorted(map(list, ((int(i), int(j)) for i, j in res.broken_covalent))),
                                         "d1": res.distances_1.tolist(), "d2": res.distances_2.tolist()})
    try:
        bc["compare_structures"](types.SimpleNamespace(atoms=["H", "C"], coords3d=np.zeros((2, 3))),
                                 types.SimpleNamespace(atoms=["C", "H"], coords3d=np.zeros((2, 3))), device="cpu")
        fx["compare_structures_mismatch"] = None
    except AssertionError as exc:
        fx["compare_structures_mismatch"] = {"raises": "AssertionError", "message": str(exc)}

    # ---- ASE .trj writer of the DMF path (path_opt.py:276-290); `images` only need the two ASE accessors it calls
    wr = grab(REF / "path_opt.py", ["_write_ase_trj_with_energy"])["_write_ase_trj_with_energy"]

    class _Img:
        def __init__(self, sym, pos):
            self._s, self._p = list(sym), np.asarray(pos, dtype=np.float64)

        def get_chemical_symbols(self):
            return list(self._s)

        def get_positions(self):
            return self._p.copy()

    fx["write_ase_trj"] = []
    with tempfile.TemporaryDirectory() as td:
        for n_at, k in ((3, 2), (5, 4), (1, 1)):
            sym = [str(a) for a in rng.choice(["H", "C", "N", "O", "Cl"], size=n_at)]
            imgs = [rng.normal(size=(n_at, 3)) * 4.0 for _ in range(k)]
            imgs[0][0] = [0.0, -0.0, 1e-17]
            en = list((rng.normal(size=k) * 100.0).tolist())
            en[0] = -228.1234567890123456
            p = Path(td) / "dmf.trj"
            wr([_Img(sym, x) for x in imgs], en, p)
            fx["write_ase_trj"].append({"symbols": sym, "images_ang": [x.tolist() for x in imgs], "energies_hartree": en, "text": p.read_text()})

    # ---- the restraint wrapper around the calculator (opt.py:286-343): the whole class as written, on the reference's own get_forces /
    # get_energy over the toy core
    Bias = grab(REF / "opt.py", ["HarmonicBiasCalculator"])["HarmonicBiasCalculator"]
    fx["harmonic_bias_wrapper"] = []
    for n_at, seed, k_ev, fz in ((5, 31, 10.0, []), (7, 32, 2.5, [0, 3])):
        x = toy_geometry(n_at, seed) * ANG2BOHR
        prs = [(int(i), int(j), float(t)) for i, j, t in zip(rng.integers(0, n_at, 5), rng.integers(0, n_at, 5), rng.uniform(0.9, 2.8, 5))]
        prs += [(0, n_at + 2, 1.0), (2, 2, 1.1)]
        base = ref_calc(ToyPairCore(n_at, seed=seed), freeze_atoms=fz)
        wb = Bias(base, k=k_ev)
        wb.set_pairs([(np.int64(i), j, np.float32(t)) for i, j, t in prs])          # set_pairs normalises the element types
        el = ["C"] * n_at
        rf = wb.get_forces(el, x.reshape(-1))
        re_ = wb.get_energy(el, x)
        e2, f2 = wb.get_energy_and_forces(el, x)
        e3, g3 = wb.get_energy_and_gradient(el, x.reshape(-1))
        fx["harmonic_bias_wrapper"].append({
            "n_atoms": n_at, "core_seed": seed, "k_ev_ang2": k_ev, "freeze_atoms": fz, "coords_bohr": x.tolist(),
            "pairs_in": [[int(i), int(j), float(np.float32(t))] for i, j, t in prs],
            "pairs_stored": [list(p) for p in wb._pairs], "pair_types": [[type(v).__name__ for v in p] for p in wb._pairs][0],
            "k_au_bohr2": wb.k_au_bohr2, "get_forces": {"energy": rf["energy"], "forces": np.asarray(rf["forces"]).tolist()},
            "get_energy": re_["energy"], "energy_and_forces": [e2, np.asarray(f2).tolist()], "energy_and_gradient": [e3, np.asarray(g3).tolist()],
            "forwarded_freeze_atoms": list(wb.freeze_atoms), "base_calls": base._core.calls})

    OUT_BOUNDARY.write_text(json.dumps(fx, separators=(",", ":")) + "\n")
    print(f"wrote {OUT_BOUNDARY} ({OUT_BOUNDARY.stat().st_size} bytes): " + ", ".join(f"{k}={len(v)}" for k, v in fx.items() if isinstance(v, list)))


def energy_series_fixtures():
    """trj2fig.py:112-205,287-303: the consumer of the `.trj` files -- energies re-scored frame by frame through the calculator, the
    dE series and its CSV.  The reference's `recompute_energies` is compiled as written; the ASE reader it calls (`read(path, index=":",
    format="xyz")`) is replaced by a plain XYZ-frame parser handing out objects with the two `Atoms` methods it uses, and
    `uma_pysis(charge=, spin=)` by the reference's own get_energy (compiled from uma_pysis.py as above) on the toy core."""
    import csv
    import io
    import contextlib

    sys.path.insert(0, str(ROOT / "tests"))
    from toy_core import ToyPairCore, toy_geometry
    from pdb2reaction_amd.formats import AU2KCALPERMOL

    EV2AU = 1.0 / AU2EV
    ns_calc = grab(REF / "uma_pysis.py", ["get_energy", "_au_energy", "_ensure_core"], cls="uma_pysis", extra={"EV2AU": EV2AU})
    RefCalc = type("RefCalc", (), {m: ns_calc[m] for m in ("get_energy", "_au_energy", "_ensure_core")})
    made = []

    class _Atoms:
        def __init__(self, sym, pos):
            self._s, self._p = list(sym), np.asarray(pos, dtype=float)

        def get_chemical_symbols(self):
            return list(self._s)

        def get_positions(self):
            return self._p.copy()

    def _read(path, index=":", format="xyz"):
        assert index == ":" and format == "xyz"
        lines = Path(path).read_text().split("\n")
        out, i = [], 0
        while i < len(lines) and lines[i].strip():
            n = int(lines[i])
            rows = [ln.split() for ln in lines[i + 2:i + 2 + n]]
            out.append(_Atoms([r[0] for r in rows], [[float(v) for v in r[1:4]] for r in rows]))
            i += 2 + n
        return out

    def _factory(charge=0, spin=1):
        me = RefCalc()
        me._core = ToyPairCore(made[-1]["n_atoms"], seed=made[-1]["seed"])
        made[-1]["ctor"] = {"charge": charge, "spin": spin}
        return me

    ns = grab(REF / "trj2fig.py", ["recompute_energies", "_parse_reference_spec", "_resolve_reference_index", "transform_series", "write_csv"],
              extra={"read": _read, "Atoms": _Atoms, "uma_pysis": _factory, "AU2KCALPERMOL": AU2KCALPERMOL, "csv": csv})
    rng = np.random.default_rng(20261006)
    fx: Dict[str, Any] = {"generated_by": "tools/make_reference_fixtures.py:energy_series_fixtures",
                          "reference": "t-0hmura/pdb2reaction (local checkout): pdb2reaction/trj2fig.py:112-205,287-303",
                          "constants": {"AU2KCALPERMOL": AU2KCALPERMOL, "ANG2BOHR": ANG2BOHR, "EV2AU": EV2AU}}
    fx["recompute_energies"] = []
    with tempfile.TemporaryDirectory() as td:
        for n_at, k, seed, charge, mult in ((4, 5, 3, 0, 1), (6, 3, 4, -1, 2), (3, 1, 5, None, 3), (5, 7, 6, 2, None)):
            sym = [str(a) for a in rng.choice(["H", "C", "N", "O"], size=n_at)]
            base = toy_geometry(n_at, seed)
            frames = [base + rng.normal(size=(n_at, 3)) * 0.05 for _ in range(k)]
            text = "".join(f"{n_at}\nframe {i}\n" + "".join(f"{s} {x:.15f} {y:.15f} {z:.15f}\n" for s, (x, y, z) in zip(sym, fr))
                           for i, fr in enumerate(frames))
            path = Path(td) / "a.trj"
            path.write_text(text)
            made.append({"n_atoms": n_at, "seed": seed})
            en = ns["recompute_energies"](path, charge, mult)
            fx["recompute_energies"].append({"text": text, "charge": charge, "multiplicity": mult, "n_atoms": n_at, "core_seed": seed,
                                             "ctor": made[-1]["ctor"], "energies": en})
    fx["transform_series"] = []
    series = [[-1.5, -1.49, -1.52, -1.4801], [0.25], list((rng.normal(size=9) * 0.01 - 228.0).tolist())]
    for en in series:
        for spec in (None, "init", "INIT", "none", "Null", "0", " 2 ", "-1", "7", "abc", "1.5", str(len(en) - 1), str(len(en))):
            for unit in ("kcal", "hartree"):
                for rev in (False, True):
                    rec = {"energies": en, "reference": spec, "unit": unit, "reverse_x": rev}
                    try:
                        v, lab, isd = ns["transform_series"](en, spec, unit, rev)
                        rec.update(values=v, ylabel=lab, is_delta=isd)
                    except Exception as exc:
                        rec.update(raises=type(exc).__name__, message=str(exc))
                    fx["transform_series"].append(rec)
    fx["write_csv"] = []
    with tempfile.TemporaryDirectory() as td:
        for en, unit, isd in ((series[0], "kcal", True), (series[2], "hartree", False), (series[1], "kcal", False)):
            vals = [float(x) for x in rng.normal(size=len(en)) * 12.0]
            p2 = Path(td) / "o.csv"
            with contextlib.redirect_stdout(io.StringIO()):
                ns["write_csv"](p2, en, vals, unit, isd)
            fx["write_csv"].append({"energies": en, "series": vals, "unit": unit, "is_delta": isd, "bytes": p2.read_bytes().decode("utf-8")})
    OUT_SERIES.write_text(json.dumps(fx, indent=1) + "\n")
    print(f"wrote {OUT_SERIES} ({OUT_SERIES.stat().st_size} bytes): " + ", ".join(f"{k}={len(v)}" for k, v in fx.items() if isinstance(v, list)))


if __name__ == "__main__":
    main()
