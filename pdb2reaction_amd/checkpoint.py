"""State-dict -> UMXW0001 blob conversion with Mixture-of-Linear-Experts (MoLE) merging.

The reference obtains its model from fairchem (``pretrained_mlip.get_predict_unit(model, device)``, ``uma_pysis.py:246-250``;
the checkpoint is a gated download and fairchem is not installable here), and fairchem merges the MoLE experts of every SO(2)
linear once per system because the routing coefficients depend only on (composition, charge, spin, task) -- constant along a
reaction path.  This module does the same arithmetic on plain arrays:

* ``merge_mole``: ``W = sum_k alpha_k W_k`` for every expert-stacked tensor;
* ``from_state_dict``: prefix stripping, optional renaming, MoLE merge, shape validation against
  ``weights.param_shapes()`` (the names the engine loads), float32 conversion;
* ``convert``: state dict -> blob bytes for ``Engine.load_weights`` / ``umx_load_weights``.

* ``mole_coefficients`` / ``convert_for_system``: the ROUTING network restated from SURVEY.md Appendix A.7 -- alpha =
  softmax(routing_mlp([mean_i comp_emb[Z_i] || sys_emb])) with sys_emb = SiLU(mix_csd([chg_emb, spin_emb, dataset_emb])) --
  so a fairchem-style state dict can be turned into a merged blob FOR ONE SYSTEM without fairchem.  The blob records that
  system (``merged_for``) and every consumer refuses to bind it to another one (``weights.check_merged_for``).

[3P-UNVERIFIED] The real checkpoint's key names and the exact form of its routing network could not be inspected (neither
fairchem nor a checkpoint exists here).  The defaults assume fairchem-style names (``backbone.`` prefix, expert-stacked tensors
of shape (n_experts, out, in) under ``<layer>.weights``, ``routing_mlp.<2i>.{weight,bias}`` Linear layers with SiLU between
them, ``composition_embedding.weight``); every name is a keyword argument, and ``coefficients=`` accepts a vector dumped
from fairchem itself, which is the safer path until the restatement has been checked against a real checkpoint.
"""
from __future__ import annotations

import warnings
from typing import Any, Callable, Dict, List, Mapping, Optional, Sequence, Tuple, Union

import numpy as np

from . import weights as W

Array = np.ndarray

ASSUME_UMA_S = "assume-uma-s"


class UnsupportedCheckpoint(ValueError):
    """The checkpoint's own model config asks for something libumx does not implement."""


# ---- the checkpoint's model config: read it, take what is a free parameter, refuse what the engine does not implement ----------
# What the HIP engine IS (csrc/umx_common.h constants, weights.py).  Every hyper-parameter below is compiled into the kernels; a
# checkpoint that says otherwise cannot be evaluated by this library and must be refused -- not evaluated as if it were UMA-S
# (VERDICT r3 item 6).  Names are fairchem's eSCN-MD backbone keywords as recalled in SURVEY.md Appendix A [3P-UNVERIFIED]; aliases cover
# the spellings seen in public configs.  A config that does not mention a key leaves that key unchecked (listed in the blob's
# ``model.unchecked_engine_keys``).
ENGINE_CONFIG: Dict[str, Any] = {
    "sphere_channels": W.SPHERE_CHANNELS, "hidden_channels": W.HIDDEN_CHANNELS, "edge_channels": W.EDGE_CHANNELS,
    "lmax": W.LMAX, "mmax": W.MMAX, "num_layers": W.NUM_LAYERS, "num_distance_basis": W.NUM_DISTANCE_BASIS,
    "distance_function": "gaussian", "norm_type": "rms_norm_sh", "act_type": "gate",
    "max_num_elements": W.MAX_NUM_ELEMENTS, "direct_forces": False, "regress_stress": False,
    "always_use_pbc": False, "use_pbc": False, "use_pbc_single": False,
}
# hyper-parameters with SEVERAL implemented values (round 5: the model variants SURVEY.md section 2.4 K8 / Appendix A mark "unsure" are
# evaluated, not refused): the checkpoint's value selects the kernels / host code, anything outside the set is refused
ENGINE_CHOICES: Dict[str, Tuple[str, ...]] = {"ff_type": W.FF_TYPES, "chg_spin_emb_type": W.EMB_TYPES}
_ALIASES = {"num_sphere_channels": "sphere_channels", "n_layers": "num_layers", "num_gaussians": "num_distance_basis",
            "max_neighbours": "max_neighbors", "max_neigh": "max_neighbors", "radius": "cutoff", "cutoff_radius": "cutoff"}
# free parameters the engine takes FROM the checkpoint (umx_set_system defaults; the reference reads them from the backbone,
# uma_pysis.py:301-309): validated for type / range only
_TAKEN = ("cutoff", "max_neighbors", "dataset_list", "num_experts", "ff_type", "chg_spin_emb_type", "use_dataset_embedding", "grid_resolution")
# keys that do not change the arithmetic of an inference call
_BENIGN = {"name", "model", "otf_graph", "activation_checkpointing", "regress_forces", "regress_energy", "use_compile", "compile",
           "heads", "backbone", "freeze_backbone", "pass_through_head_outputs", "model_id", "finetune", "cs_emb_grad", "dataset_emb_grad"}


def find_model_config(ckpt: Mapping[str, Any]) -> Optional[Dict[str, Any]]:
    """The backbone's hyper-parameter dict inside a checkpoint mapping: the first dict (breadth first) that has both ``lmax`` and
    ``sphere_channels`` (or their aliases) -- wherever the trainer nested it (``config.model.backbone``, ``model_config``,
    ``hyper_parameters`` ... [3P-UNVERIFIED]).  None when the checkpoint carries no such dict."""
    queue: List[Any] = [ckpt]
    seen = 0
    while queue and seen < 10000:
        node = queue.pop(0)
        seen += 1
        if isinstance(node, Mapping):
            keys = {_ALIASES.get(str(k), str(k)) for k in node.keys()}
            if "lmax" in keys and "sphere_channels" in keys:
                return {str(k): v for k, v in node.items()}
            queue.extend(v for k, v in node.items() if isinstance(v, (Mapping, list, tuple)) and "state_dict" not in str(k))
        elif isinstance(node, (list, tuple)):
            queue.extend(v for v in node if isinstance(v, (Mapping, list, tuple)))
    return None


def validate_model_config(config: Mapping[str, Any], *, strict_unknown: bool = False) -> Dict[str, Any]:
    """Hold a checkpoint's model config against what the engine implements.  Returns the ``model`` record that goes into the blob
    trailer: ``{"cutoff", "max_neighbors", "num_experts", "checked": [...], "unchecked_engine_keys": [...], "unknown_keys": [...]}``.

    * a key of :data:`ENGINE_CONFIG` with another value -> :class:`UnsupportedCheckpoint` naming EVERY mismatch (``lmax=3``: the Wigner
      blocks, the 9-row layouts and every GEMM shape are lmax=2);
    * ``ff_type`` ("spectral" | "grid") and ``chg_spin_emb_type`` ("rand_emb" | "pos_emb" | "lin_emb") select among the implemented
      forms (:data:`ENGINE_CHOICES`) and are recorded -- ``from_state_dict`` then requires exactly that variant's tensors;
    * ``cutoff`` / ``max_neighbors`` are the checkpoint's to choose (any positive value: they are run-time arguments of
      ``umx_set_system``); ``dataset_list`` is the checkpoint's too -- ANY order and any 1..32 names: the rows of the blob's
      dataset embedding follow it and task names are mapped through it (``Engine.dataset_list``); ``use_dataset_embedding=False``
      = a model without dataset embedding;
    * keys nobody knows are reported (warning, or an error with ``strict_unknown``) -- they may or may not change the arithmetic."""
    cfg = {_ALIASES.get(str(k), str(k)): v for k, v in config.items()}
    bad, checked = [], []
    for key, want in ENGINE_CONFIG.items():
        if key not in cfg:
            continue
        have = cfg[key]
        same = (str(have).lower().replace("-", "_") == str(want).lower()) if isinstance(want, str) else (have == want and type(have) in (type(want), int, bool, float))
        if isinstance(want, bool):
            same = bool(have) == want
        (checked if same else bad).append(key if same else f"{key}={have!r} (the engine implements {want!r})")
    model: Dict[str, Any] = {}
    if "cutoff" in cfg:
        c = float(cfg["cutoff"])
        if not (0.0 < c < 100.0):
            bad.append(f"cutoff={cfg['cutoff']!r} (must be a positive radius in Angstrom)")
        model["cutoff"] = c
    if "max_neighbors" in cfg:
        m = int(cfg["max_neighbors"])
        if m <= 0:
            bad.append(f"max_neighbors={cfg['max_neighbors']!r} (must be positive)")
        model["max_neighbors"] = m
    for key, allowed in ENGINE_CHOICES.items():
        if key in cfg:
            have = str(cfg[key]).lower().replace("-", "_")
            if have in allowed:
                model[key] = have
                checked.append(key)
            else:
                bad.append(f"{key}={cfg[key]!r} (the engine implements {list(allowed)!r})")
    if "dataset_list" in cfg:
        dl = tuple(str(x) for x in cfg["dataset_list"])
        if not (1 <= len(dl) <= W.MAX_DATASETS) or len(set(dl)) != len(dl):
            bad.append(f"dataset_list={list(dl)!r} (1..{W.MAX_DATASETS} distinct names)")
        model["dataset_list"] = list(dl)
    if "use_dataset_embedding" in cfg and not bool(cfg["use_dataset_embedding"]):
        model["dataset_list"] = []
    if "grid_resolution" in cfg and cfg["grid_resolution"] is not None:
        gr = cfg["grid_resolution"]
        pts = int(gr) * int(gr) if not isinstance(gr, (list, tuple)) else int(gr[0]) * int(gr[1])
        if not (1 <= pts <= W.GRID_POINTS_MAX):
            bad.append(f"grid_resolution={gr!r} ({pts} grid points; the grid kernels take up to {W.GRID_POINTS_MAX})")
    if "num_experts" in cfg:
        model["num_experts"] = int(cfg["num_experts"])
    if bad:
        raise UnsupportedCheckpoint("this checkpoint's model config is not the UMA-S (eSCN-MD, lmax = mmax = 2) model libumx implements: " + "; ".join(bad))
    unknown = sorted(k for k in cfg if k not in ENGINE_CONFIG and k not in _TAKEN and k not in _BENIGN)
    if unknown:
        msg = (f"checkpoint model config has {len(unknown)} key(s) this loader does not know: {unknown[:12]}{' ...' if len(unknown) > 12 else ''} -- "
               "they are NOT checked against the engine")
        if strict_unknown:
            raise UnsupportedCheckpoint(msg)
        warnings.warn(msg, RuntimeWarning, stacklevel=2)
    model["checked"] = sorted(checked)
    model["unchecked_engine_keys"] = sorted(k for k in list(ENGINE_CONFIG) + list(ENGINE_CHOICES) if k not in cfg)
    model["unknown_keys"] = unknown
    return model


def _model_record(model_config: Union[None, str, Mapping[str, Any]], strict_unknown: bool) -> Dict[str, Any]:
    if model_config is None:
        raise ValueError("convert: pass model_config= the checkpoint's own model config (checkpoint.find_model_config(ckpt)) so that it can be "
                         f"held against the engine, or model_config={ASSUME_UMA_S!r} to state explicitly that the tensors are UMA-S "
                         "(cutoff 6.0 A, max_neighbors 300, spectral feed-forward ...) without a config to prove it")
    if isinstance(model_config, str):
        if model_config != ASSUME_UMA_S:
            raise ValueError(f"model_config must be a mapping or {ASSUME_UMA_S!r}")
        return {"cutoff": W.CUTOFF, "max_neighbors": W.MAX_NEIGHBORS, "assumed": True}     # (the variant is then read off the tensors)
    return validate_model_config(model_config, strict_unknown=strict_unknown)


def _np(t) -> Array:
    """torch tensors (any device/dtype) or array-likes -> float64 numpy."""
    if hasattr(t, "detach"):
        t = t.detach().cpu().to(dtype=__import__("torch").float64).numpy()
    return np.asarray(t, dtype=np.float64)


def merge_mole(experts: Array, coefficients: Array) -> Array:
    """``sum_k alpha_k W_k`` for an expert stack (n_experts, ...); coefficients must have one entry per expert."""
    e = np.asarray(experts, dtype=np.float64)
    a = np.asarray(coefficients, dtype=np.float64).reshape(-1)
    if e.shape[0] != a.shape[0]:
        raise ValueError(f"MoLE merge: {e.shape[0]} experts but {a.shape[0]} coefficients")
    return np.tensordot(a, e, axes=(0, 0))


# fairchem-style module names -> the engine's flat names [3P-UNVERIFIED, from memory of fairchem-core 2.x models/uma]: the lookup tables of
# ChgSpinEmbedding live under ``.rand_emb``, DatasetEmbedding keeps one (1, C) table per dataset NAME, the S2-grid matrices are buffers of
# ``SO3_grid["lmax_lmax"]`` with shape (lat, long, 9).  Applied after prefix stripping and before a user ``rename``.
_DEFAULT_RENAMES = {
    "charge_embedding.rand_emb.weight": "charge_embedding.weight", "spin_embedding.rand_emb.weight": "spin_embedding.weight",
    "SO3_grid.lmax_lmax.to_grid_mat": "so3_grid.to_grid_mat", "SO3_grid.lmax_lmax.from_grid_mat": "so3_grid.from_grid_mat",
}
_DATASET_DICT = "dataset_embedding.dataset_emb_dict."
_DROPPED = ("SO3_grid.lmax_mmax.",)         # the mmax grid of the edge-wise activation variants the engine does not use (act_type = gate)


def _all_variant_shapes() -> Dict[str, Tuple[int, ...]]:
    out: Dict[str, Tuple[int, ...]] = {}
    for ff in W.FF_TYPES:
        for emb in W.EMB_TYPES:
            for k, v in W.param_shapes(ff, emb, grid_bias=True).items():
                out.setdefault(k, tuple(v))
    return out


def element_refs_from(atom_refs: Any = None, form_elem_refs: Any = None, *, task: str, formation_energy: bool = False) -> Array:
    """The per-element reference energies the engine adds (``element_refs``, K11) from what the reference hands to fairchem
    (``pretrained_mlip.get_reference_energies(model, "atom_refs" | "form_elem_refs")``, ``uma_pysis.py:231-239``).  Each may be a
    mapping task -> per-Z sequence (keys ``"omol"`` or ``"omol_elem_refs"``) or one per-Z sequence.  ``atom_refs`` are the linear
    element references of the task (added back onto the model energy); ``form_elem_refs`` are only applied when the caller asks for
    formation energies (then they are SUBTRACTED: E_form = E - sum_i form_ref[Z_i]) -- the reference itself never does
    (it uses total energies), so the default leaves them out [3P-UNVERIFIED]."""
    def pick(refs):
        if refs is None:
            return np.zeros(W.MAX_NUM_ELEMENTS)
        if isinstance(refs, Mapping):
            for key in (task, f"{task}_elem_refs"):
                if key in refs:
                    refs = refs[key]
                    break
            else:
                raise KeyError(f"element references have no entry for task {task!r} (keys: {sorted(map(str, refs))[:8]})")
        a = _np(refs).reshape(-1)
        if a.size > W.MAX_NUM_ELEMENTS and np.any(a[W.MAX_NUM_ELEMENTS:] != 0.0):
            raise ValueError(f"element references for Z >= {W.MAX_NUM_ELEMENTS} are not representable (max_num_elements = {W.MAX_NUM_ELEMENTS})")
        out = np.zeros(W.MAX_NUM_ELEMENTS)
        out[: min(a.size, W.MAX_NUM_ELEMENTS)] = a[: W.MAX_NUM_ELEMENTS]
        return out
    if atom_refs is None and form_elem_refs is None:
        raise ValueError("element_refs_from: give atom_refs and / or form_elem_refs")
    return pick(atom_refs) - (pick(form_elem_refs) if formation_energy else 0.0)


def from_state_dict(state: Mapping[str, object], *, prefix: str = "backbone.", coefficients: Optional[Array] = None,
                    expert_suffix: str = ".weights", merged_suffix: str = ".weight",
                    rename: Optional[Union[Mapping[str, str], Callable[[str], Optional[str]]]] = None,
                    extra: Optional[Mapping[str, object]] = None, strict: bool = True,
                    dataset_list: Optional[Sequence[str]] = None, variant: Optional[Mapping[str, Any]] = None,
                    task: Optional[str] = None, info: Optional[Dict[str, Any]] = None) -> Dict[str, Array]:
    """Turn a (fairchem-style) state dict into the engine's parameter dict.

    1. keys are stripped of ``prefix`` (keys without it are kept as they are); fairchem-style module names are mapped to the engine's
       (:data:`_DEFAULT_RENAMES`; the per-name dataset tables ``dataset_embedding.dataset_emb_dict.<name>.weight`` are stacked into
       ``dataset_embedding.weight`` in the order of ``dataset_list`` -- without one, in UMA's own order ``weights.DATASET_LIST`` when
       the names are exactly those, otherwise refused: an order is never invented; the order actually used is recorded in
       ``info["dataset_order"]`` and is what ``convert`` writes into the blob trailer; a stacked ``dataset_embedding.weight`` is taken
       as it is -- its rows then ARE in ``dataset_list`` order; (lat, long, 9) grid matrices are flattened to (G, 9));
    2. ``rename`` (dict or callable returning the new name, or None to drop the key) is applied;
    3. a key ending in ``expert_suffix`` whose tensor has one more dimension than the target is an expert stack: it is merged
       with ``coefficients`` and stored under ``<stem> + merged_suffix``;
    4. ``extra`` adds tensors that live outside the module tree: ``normalizer.rmsd``, ``element_refs`` -- or ``atom_refs`` /
       ``form_elem_refs`` as the reference obtains them (``uma_pysis.py:231-239``; :func:`element_refs_from`, needs ``task``) -- and,
       for a grid model whose SO3_Grid buffers are not in the state dict, ``so3_grid.to_grid_mat`` / ``so3_grid.from_grid_mat``
       dumped from the loaded model (they are never re-derived here);
    5. the model VARIANT is read off the tensors present (``weights.variant_of``) and, when ``variant`` (the validated config's
       ``ff_type`` / ``chg_spin_emb_type``) is given, must agree with it; every name of ``weights.param_shapes(**variant)`` must be
       present with exactly that shape (``strict``), unknown names raise.
    """
    loose = _all_variant_shapes()
    out: Dict[str, Array] = {}
    per_name: Dict[str, Array] = {}
    ext = dict(extra or {})
    if ("atom_refs" in ext or "form_elem_refs" in ext) and "element_refs" not in ext:
        if task is None:
            raise ValueError("from_state_dict: atom_refs / form_elem_refs need task= (the references are per task)")
        ext["element_refs"] = element_refs_from(ext.pop("atom_refs", None), ext.pop("form_elem_refs", None), task=task)
    ext.pop("atom_refs", None)
    ext.pop("form_elem_refs", None)
    for key, val in list(state.items()) + list(ext.items()):
        name = key[len(prefix):] if prefix and key.startswith(prefix) else key
        if any(name.startswith(d) for d in _DROPPED):
            continue
        name = _DEFAULT_RENAMES.get(name, name)
        if rename is not None:
            name = rename(name) if callable(rename) else rename.get(name, name)
            if name is None:
                continue
        arr = _np(val)
        if name.startswith(_DATASET_DICT) and name.endswith(".weight"):
            per_name[name[len(_DATASET_DICT):-len(".weight")]] = arr.reshape(-1)
            continue
        if name in ("so3_grid.to_grid_mat", "so3_grid.from_grid_mat") and arr.ndim == 3:
            arr = arr.reshape(-1, arr.shape[-1])
        if name.endswith(expert_suffix) and (name[: -len(expert_suffix)] + merged_suffix) in loose:
            target = name[: -len(expert_suffix)] + merged_suffix
            if arr.ndim == len(loose[target]) + 1:
                if coefficients is None:
                    raise ValueError(f"{key}: expert stack of {arr.shape[0]} needs MoLE coefficients")
                arr, name = merge_mole(arr, coefficients), target
        if name in out:
            raise KeyError(f"{name} assigned twice (last from {key})")
        out[name] = arr
    if per_name:
        if "dataset_embedding.weight" in out:
            raise KeyError("dataset embedding given both stacked (dataset_embedding.weight) and per name (dataset_emb_dict.*)")
        if dataset_list:
            order = [str(d) for d in dataset_list]
        elif set(per_name) == set(W.DATASET_LIST):
            order = list(W.DATASET_LIST)          # UMA's own task order; never an invented (e.g. alphabetical) one: the engine maps task names through it
        else:
            raise KeyError(f"per-name dataset tables {sorted(per_name)} without a dataset_list: the row order of the stacked table cannot be "
                           f"known (pass dataset_list=... or a model config that has one; UMA's own names are {list(W.DATASET_LIST)})")
        missing_ds = [d for d in order if d not in per_name]
        if missing_ds:
            raise KeyError(f"dataset_list names {missing_ds} have no dataset_embedding.dataset_emb_dict.<name>.weight in the state dict")
        extra_ds = sorted(d for d in per_name if d not in order)
        if extra_ds:
            raise KeyError(f"the state dict carries dataset tables {extra_ds} that dataset_list {order} does not name: refusing to drop them silently")
        out["dataset_embedding.weight"] = np.stack([per_name[d] for d in order])
        if info is not None:
            info["dataset_order"] = list(order)
    have = W.variant_of(out)
    if variant:
        for k in ("ff_type", "chg_spin_emb_type"):
            if variant.get(k) is not None and variant[k] != have[k]:
                raise UnsupportedCheckpoint(f"the model config says {k}={variant[k]!r} but the state dict carries the tensors of {have[k]!r}")
    if have["ff_type"] == "grid" and have["grid_points"] == 0:
        raise KeyError("grid feed-forward: the S2-grid matrices are not in the state dict (non-persistent SO3_Grid buffers) -- pass them as "
                       "extra={'so3_grid.to_grid_mat': model.backbone.SO3_grid['lmax_lmax'].to_grid_mat, 'so3_grid.from_grid_mat': ...} "
                       "dumped from the loaded model; they are data of the checkpoint and are not re-derived here")
    shapes = W.param_shapes(**have)
    res: Dict[str, Array] = {}
    for name, arr in out.items():
        if name not in shapes:
            if strict:
                raise KeyError(f"{name!r} is not a parameter of the UMA-S engine for this variant ({have}; see weights.param_shapes())")
            continue
        if tuple(arr.shape) != tuple(shapes[name]):
            raise ValueError(f"{name}: shape {tuple(arr.shape)} != expected {tuple(shapes[name])}")
        res[name] = np.ascontiguousarray(arr, dtype=np.float32)
    missing = [n for n in shapes if n not in res]
    if missing and strict:
        raise KeyError(f"{len(missing)} parameters missing, first: {missing[:4]}")
    return res


def _silu(x: Array) -> Array:
    return x / (1.0 + np.exp(-x))


def system_embedding(get: Callable[[str], Array], has: Callable[[str], bool], charge: int, spin: int, task: str,
                     dataset_list: Sequence[str]) -> Array:
    """SiLU(mix_csd([chg_emb | spin_emb (| dataset_emb)])) in float64 for whichever ChgSpinEmbedding / DatasetEmbedding form the state
    dict carries (the arithmetic of ``umx_set_system`` and of ``oracle.Oracle.system_embedding``):
    rand_emb tables (``….rand_emb.weight`` / ``….weight``), pos_emb frequencies (``….W``), lin_emb (``….lin_emb.weight/bias``);
    per-name dataset tables or a stacked table in ``dataset_list`` order; no dataset part when the model has none."""
    def cs(which: str, v: int) -> Array:
        null = which == "spin" and int(v) == 0
        if has(f"{which}_embedding.W"):
            ang = 2.0 * np.pi * float(v) * get(f"{which}_embedding.W").reshape(-1)
            e = np.concatenate([np.sin(ang), np.cos(ang)])
            return np.zeros_like(e) if null else e
        if has(f"{which}_embedding.lin_emb.weight"):
            return get(f"{which}_embedding.lin_emb.weight").reshape(-1) * (-100.0 if null else float(v)) + get(f"{which}_embedding.lin_emb.bias")
        tab = get(f"{which}_embedding.rand_emb.weight") if has(f"{which}_embedding.rand_emb.weight") else get(f"{which}_embedding.weight")
        return tab[int(v) + (W.CHARGE_OFFSET if which == "charge" else 0)]

    parts = [cs("charge", charge), cs("spin", spin)]
    if has(_DATASET_DICT + f"{task}.weight"):
        if task not in dataset_list:          # routing and engine must agree on which tasks exist (the engine maps names through this list)
            raise ValueError(f"task_name {task!r} not in {tuple(dataset_list)}")
        parts.append(get(_DATASET_DICT + f"{task}.weight").reshape(-1))
    elif has("dataset_embedding.weight"):
        if task not in dataset_list:
            raise ValueError(f"task_name {task!r} not in {tuple(dataset_list)}")
        parts.append(get("dataset_embedding.weight")[list(dataset_list).index(task)])
    return _silu(get("mix_csd.weight") @ np.concatenate(parts) + get("mix_csd.bias"))


def mole_coefficients(state: Mapping[str, object], atomic_numbers, charge: int, spin: int, task: str, *, prefix: str = "backbone.",
                      composition_key: str = "composition_embedding.weight", routing_prefix: str = "routing_mlp",
                      use_system_embedding: bool = True, dataset_list: Optional[Sequence[str]] = None) -> Array:
    """Expert mixing coefficients alpha (n_experts,) of ONE system -- SURVEY.md Appendix A.7, [3P-UNVERIFIED].

    composition = mean over atoms of ``composition_embedding[Z_i]`` (order independent); with ``use_system_embedding`` the
    charge / spin / task embedding of Appendix A.5, ``SiLU(mix_csd([chg_emb[q+100], spin_emb[s], dataset_emb[t]]))``, is
    appended; the routing MLP is every ``<routing_prefix>.<n>.weight/bias`` Linear in ascending n with SiLU between them;
    alpha = softmax of its output.  float64 throughout (the merge is setup work, once per system)."""
    def get(name: str) -> Array:
        for key in (prefix + name, name):
            if key in state:
                return _np(state[key])
        raise KeyError(f"state dict has no {prefix}{name!r} (needed for MoLE routing)")

    z = np.asarray(atomic_numbers, dtype=np.int64).reshape(-1)
    if z.size == 0:
        raise ValueError("mole_coefficients: empty system")
    x = get(composition_key)[z].mean(axis=0)
    if use_system_embedding:
        has = lambda name: (prefix + name) in state or name in state          # noqa: E731
        x = np.concatenate([x, system_embedding(get, has, charge, spin, task, tuple(dataset_list or W.DATASET_LIST))])
    layers = sorted({int(k[len(p):].split(".")[0]) for p in (prefix + routing_prefix + ".", routing_prefix + ".")
                     for k in state if k.startswith(p) and k.endswith(".weight")})
    if not layers:
        raise KeyError(f"state dict has no {prefix}{routing_prefix}.<n>.weight layers")
    for n, li in enumerate(layers):
        w, b = get(f"{routing_prefix}.{li}.weight"), get(f"{routing_prefix}.{li}.bias")
        if w.shape[1] != x.shape[0]:
            raise ValueError(f"{routing_prefix}.{li}.weight expects {w.shape[1]} inputs, got {x.shape[0]}")
        x = w @ x + b
        if n + 1 < len(layers):
            x = _silu(x)
    e = np.exp(x - x.max())
    return e / e.sum()


def convert_for_system(state: Mapping[str, object], atomic_numbers, charge: int, spin: int, task: str, *, routing_kw: Optional[dict] = None,
                       **kw) -> bytes:
    """State dict with MoLE experts + routing network -> merged blob for exactly this system: alpha from
    :func:`mole_coefficients`, merge, ``merged_for`` stamped.  Routing / composition tensors are dropped from the engine's
    parameter set (``rename`` is extended accordingly)."""
    rk = dict(routing_kw or {})
    cfg = kw.get("model_config")
    if "dataset_list" not in rk and isinstance(cfg, Mapping) and cfg.get("dataset_list"):
        rk["dataset_list"] = tuple(str(x) for x in cfg["dataset_list"])            # a stacked dataset table is indexed in the checkpoint's order
    rk_names = {k: v for k, v in rk.items() if k in ("routing_prefix", "composition_key")}
    alpha = mole_coefficients(state, atomic_numbers, charge, spin, task, prefix=kw.get("prefix", "backbone."), **rk)
    kw.setdefault("task", task)
    rk = rk_names
    user_rename = kw.pop("rename", None)
    drop = (rk.get("routing_prefix", "routing_mlp") + ".", rk.get("composition_key", "composition_embedding.weight"))

    def rename(name: str):
        if name.startswith(drop[0]) or name == drop[1]:
            return None
        if user_rename is None:
            return name
        return user_rename(name) if callable(user_rename) else user_rename.get(name, name)

    return convert(state, coefficients=alpha, merged_for=W.system_record(atomic_numbers, charge, spin, task), rename=rename, **kw)


def convert(state: Mapping[str, object], *, merged_for: Optional[Mapping[str, object]] = None,
            model_config: Union[None, str, Mapping[str, Any]] = None, strict_unknown: bool = False, **kw) -> bytes:
    """State dict -> UMXW0001 blob (``weights.pack_blob`` of ``from_state_dict``).

    ``model_config`` (REQUIRED): the checkpoint's own hyper-parameters (:func:`find_model_config`) -- held against the engine
    (:func:`validate_model_config`: anything libumx does not implement raises :class:`UnsupportedCheckpoint`), and its ``cutoff`` /
    ``max_neighbors`` go into the blob trailer (``model``), from where ``UMAcore`` takes them as the defaults of ``umx_set_system``
    exactly as the reference takes them from the backbone (``uma_pysis.py:301-309``).  ``"assume-uma-s"`` states the assumption
    explicitly for tensors that come without a config.  The normaliser (``normalizer.rmsd``) and the element references
    (``element_refs``) are mandatory (``extra=``): the reference applies them (``uma_pysis.py:231-239``), so a blob without them would
    return energies on another scale.

    ``merged_for`` = ``weights.system_record(atomic_numbers, charge, spin, task)`` of the system the MoLE coefficients were
    computed for; it is stored in the blob trailer and checked whenever the blob is bound to a system.  It is REQUIRED when
    ``coefficients`` are given: a merged parameter set is only valid for that one (composition, charge, spin, task)."""
    if kw.get("coefficients") is not None and merged_for is None:
        raise ValueError("convert(coefficients=...) needs merged_for=weights.system_record(...): a MoLE merge is valid for one system only")
    model = _model_record(model_config, strict_unknown)
    have = set(kw.get("extra") or {}) | set(state)
    if not any(k == "normalizer.rmsd" or k.endswith(".normalizer.rmsd") for k in have):
        raise KeyError("convert: 'normalizer.rmsd' is mandatory (extra={...}): the energy normaliser is part of the model's answer")
    if not any(k in ("element_refs", "atom_refs", "form_elem_refs") or k.endswith(".element_refs") for k in have):
        raise KeyError("convert: 'element_refs' is mandatory (extra={...}) -- or 'atom_refs' / 'form_elem_refs' as the reference obtains them "
                       "(pretrained_mlip.get_reference_energies, uma_pysis.py:231-239): the per-element reference energies of the task are "
                       "part of the model's answer")
    if kw.get("task") is None and merged_for is not None:
        kw["task"] = dict(merged_for).get("task")
    if model.get("dataset_list") == []:       # use_dataset_embedding = False: the model has no dataset embedding
        pfx = kw.get("prefix", "backbone.")
        tables = sorted(k for k in state if (k[len(pfx):] if pfx and k.startswith(pfx) else k).startswith("dataset_embedding."))
        if tables:
            raise UnsupportedCheckpoint(f"the model config says use_dataset_embedding=False but the state dict carries {tables[:3]}"
                                        f"{' ...' if len(tables) > 3 else ''}: config and tensors disagree")
    kw.setdefault("dataset_list", model.get("dataset_list") or None)
    kw.setdefault("variant", {k: model[k] for k in ENGINE_CHOICES if k in model} or None)
    info: Dict[str, Any] = {}
    params = from_state_dict(state, info=info, **kw)
    v = W.variant_of(params)
    model.update(ff_type=v["ff_type"], chg_spin_emb_type=v["chg_spin_emb_type"])
    if v["n_datasets"] == 0:
        model["dataset_list"] = []
    elif info.get("dataset_order"):           # per-name tables: exactly the order they were stacked in
        model["dataset_list"] = list(info["dataset_order"])
    elif kw.get("dataset_list") and not model.get("dataset_list"):
        model["dataset_list"] = [str(d) for d in kw["dataset_list"]]
    elif not model.get("dataset_list"):       # a stacked table without any list: only UMA's own five rows can be named
        if v["n_datasets"] != len(W.DATASET_LIST):
            raise UnsupportedCheckpoint(f"the dataset embedding has {v['n_datasets']} rows but no dataset_list says which tasks they are "
                                        "(model_config['dataset_list'] or dataset_list=...)")
        model["dataset_list"] = list(W.DATASET_LIST)
    if v["n_datasets"] and len(model["dataset_list"]) != v["n_datasets"]:
        raise UnsupportedCheckpoint(f"dataset_list has {len(model['dataset_list'])} names, the dataset embedding {v['n_datasets']} rows")
    meta: Dict[str, Any] = {"model": model}
    if merged_for is not None:
        meta["merged_for"] = dict(merged_for)
    return W.pack_blob(params, meta=meta)


def convert_checkpoint(ckpt: Mapping[str, Any], atomic_numbers, charge: int, spin: int, task: str, *, state_key: Optional[str] = None,
                       **kw) -> bytes:
    """A whole checkpoint mapping (what ``torch.load`` returns: model config + state dict [+ normaliser / references]) -> merged blob
    for one system.  The model config is FOUND in the checkpoint (:func:`find_model_config`) -- a checkpoint without one is refused
    unless ``model_config=`` is given -- validated, and the state dict is taken from ``state_key`` or the first of
    ``ema_state_dict`` / ``state_dict`` / ``model_state_dict`` / ``model`` that holds tensors [3P-UNVERIFIED]."""
    cfg = kw.pop("model_config", None)
    if cfg is None:
        cfg = find_model_config(ckpt)
        if cfg is None:
            raise UnsupportedCheckpoint("the checkpoint carries no model config (no mapping with 'lmax' and 'sphere_channels'): it cannot be "
                                        f"checked against the engine; pass model_config=... (or {ASSUME_UMA_S!r}) explicitly")
    state = None
    for key in ([state_key] if state_key else ["ema_state_dict", "state_dict", "model_state_dict", "model"]):
        cand = ckpt.get(key)
        if isinstance(cand, Mapping) and any(hasattr(v, "shape") for v in cand.values()):
            state = cand
            break
    if state is None:
        raise KeyError("convert_checkpoint: no state dict found in the checkpoint (tried " + (state_key or "ema_state_dict, state_dict, model_state_dict, model") + ")")
    return convert_for_system(state, atomic_numbers, charge, spin, task, model_config=cfg, **kw)
