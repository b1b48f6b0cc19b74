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

from typing import Callable, Dict, Mapping, Optional, Union

import numpy as np

from . import weights as W

Array = np.ndarray


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


def from_state_dict(state: Mapping[str, object], *, prefix: str = "backbone.", coefficients: Optional[Array] = None,
                    expert_suffix: str = ".weights", merged_suffix: str = ".weight",
                    rename: Optional[Union[Mapping[str, str], Callable[[str], Optional[str]]]] = None,
                    extra: Optional[Mapping[str, object]] = None, strict: bool = True) -> Dict[str, Array]:
    """Turn a (fairchem-style) state dict into the engine's parameter dict.

    1. keys are stripped of ``prefix`` (keys without it are kept as they are);
    2. ``rename`` (dict or callable returning the new name, or None to drop the key) is applied;
    3. a key ending in ``expert_suffix`` whose tensor has one more dimension than the target is an expert stack: it is merged
       with ``coefficients`` and stored under ``<stem> + merged_suffix``;
    4. ``extra`` adds tensors that live outside the module tree (``normalizer.rmsd``, ``element_refs``);
    5. every name of ``weights.param_shapes()`` must be present with exactly that shape (``strict``), unknown names raise.
    """
    shapes = W.param_shapes()
    out: Dict[str, Array] = {}
    for key, val in list(state.items()) + list((extra or {}).items()):
        name = key[len(prefix):] if prefix and key.startswith(prefix) else key
        if rename is not None:
            name = rename(name) if callable(rename) else rename.get(name, name)
            if name is None:
                continue
        arr = _np(val)
        if name.endswith(expert_suffix) and (name[: -len(expert_suffix)] + merged_suffix) in shapes:
            target = name[: -len(expert_suffix)] + merged_suffix
            if arr.ndim == len(shapes[target]) + 1:
                if coefficients is None:
                    raise ValueError(f"{key}: expert stack of {arr.shape[0]} needs MoLE coefficients")
                arr, name = merge_mole(arr, coefficients), target
        if name not in shapes:
            if strict:
                raise KeyError(f"{key} -> {name!r} is not a parameter of the UMA-S engine (see weights.param_shapes())")
            continue
        if tuple(arr.shape) != tuple(shapes[name]):
            raise ValueError(f"{key}: shape {tuple(arr.shape)} != expected {tuple(shapes[name])} for {name}")
        if name in out:
            raise KeyError(f"{name} assigned twice (last from {key})")
        out[name] = np.ascontiguousarray(arr, dtype=np.float32)
    missing = [n for n in shapes if n not in out]
    if missing and strict:
        raise KeyError(f"{len(missing)} parameters missing, first: {missing[:4]}")
    return out


def _silu(x: Array) -> Array:
    return x / (1.0 + np.exp(-x))


def mole_coefficients(state: Mapping[str, object], atomic_numbers, charge: int, spin: int, task: str, *, prefix: str = "backbone.",
                      composition_key: str = "composition_embedding.weight", routing_prefix: str = "routing_mlp",
                      use_system_embedding: bool = True) -> Array:
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
        if task not in W.DATASET_LIST:
            raise ValueError(f"task_name {task!r} not in {W.DATASET_LIST}")
        v = np.concatenate([get("charge_embedding.weight")[int(charge) + W.CHARGE_OFFSET], get("spin_embedding.weight")[int(spin)],
                            get("dataset_embedding.weight")[W.DATASET_LIST.index(task)]])
        x = np.concatenate([x, _silu(get("mix_csd.weight") @ v + get("mix_csd.bias"))])
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
    alpha = mole_coefficients(state, atomic_numbers, charge, spin, task, prefix=kw.get("prefix", "backbone."), **rk)
    user_rename = kw.pop("rename", None)
    drop = (rk.get("routing_prefix", "routing_mlp") + ".", rk.get("composition_key", "composition_embedding.weight"))

    def rename(name: str):
        if name.startswith(drop[0]) or name == drop[1]:
            return None
        if user_rename is None:
            return name
        return user_rename(name) if callable(user_rename) else user_rename.get(name, name)

    return convert(state, coefficients=alpha, merged_for=W.system_record(atomic_numbers, charge, spin, task), rename=rename, **kw)


def convert(state: Mapping[str, object], *, merged_for: Optional[Mapping[str, object]] = None, **kw) -> bytes:
    """State dict -> UMXW0001 blob (``weights.pack_blob`` of ``from_state_dict``).

    ``merged_for`` = ``weights.system_record(atomic_numbers, charge, spin, task)`` of the system the MoLE coefficients were
    computed for; it is stored in the blob trailer and checked whenever the blob is bound to a system.  It is REQUIRED when
    ``coefficients`` are given: a merged parameter set is only valid for that one (composition, charge, spin, task)."""
    if kw.get("coefficients") is not None and merged_for is None:
        raise ValueError("convert(coefficients=...) needs merged_for=weights.system_record(...): a MoLE merge is valid for one system only")
    meta = {"merged_for": dict(merged_for)} if merged_for is not None else None
    return W.pack_blob(from_state_dict(state, **kw), meta=meta)
