"""State-dict -> UMXW0001 blob conversion with Mixture-of-Linear-Experts (MoLE) merging.

The reference obtains its model from fairchem (``pretrained_mlip.get_predict_unit(model, device)``, ``uma_pysis.py:246-250``;
the checkpoint is a gated download and fairchem is not installable here), and fairchem merges the MoLE experts of every SO(2)
linear once per system because the routing coefficients depend only on (composition, charge, spin, task) -- constant along a
reaction path.  This module does the same arithmetic on plain arrays:

* ``merge_mole``: ``W = sum_k alpha_k W_k`` for every expert-stacked tensor;
* ``from_state_dict``: prefix stripping, optional renaming, MoLE merge, shape validation against
  ``weights.param_shapes()`` (the names the engine loads), float32 conversion;
* ``convert``: state dict -> blob bytes for ``Engine.load_weights`` / ``umx_load_weights``.

[3P-UNVERIFIED] The real checkpoint's key names and its routing network could not be inspected.  The defaults below assume
fairchem-style names (``backbone.`` prefix, expert-stacked tensors of shape (n_experts, out, in) under ``<layer>.weights``);
pass ``rename=`` / ``coefficients=`` for anything else.  The routing network itself is NOT restated: hand in the coefficient
vector (e.g. dumped once from fairchem for the system at hand), or a state dict that is already merged.
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


def convert(state: Mapping[str, object], *, merged_for: Optional[Mapping[str, object]] = None, **kw) -> bytes:
    """State dict -> UMXW0001 blob (``weights.pack_blob`` of ``from_state_dict``).

    ``merged_for`` = ``weights.system_record(atomic_numbers, charge, spin, task)`` of the system the MoLE coefficients were
    computed for; it is stored in the blob trailer and checked whenever the blob is bound to a system.  It is REQUIRED when
    ``coefficients`` are given: a merged parameter set is only valid for that one (composition, charge, spin, task)."""
    if kw.get("coefficients") is not None and merged_for is None:
        raise ValueError("convert(coefficients=...) needs merged_for=weights.system_record(...): a MoLE merge is valid for one system only")
    meta = {"merged_for": dict(merged_for)} if merged_for is not None else None
    return W.pack_blob(from_state_dict(state, **kw), meta=meta)
