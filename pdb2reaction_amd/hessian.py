"""Finite-difference Hessian assembly and atomic-unit conversion for the calculator boundary.

Semantics follow the reference calculator (``pdb2reaction/uma_pysis.py``): central differences of the float32 model
forces with h = 1e-3 Angstrom over the ACTIVE degrees of freedom only (``:595-686``), optional reduction to the
active block (``:678-684``), symmetrisation 0.5 (H + H^T), eV/A^2 -> Hartree/Bohr^2, float64 when ``hessian_double``
and torch-on-device or NumPy output (``:515-551``).  The implementation is ours: the 2 * 3N_active displaced geometries
are evaluated as batches of images through one engine call each instead of 2 serial calls per column.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from ._calculator_base import ANG2BOHR, AU2EV
from ._host import with_small_host_math

EV_TO_HARTREE = 1.0 / AU2EV
EV_PER_ANG_TO_AU = EV_TO_HARTREE / ANG2BOHR
EV_PER_ANG2_TO_AU = EV_TO_HARTREE / ANG2BOHR / ANG2BOHR
FD_STEP_ANG = 1.0e-3


def dof_partition(n_atoms: int, frozen: Sequence[int]) -> Tuple[List[int], List[int]]:
    """(active DOF indices, frozen DOF indices) for 0-based frozen atom indices."""
    fz = set(int(i) for i in frozen)
    active = [3 * a + c for a in range(n_atoms) if a not in fz for c in range(3)]
    dead = [3 * a + c for a in sorted(fz) for c in range(3)]
    return active, dead


def mask_frozen(forces: np.ndarray, frozen: Sequence[int]) -> np.ndarray:
    """Copy of `forces` ((N,3) or (K,N,3), eV/A) with the rows of frozen atoms set to zero."""
    if forces is None or len(frozen) == 0:
        return forces
    out = np.array(forces, copy=True)
    out[..., np.asarray(list(frozen), dtype=int), :] = 0.0
    return out


@with_small_host_math
def fd_hessian(batch_forces: Callable[[np.ndarray], np.ndarray], coord_ang: np.ndarray, frozen: Sequence[int], *, device,
               double: bool, partial: bool, batch: int = 64, step: float = FD_STEP_ANG, shard: bool = False, group=None, engine=None,
               batch_forces_dev: Optional[Callable] = None):
    """Central-difference Hessian in eV/A^2 as a torch tensor (n_out, 3, n_out, 3) on `device`.

    batch_forces(coords[K,N,3]) -> forces [K,N,3] float32.  Columns of frozen DOF stay zero (full output) or are
    dropped together with their rows (`partial`).

    batch_forces_dev (round 6, optional): the DEVICE form ``batch_forces_dev(coords: torch float32 [K,N,3] on `device`) -> forces torch
    float32 [K,N,3] on `device```.  With it the displaced geometries are built on the device (float64 base point +- step, rounded to the
    model's float32 positions exactly as the host path rounds them) and forces never leave it: no PCIe copy of 2 x 64 x N x 3 floats per
    call (``UMAcore.compute_batch_dev``: the engine's device-pointer entry on torch's current stream).  Same columns, bit for bit.

    Multi-GPU is OPT-IN: ``shard=True`` makes this call a COLLECTIVE over `group` (default: the world).
    Every rank of the group must enter it with the same geometry and frozen set; active columns are dealt round-robin
    (column k of the active list -> rank k mod G), every rank evaluates only its own displaced geometries and ONE
    all-reduce of the (3N x 3N) matrix assembles the result on every rank (columns are disjoint, so the sum is an exact
    gather).  The geometry is checked across ranks first (max |x - x_rank0| must be 0) so that ranks bound to different
    structures fail loudly instead of summing inconsistent columns.  With ``shard=False`` (the default, and what
    ``uma_pysis.get_hessian`` does unless sharding was enabled on the calculator) the call is purely local even inside an
    initialised process group -- a rank-0-only frequency step neither hangs nor returns a 1/G-filled matrix.

    `engine` (optional, sharded mode): the ``Engine`` behind `batch_forces`.  An engine that meets an fp16 range violation widens
    itself to bf16 forward planes locally (``Engine._widen``); the ranks then agree on that before the columns are summed (one
    MAX all-reduce of the flag) and a rank that computed columns in the narrower arithmetic computes them again, so one Hessian
    never mixes two arithmetics.
    """
    import torch
    import torch.distributed as dist

    world, rank = 1, 0
    x0 = np.asarray(coord_ang, dtype=np.float64)
    if shard:
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("fd_hessian(shard=True) needs an initialised torch.distributed process group")
        world = dist.get_world_size(group)
        rank = dist.get_rank(group)
        if world > 1:
            host_coll = dist.get_backend(group) == "gloo"
            ref = torch.as_tensor(x0, dtype=torch.float64, device="cpu" if host_coll else device).clone()
            mine_x = ref.clone()
            dist.broadcast(ref, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            bad = torch.tensor([float((ref - mine_x).abs().max()), float(len(frozen))], dtype=torch.float64, device=ref.device)
            lo, hi = bad.clone(), bad.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
            if float(hi[0]) != 0.0 or float(lo[1]) != float(hi[1]):
                raise RuntimeError("fd_hessian(shard=True): ranks entered with different geometries or frozen sets "
                                   f"(max |x - x_rank0| = {float(hi[0]):.3e}, frozen counts {int(lo[1])}..{int(hi[1])})")

    n = x0.shape[0]
    dof = 3 * n
    active, _ = dof_partition(n, frozen)
    dtype = torch.float64 if double else torch.float32
    hess = torch.zeros((dof, dof), device=device, dtype=dtype)
    per_call = max(batch // 2, 1)
    mine = active[rank::world] if world > 1 else active

    x0_dev = torch.as_tensor(x0, dtype=torch.float64, device=device) if batch_forces_dev is not None else None

    def my_columns():
        for start in range(0, len(mine), per_call):
            cols = mine[start: start + per_call]
            if batch_forces_dev is not None:
                m = len(cols)
                ct = torch.as_tensor(cols, device=device, dtype=torch.long)
                disp_t = x0_dev.reshape(1, dof).repeat(2 * m, 1)
                rows = torch.arange(m, device=device)
                disp_t[2 * rows, ct] += step
                disp_t[2 * rows + 1, ct] -= step
                f = batch_forces_dev(disp_t.reshape(2 * m, n, 3).to(torch.float32)).reshape(2 * m, dof).to(dtype)
                hess[:, ct] = (-(f[0::2] - f[1::2]) / (2.0 * step)).T
                continue
            disp = np.repeat(x0[None], 2 * len(cols), axis=0)
            for m, k in enumerate(cols):
                a, c = divmod(k, 3)
                disp[2 * m, a, c] += step
                disp[2 * m + 1, a, c] -= step
            f = torch.from_numpy(np.ascontiguousarray(batch_forces(disp)).reshape(2 * len(cols), dof)).to(device, dtype=dtype)
            hess[:, torch.as_tensor(cols, device=device, dtype=torch.long)] = (-(f[0::2] - f[1::2]) / (2.0 * step)).T

    was_wide = bool(getattr(engine, "widened", False))
    my_columns()
    if world == 1 and engine is not None and bool(getattr(engine, "widened", False)) and not was_wide:
        # the engine left the fp16 operand range in the middle of THIS Hessian and widened itself: the columns computed before the switch
        # are in the narrower arithmetic -- compute all of them again, so that one Hessian never mixes two arithmetics
        my_columns()
    if world > 1 and engine is not None:
        flag = torch.tensor([1.0 if engine.widened else 0.0], dtype=torch.float64,
                            device="cpu" if dist.get_backend(group) == "gloo" else device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        if float(flag[0]) > 0.0 and not was_wide:
            # some rank left the fp16 operand range during THIS Hessian: all ranks move to bf16 forward planes, and every rank
            # repeats its columns (those it computed before the switch included) so that all columns share one arithmetic
            if not engine.widened:
                engine.widen("fp16 range violation on a peer rank during a sharded FD Hessian")
            my_columns()
    if world > 1:
        if dist.get_backend(group) == "gloo" and hess.device.type != "cpu":
            host = hess.cpu()
            dist.all_reduce(host, group=group)
            hess.copy_(host)
        else:
            dist.all_reduce(hess, group=group)
    if partial:
        idx = torch.as_tensor(active, device=device, dtype=torch.long)
        hess = hess.index_select(0, idx).index_select(1, idx)
        n = len(active) // 3
    return hess.view(n, 3, n, 3)


def active_trim(hess, frozen: Sequence[int], *, partial: bool):
    """Freeze semantics for a Hessian that was NOT built column by column (an analytical one handed over by a core that
    exposes it; reference ``:569-592``): (n,3,n,3) eV/A^2 -> the active block (`partial`) or the full matrix with the
    COLUMNS of frozen DOF zeroed -- the same shape of result the finite-difference route gives.  No frozen atoms: unchanged."""
    import torch

    n = hess.size(0)
    fz = sorted(set(int(i) for i in frozen))
    if not fz:
        return hess
    active, dead = dof_partition(n, fz)
    h2 = hess.reshape(3 * n, 3 * n)
    if partial:
        idx = torch.as_tensor(active, device=h2.device, dtype=torch.long)
        m = len(active) // 3
        return h2.index_select(0, idx).index_select(1, idx).view(m, 3, m, 3)
    h2 = h2.clone()
    h2[:, torch.as_tensor(dead, device=h2.device, dtype=torch.long)] = 0.0
    return h2.view(n, 3, n, 3)


def hessian_to_au(hess, *, double: bool, as_torch: bool):
    """(n,3,n,3) eV/A^2 -> symmetrised (3n,3n) Hartree/Bohr^2 (torch on device or NumPy)."""
    import torch

    n = hess.size(0)
    h2 = hess.reshape(3 * n, 3 * n)
    h2 = 0.5 * (h2 + h2.T) * EV_PER_ANG2_TO_AU
    if double:
        h2 = h2.to(torch.float64)
    return h2.detach() if as_torch else h2.detach().cpu().numpy()
