"""UMA calculator for pysisyphus -- MI355X-native drop-in for ``pdb2reaction.uma_pysis``.

Mirrors the reference interface (reference ``pdb2reaction/uma_pysis.py``):

* ``CALC_KW`` / ``GEOM_KW_DEFAULT`` (``:132-165``), unit constants (``:127-129``);
* ``UMAcore`` (``:170-419``): ``compute(coord_ang, forces=, hessian=)`` -> eV, eV/A;
* ``uma_pysis(Calculator)`` (``:425-780``): ``get_energy / get_forces / get_hessian(elem, coords)``
  with coordinates in Bohr, results in Hartree, Hartree/Bohr, Hartree/Bohr^2, frozen-atom force
  zeroing (``:561-567``), finite-difference Hessian (``:595-686``), Hessian symmetrisation / unit
  conversion / dtype (``:515-551``), mode dispatch (``:708-780``);
* ``run_pysis`` (``:784-789``).

What differs underneath: ``predict_unit.predict(batch)`` (fairchem, ``:373,385``) is replaced by the
HIP engine (``libumx.so``), and two batched entry points are added so a string driver can evaluate
all images at once: ``UMAcore.compute_batch`` and ``uma_pysis.get_forces_batch``.  The FD Hessian
uses the batched path (the 2*3N_active displaced geometries are images of one batch) but keeps the
reference's arithmetic: float32 forces, central difference with h = 1e-3 A, columns assembled in
float64 when ``hessian_double``.

There is no CPU fallback: importing works anywhere, evaluating requires a gfx950 GPU and the built
library, and fails loudly otherwise.
"""
from __future__ import annotations

import os
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from ._calculator_base import ANG2BOHR, AU2EV, BOHR2ANG, Calculator  # noqa: F401
from . import hessian as H
from ._host import cap_pools_to_usable_cores
from . import synth
from . import weights as W

# ------------ unit conversion constants (reference uma_pysis.py:127-129) ----------------------
EV2AU = H.EV_TO_HARTREE                   # eV -> Hartree
F_EVAA_2_AU = H.EV_PER_ANG_TO_AU          # eV/A -> Hartree/Bohr
H_EVAA_2_AU = H.EV_PER_ANG2_TO_AU         # eV/A^2 -> Hartree/Bohr^2

# reference uma_pysis.py:132-135
GEOM_KW_DEFAULT: Dict[str, Any] = {
    "coord_type": "cart",
    "freeze_atoms": [],
}

# reference uma_pysis.py:138-165
CALC_KW: Dict[str, Any] = {
    "charge": 0,
    "spin": 1,
    "model": "uma-s-1p1",
    "task_name": "omol",
    "device": "auto",
    "workers": 1,
    "workers_per_node": 1,
    "max_neigh": None,
    "radius": None,
    "r_edges": False,
    "out_hess_torch": True,
    "freeze_atoms": None,
    "hessian_calc_mode": "FiniteDifference",
    "return_partial_hessian": False,
    "hessian_double": True,
}

# number of displaced geometries evaluated per engine call while building an FD Hessian
FD_BATCH = int(os.environ.get("UMX_FD_BATCH", "64"))


def resolve_weights(model: str) -> "W.WeightSet":
    """Map the reference's ``model`` keyword to a merged UMA-S parameter set.

    * a path to a ``.umxw`` blob -> loaded as is;
    * a model name -> ``$UMX_WEIGHTS_DIR/<name>.umxw``;
    * ``"synthetic"`` / ``"synthetic:<seed>"`` -> the deterministic stand-in of ``weights.make_synthetic_weights``
      (what tests and bench.py use: the real checkpoint is a gated download that does not exist here);
    * anything else raises ``FileNotFoundError`` -- like the reference, which raises when the checkpoint cannot be
      obtained (``uma_pysis.py:246-250``).  A calculator must never hand out energies of random weights unasked;
      ``UMX_ALLOW_SYNTHETIC=1`` turns the miss into the synthetic stand-in (with a warning) for plumbing runs.
    """
    m = str(model)
    if m == "synthetic" or m.startswith("synthetic:"):
        return W.make_synthetic_weights(int(m.split(":", 1)[1]) if ":" in m else int(os.environ.get("UMX_SYNTHETIC_SEED", "0")))
    if os.path.isfile(m):
        return W.load_weights(m)
    wdir = os.environ.get("UMX_WEIGHTS_DIR")
    cand = os.path.join(wdir, f"{m}.umxw") if wdir else None
    if cand and os.path.isfile(cand):
        return W.load_weights(cand)
    if os.environ.get("UMX_ALLOW_SYNTHETIC", "0") == "1":
        import warnings

        warnings.warn(f"model {m!r}: no weight blob found, UMX_ALLOW_SYNTHETIC=1 -> RANDOM synthetic weights; energies and "
                      "forces are physically meaningless", RuntimeWarning, stacklevel=2)
        return W.make_synthetic_weights(int(os.environ.get("UMX_SYNTHETIC_SEED", "0")))
    where = cand if cand else f"$UMX_WEIGHTS_DIR/{m}.umxw (UMX_WEIGHTS_DIR is not set)"
    raise FileNotFoundError(
        f"UMA weights for model {m!r} not found: expected {where}, or pass model=<path to a .umxw blob> "
        "(convert a fairchem checkpoint with pdb2reaction_amd.checkpoint.convert). "
        "Use model='synthetic' or UMX_ALLOW_SYNTHETIC=1 only for tests and benchmarks.")


def _device_index(device: str) -> int:
    """'auto' | 'cuda' | 'cuda:N' -> HIP ordinal; 'cpu' is refused (no CPU path exists)."""
    d = str(device).lower()
    if d == "cpu":
        raise RuntimeError("device='cpu' requested, but this calculator has no CPU path; it requires an MI355X (gfx950) GPU.")
    if d in ("auto", "cuda", "hip", "gpu"):
        return int(os.environ.get("LOCAL_RANK", "0")) if d == "auto" and "LOCAL_RANK" in os.environ else 0
    if ":" in d:
        return int(d.split(":", 1)[1])
    raise ValueError(f"unrecognised device {device!r}")


# ===================================================================
#                         UMA core wrapper
# ===================================================================
class UMAcore:
    """Thin wrapper around the HIP engine (counterpart of reference ``UMAcore``, ``:170-419``)."""

    def __init__(
        self,
        elem: Sequence[str],
        *,
        charge: int = 0,
        spin: int = 1,
        model: str = "uma-s-1p1",
        task_name: str = "omol",
        device: str = "auto",
        workers: int = 1,
        workers_per_node: int = 1,
        max_neigh: Optional[int] = None,
        radius: Optional[float] = None,
        r_edges: bool = False,
        precision: Optional[str] = None,
    ):
        from .engine import Engine  # raises ImportError loudly when libumx.so is missing

        self.device_str = device
        self.workers = max(int(workers) if workers is not None else 1, 1)
        self.workers_per_node = max(int(workers_per_node) if workers_per_node is not None else 1, 1)
        # The reference's workers>1 is graph-parallel inference of ONE image (ParallelMLIPPredictUnit,
        # :220-242).  Here images are the parallel unit (one engine per GPU, see parallel.py); inside
        # a process `workers` only keeps the reference's side effect: no analytical Hessian.
        self.parallel_predict = self.workers > 1
        self.has_torch_model = False       # no nn.Module is exposed -> analytical Hessian unavailable
        self.elem = [e.capitalize() for e in elem]
        self.charge = charge
        self.spin = spin
        self.task_name = task_name
        self._max_neigh_user = max_neigh
        self._radius_user = radius
        self._r_edges_user = r_edges

        weights = resolve_weights(model)                    # raises FileNotFoundError before any GPU work
        # graph defaults come from the MODEL, as in the reference (backbone.cutoff / backbone.max_neighbors, fallback 6.0 A;
        # uma_pysis.py:301-309): a converted checkpoint records them in the blob trailer (checkpoint.validate_model_config)
        model_rec = (getattr(weights, "meta", None) or {}).get("model") or {}
        if radius is None and model_rec.get("cutoff") is not None:
            radius = float(model_rec["cutoff"])
        if max_neigh is None and model_rec.get("max_neighbors") is not None:
            max_neigh = int(model_rec["max_neighbors"])
        self.model_record = dict(model_rec)
        if r_edges:
            import warnings
            warnings.warn("uma_pysis: r_edges=True has no effect here: the radius graph is always built on the device for every call "
                          "(the reference collates with otf_graph=True as well, uma_pysis.py:322, so pre-computed edges are not what its model uses)",
                          RuntimeWarning, stacklevel=3)
        if self.workers_per_node != 1:
            import warnings
            warnings.warn(f"uma_pysis: workers_per_node={self.workers_per_node} has no effect here: there are no Ray actors to place "
                          "(uma_pysis.py:228-242); under torch.distributed the launcher decides which rank runs on which node",
                          RuntimeWarning, stacklevel=3)
        self.z = synth.symbols_to_z(self.elem)
        W.check_merged_for(weights, self.z, charge, spin, task_name)   # a MoLE merge is valid for one system only
        self.engine = Engine(_device_index(device), precision=precision)     # None: UMX_PRECISION (default "auto")
        cap_pools_to_usable_cores()                       # BLAS pools sized for the machine inside a CPU-quota container starve the GPU feeder
        self.engine.load_weights(weights)
        self.engine.set_system(self.z, charge=charge, spin=spin, task=task_name, radius=radius, max_neigh=max_neigh)
        self._gp = None
        if self.workers > 1 and os.environ.get("UMX_WORKERS_GP", "1") != "0":
            # the reference's workers > 1 IS graph-parallel inference of one structure over `workers` processes
            # (ParallelMLIPPredictUnit, :220-242).  Under torch.distributed with one rank per GPU and a world of exactly that
            # many ranks the same thing happens here: the graph of every geometry is partitioned over the ranks -- which is also
            # the route for a single structure too large for one GPU's workspace (UMX_ERR_CAPACITY).  Every rank must then call
            # the calculator with the same coordinates (SPMD), as every rank of such a job runs the same driver.
            import torch.distributed as dist

            if dist.is_available() and dist.is_initialized() and dist.get_world_size() == self.workers:
                self.enable_graph_parallel(True)

    @property
    def device(self):
        import torch
        return torch.device("cuda", self.engine.device)

    def enable_graph_parallel(self, on: bool = True, group=None) -> None:
        """Evaluate every geometry with its GRAPH partitioned over the ranks of `group` (default: the world) -- the reference's
        ``workers > 1`` mode (``ParallelMLIPPredictUnit``, ``:220-242``), for single large structures when there are fewer images
        than GPUs.  From then on ``compute`` / ``compute_batch`` are COLLECTIVES: every rank must call them with the same
        coordinates.  Images stay the preferred unit of parallelism (``parallel.ShardedImageEvaluator``): this mode moves
        9 x N x 4.6 KB through all-reduces per evaluation."""
        if not on:
            self._gp = None
            return
        from .parallel import GraphParallelEvaluator

        self._gp = GraphParallelEvaluator(self.engine, len(self.z), self.device, group)

    def _gp_eval(self, coords_ang: np.ndarray):
        import torch

        c = np.asarray(coords_ang, dtype=np.float64).reshape(-1, len(self.z), 3)
        es, fs = [], []
        for k in range(c.shape[0]):                       # images one after another, each spread over all ranks
            if not np.isfinite(c[k]).all():
                raise ValueError(f"non-finite position in image {k}")
            e, f = self._gp(torch.as_tensor(c[k], dtype=torch.float32, device=self.device))
            es.append(float(e[0]))
            if not np.isfinite(es[-1]):
                # the device-pointer entries are asynchronous and cannot refuse the result themselves (include/umx.h, UMX_ERR_RANGE);
                # every rank sees the same energy, so every rank raises here
                raise RuntimeError(f"image {k}: non-finite energy in graph-parallel mode (an activation beyond the fp16 operand range of "
                                   "UMX_PRECISION=split? create the calculators with UMX_PRECISION=split-bf16)")
            fs.append(f.cpu().numpy())
        return np.asarray(es, dtype=np.float64), np.stack(fs)

    # ----------------------------------------------------------------
    def compute_batch(self, coords_ang: np.ndarray, *, forces: bool = True) -> Dict[str, Any]:
        """Batched evaluation: (K,N,3) A -> {"energy": (K,) float64 eV, "forces": (K,N,3) float32 eV/A}."""
        if getattr(self, "_gp", None) is not None:
            e, f = self._gp_eval(coords_ang)
            return {"energy": e, "forces": f if forces else None}
        e, f = self.engine.energy_forces(np.asarray(coords_ang), forces=forces)
        return {"energy": e, "forces": f}

    def compute_batch_dev(self, pos32):
        """Device form of :meth:`compute_batch` for loops that stay on the GPU (the batched FD Hessian): ``pos32`` torch float32 [K,N,3] on
        the engine's device -> forces torch float32 [K,N,3] on the same device, through the engine's device-pointer entry on torch's
        current stream.  That entry is asynchronous and cannot refuse a non-finite energy itself, so the energies are looked at here (one
        scalar read per call); a range violation of the fast split-f16 mode widens the engine and repeats the call, as the host entry does."""
        import torch

        k = int(pos32.shape[0])
        pos32 = pos32.contiguous()
        e = torch.empty(k, dtype=torch.float64, device=pos32.device)
        f = torch.empty(k, pos32.shape[1], 3, dtype=torch.float32, device=pos32.device)
        for attempt in range(2):
            self.engine.energy_forces_dev(k, pos32.data_ptr(), e.data_ptr(), f.data_ptr(), stream=torch.cuda.current_stream(pos32.device).cuda_stream)
            if bool(torch.isfinite(e).all()):
                return f
            self.engine.take_range_error()
            if attempt == 0 and self.engine.widen("non-finite energy in a device-resident batch"):
                continue
            raise RuntimeError(f"non-finite energy in a device-resident batch (precision mode {self.engine.precision_mode()}): non-finite "
                               "coordinates, or an overflow that wider forward planes cannot cure")
        return f

    def compute(self, coord_ang: np.ndarray, *, forces: bool = False, hessian: bool = False) -> Dict[str, Any]:
        """Energy (eV) and optionally forces (eV/A) of one geometry; same contract as reference ``:330-419``."""
        if hessian:
            raise RuntimeError(
                "Analytical Hessian is not available when predictor workers > 1 "
                "or when predictor.model is not exposed. Use FiniteDifference Hessian."
            )
        if getattr(self, "_gp", None) is not None:
            e, f = self._gp_eval(coord_ang)
        else:
            e, f = self.engine.energy_forces(np.asarray(coord_ang, dtype=np.float64).reshape(1, -1, 3), forces=forces)
        return {"energy": float(e[0]), "forces": (f[0] if forces else None), "hessian": None}


# ===================================================================
#                    PySisyphus calculator class
# ===================================================================
class uma_pysis(Calculator):
    """PySisyphus-compatible UMA calculator (counterpart of reference ``:425-780``)."""

    implemented_properties = ["energy", "forces", "hessian"]

    def __init__(
        self,
        *,
        charge: int = CALC_KW["charge"],
        spin: int = CALC_KW["spin"],
        model: str = CALC_KW["model"],
        task_name: str = CALC_KW["task_name"],
        device: str = CALC_KW["device"],
        workers: int = CALC_KW["workers"],
        workers_per_node: int = CALC_KW["workers_per_node"],
        out_hess_torch: bool = CALC_KW["out_hess_torch"],
        max_neigh: Optional[int] = CALC_KW["max_neigh"],
        radius: Optional[float] = CALC_KW["radius"],
        r_edges: bool = CALC_KW["r_edges"],
        freeze_atoms: Optional[Sequence[int]] = CALC_KW["freeze_atoms"],
        hessian_calc_mode: str = CALC_KW["hessian_calc_mode"],
        return_partial_hessian: bool = CALC_KW["return_partial_hessian"],
        hessian_double: bool = CALC_KW["hessian_double"],
        **kwargs,
    ):
        # not a reference keyword: the arithmetic of the large GEMMs ("auto" | "split" | "split-bf16" | "bf16x3" | "fp32"; None = UMX_PRECISION).
        # Taken out of **kwargs so that the reference's signature stays as it is.
        precision = kwargs.pop("precision", None)
        if precision not in (None, "auto", "split", "split-f16", "split-bf16", "bf16x3", "split-exact", "fp32"):
            raise ValueError(f"precision must be auto, split, split-bf16, bf16x3 or fp32, got {precision!r}")
        super().__init__(charge=charge, mult=spin, **kwargs)
        self._core: Optional[UMAcore] = None
        self._core_kw = dict(
            charge=charge, spin=spin, model=model, task_name=task_name, device=device, workers=workers,
            workers_per_node=workers_per_node, max_neigh=max_neigh, radius=radius, r_edges=r_edges, precision=precision,
        )
        self._reserve_images = 0
        self.out_hess_torch = out_hess_torch
        self.hessian_calc_mode = hessian_calc_mode
        self.freeze_atoms: List[int] = sorted(set(int(i) for i in (freeze_atoms or [])))
        self.return_partial_hessian = bool(return_partial_hessian)
        self.hessian_double = bool(hessian_double)
        # multi-GPU FD Hessian is opt-in (enable_hessian_sharding): a collective must never start implicitly
        self._hess_shard = False
        self._hess_group = None

    # ---------- internals -------------------------------------------
    def _ensure_core(self, elem: Sequence[str]):
        # the first `elem` binds the instance for its lifetime, as in the reference (:502-504)
        if self._core is None:
            self._core = UMAcore(elem, **self._core_kw)
            if self._reserve_images and hasattr(self._core, "engine"):
                self._core.engine.reserve_images(self._reserve_images)
        return self._core

    def reserve_images(self, n_images: int) -> None:
        """Tell the engine that batches of up to ``n_images`` images are coming (a string that will grow to that size): its workspace is
        then allocated once instead of being re-allocated at every growth -- seconds each (``umx_reserve_images``).  A hint only."""
        self._reserve_images = max(0, int(n_images))
        eng = getattr(self._core, "engine", None)
        if eng is not None:
            eng.reserve_images(self._reserve_images)

    def close(self) -> None:
        """Release the engine (HBM workspace, weights) now instead of at garbage collection; the calculator can be used
        again afterwards (the core is rebuilt lazily on the next call)."""
        if self._core is not None:
            eng = getattr(self._core, "engine", None)
            if eng is not None:
                eng.close()
            self._core = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def enable_graph_parallel(self, elem: Sequence[str], on: bool = True, group=None) -> None:
        """Reference ``workers > 1`` semantics on the engine: partition the graph of each geometry over the ranks of `group`
        (see ``UMAcore.enable_graph_parallel``).  Every later ``get_energy / get_forces / get_hessian`` call is then a collective."""
        if on and self._hess_shard:
            raise RuntimeError("enable_graph_parallel: FD-Hessian column sharding is on (enable_hessian_sharding); the two cannot be combined")
        self._ensure_core(elem).enable_graph_parallel(on, group)

    def enable_hessian_sharding(self, on: bool = True, group=None) -> None:
        """Deal the FD-Hessian columns over the ranks of `group` (default: the world) -- c4's "freq Hessian (3N force
        batches) on 8 GPUs".  From then on ``get_hessian`` is a COLLECTIVE: every rank of the group must call it with the
        same geometry (checked).  Off by default, so a Hessian requested by one rank only stays a local computation."""
        if on and self._core is not None and getattr(self._core, "_gp", None) is not None:
            raise RuntimeError("enable_hessian_sharding: this calculator evaluates in graph-parallel mode (workers == world size, or "
                               "enable_graph_parallel) -- every force call is already a collective over all ranks on the SAME geometry, "
                               "so the columns cannot be dealt to different ranks.  Use one or the other (UMX_WORKERS_GP=0 keeps "
                               "workers>1 from switching the graph-parallel mode on)")
        self._hess_shard, self._hess_group = bool(on), group

    def _fd_hessian_ev(self, elem: Sequence[str], coord_ang: np.ndarray) -> Dict[str, Any]:
        """Base-point E/F plus the finite-difference Hessian (eV/A^2, torch on the core's device); hessian.fd_hessian."""
        core = self._ensure_core(elem)
        if self._hess_shard and getattr(core, "_gp", None) is not None:
            # with both on, every rank would displace DIFFERENT columns and enter a different number of graph-parallel
            # collectives -- a hang, or partial sums mixed across geometries.  The graph-parallel mode wins (it may be the only way the
            # structure fits); the columns are computed by all ranks together, un-sharded.
            raise RuntimeError("FD Hessian: column sharding (enable_hessian_sharding) and the graph-parallel mode (workers == world size / "
                               "enable_graph_parallel) cannot be combined: in graph-parallel mode every force call is a collective on one geometry")
        base = core.compute(coord_ang, forces=True, hessian=False)
        # the displaced geometries and their forces stay on the GPU when the core runs on the HIP engine (round 6); the graph-parallel mode and
        # stand-in cores keep the host form
        dev_fn = core.compute_batch_dev if (getattr(core, "_gp", None) is None and hasattr(core, "compute_batch_dev")
                                            and hasattr(getattr(core, "engine", None), "energy_forces_dev")
                                            and getattr(core.device, "type", "cpu") == "cuda") else None
        hess = H.fd_hessian(lambda c: core.compute_batch(c, forces=True)["forces"], coord_ang, self.freeze_atoms, device=core.device,
                            double=self.hessian_double, partial=self.return_partial_hessian, batch=FD_BATCH,
                            shard=self._hess_shard, group=self._hess_group, engine=getattr(core, "engine", None), batch_forces_dev=dev_fn)
        return {"energy": base["energy"], "forces": base["forces"], "hessian": hess}

    # ---------- PySisyphus API --------------------------------------
    def get_energy(self, elem, coords):
        core = self._ensure_core(elem)
        res = core.compute(_bohr_to_ang(coords), forces=False, hessian=False)
        return {"energy": res["energy"] * EV2AU}

    def get_forces(self, elem, coords):
        core = self._ensure_core(elem)
        res = core.compute(_bohr_to_ang(coords), forces=True, hessian=False)
        f_ev = H.mask_frozen(res["forces"], self.freeze_atoms)            # zeroed in eV/A before conversion (:701)
        return {"energy": res["energy"] * EV2AU, "forces": (np.asarray(f_ev, dtype=np.float64) * F_EVAA_2_AU).reshape(-1)}

    def get_forces_batch(self, elem, coords_batch):
        """All images of a string in ONE engine call.

        ``coords_batch``: (K, 3N) or (K, N, 3) Bohr.  Returns ``{"energy": (K,) Hartree,
        "forces": (K, 3N) Hartree/Bohr float64}`` -- per image exactly what ``get_forces`` returns.
        """
        core = self._ensure_core(elem)
        c = np.asarray(coords_batch, dtype=np.float64)
        k = c.shape[0]
        res = core.compute_batch(c.reshape(k, -1, 3) * BOHR2ANG, forces=True)
        f_ev = H.mask_frozen(res["forces"], self.freeze_atoms)
        return {"energy": np.asarray(res["energy"], dtype=np.float64) * EV2AU,
                "forces": (np.asarray(f_ev, dtype=np.float64) * F_EVAA_2_AU).reshape(k, -1)}

    def get_energy_batch(self, elem, coords_batch):
        core = self._ensure_core(elem)
        c = np.asarray(coords_batch, dtype=np.float64)
        res = core.compute_batch(c.reshape(c.shape[0], -1, 3) * BOHR2ANG, forces=False)
        return {"energy": np.asarray(res["energy"], dtype=np.float64) * EV2AU}

    def get_hessian(self, elem, coords):
        """Hessian per ``hessian_calc_mode`` with the reference's dispatch (``:708-780``): "Analytical" only when the core
        exposes a differentiable model and is not a parallel predictor, anything else (falsy, unknown, FiniteDifference) is
        the central-difference route.  The HIP engine exposes no such model (``UMAcore.has_torch_model`` is False), so on
        the engine every mode resolves to FiniteDifference -- exactly like the reference with ``workers > 1`` (``:736-737``)."""
        core = self._ensure_core(elem)
        coord_ang = _bohr_to_ang(coords)
        force_fd = core.parallel_predict or (not core.has_torch_model)
        mode = (self.hessian_calc_mode or "FiniteDifference").strip().lower()
        if (not force_fd) and mode in ("analytical", "analytic"):
            res = core.compute(coord_ang, forces=True, hessian=True)
            hess = H.active_trim(res["hessian"], self.freeze_atoms, partial=self.return_partial_hessian)
        else:
            res = self._fd_hessian_ev(elem, coord_ang)
            hess = res["hessian"]
        f_ev = H.mask_frozen(res["forces"], self.freeze_atoms)
        return {"energy": res["energy"] * EV2AU, "forces": (np.asarray(f_ev, dtype=np.float64) * F_EVAA_2_AU).reshape(-1),
                "hessian": H.hessian_to_au(hess, double=self.hessian_double, as_torch=self.out_hess_torch)}


def _bohr_to_ang(coords) -> np.ndarray:
    return np.asarray(coords, dtype=np.float64).reshape(-1, 3) * BOHR2ANG


# ---------- CLI ----------------------------------------
def run_pysis():
    """Enable ``uma_pysis input.yaml`` (reference ``:784-789``); needs pysisyphus installed."""
    try:
        from pysisyphus import run
    except ImportError as exc:  # pragma: no cover
        raise ImportError("run_pysis() needs pysisyphus, which is not installed in this environment") from exc
    run.CALC_DICT["uma_pysis"] = uma_pysis
    run.run()
