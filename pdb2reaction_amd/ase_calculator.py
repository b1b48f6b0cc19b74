"""ASE-style calculator facade -- the reference's secondary boundary for the DMF path.

The reference builds ``FAIRChemCalculator(predictor, task_name=...)`` and assigns it to ``atoms.calc`` of every DMF
image, with ``atoms.info["charge"]`` / ``atoms.info["spin"]`` set per image (reference ``path_opt.py:351-363,418-423``);
torch_dmf then calls ``get_potential_energy()`` / ``get_forces()`` (eV, eV/Angstrom) image by image.  This class offers
the same protocol on the HIP engine, plus ``calculate_images`` which evaluates a whole list of images in one launch.

ASE is not installed here: the class subclasses ``ase.calculators.calculator.Calculator`` when importable and otherwise
is a duck-typed stand-in that works with any object exposing ``get_positions()``, ``get_atomic_numbers()`` (or
``numbers``) and an ``info`` dict.
"""
from __future__ import annotations

from typing import Any, Dict, Optional, Sequence

import numpy as np

try:  # pragma: no cover - only where ASE exists
    from ase.calculators.calculator import Calculator as _AseBase, all_changes  # type: ignore

    HAVE_ASE = True
except Exception:
    HAVE_ASE = False
    all_changes = ["positions", "numbers", "cell", "pbc", "initial_charges", "initial_magmoms"]

    class _AseBase:  # minimal protocol
        def __init__(self, **kwargs):
            self.results: Dict[str, Any] = {}
            self.atoms = None

        def get_potential_energy(self, atoms=None, force_consistent=False):
            self.calculate(atoms, ["energy"], all_changes)
            return self.results["energy"]

        def get_forces(self, atoms=None):
            self.calculate(atoms, ["forces"], all_changes)
            return self.results["forces"]


def _numbers(atoms) -> np.ndarray:
    if hasattr(atoms, "get_atomic_numbers"):
        return np.asarray(atoms.get_atomic_numbers(), dtype=np.int32)
    return np.asarray(atoms.numbers, dtype=np.int32)


class UMXCalculator(_AseBase):
    """ASE calculator protocol on the MI355X engine (energies eV, forces eV/Angstrom)."""

    implemented_properties = ["energy", "forces"]

    def __init__(self, model: str = "uma-s-1p1", task_name: str = "omol", device: str = "auto", charge: int = 0, spin: int = 1,
                 radius: Optional[float] = None, max_neigh: Optional[int] = None, **kwargs):
        super().__init__(**kwargs)
        self.model, self.task_name, self.device = model, task_name, device
        self.default_charge, self.default_spin = int(charge), int(spin)
        self.radius, self.max_neigh = radius, max_neigh
        self._engine = None
        self._bound = None          # (numbers bytes, charge, spin)
        self._last = None           # (bound key, positions, results) of the most recent single-image evaluation
        if not hasattr(self, "results"):
            self.results = {}

    # ---- engine / system binding ------------------------------------------------------------------
    def _ensure(self, atoms):
        from .engine import Engine
        from . import weights as W
        from .uma_pysis import _device_index, resolve_weights

        z = _numbers(atoms)
        info = getattr(atoms, "info", {}) or {}
        charge, spin = int(info.get("charge", self.default_charge)), int(info.get("spin", self.default_spin))
        if self._engine is None:
            self._weights = resolve_weights(self.model)
            self._engine = Engine(_device_index(self.device))
            self._engine.load_weights(self._weights)
            from ._host import cap_pools_to_usable_cores

            cap_pools_to_usable_cores()          # the DMF driver's dense linear algebra between two calls must not starve the GPU feeder
        key = (z.tobytes(), charge, spin)
        if key != self._bound:
            # merged-MoLE weights depend on (composition, charge, spin, task): refuse to re-bind them to another system
            W.check_merged_for(self._weights, z, charge, spin, self.task_name)
            self._engine.set_system(z, charge=charge, spin=spin, task=self.task_name, radius=self.radius, max_neigh=self.max_neigh)
            self._bound = key
        return self._engine

    def close(self) -> None:
        """Release the engine (HBM workspace, weights) now; it is rebuilt lazily on the next calculation."""
        if self._engine is not None:
            self._engine.close()
            self._engine, self._bound, self._last = None, None, None

    # ---- ASE protocol -------------------------------------------------------------------------------
    def calculate(self, atoms=None, properties: Sequence[str] = ("energy", "forces"), system_changes=all_changes):
        if atoms is None:
            atoms = self.atoms
        if atoms is None:
            raise ValueError("no atoms to calculate")
        self.atoms = atoms
        eng = self._ensure(atoms)
        pos = np.array(atoms.get_positions(), dtype=np.float64)
        # get_potential_energy() followed by get_forces() on an unchanged image is ONE evaluation (ASE's own base class caches by atoms
        # state; the stand-in base used where ASE is absent does not, and the engine always computes both)
        if self._last is not None and self._last[0] == self._bound and np.array_equal(self._last[1], pos):
            self.results = dict(self._last[2])
            return
        e, f = eng.energy_forces(pos[None], forces=True)
        self.results = {"energy": float(e[0]), "forces": np.asarray(f[0], dtype=np.float64)}
        self._last = (self._bound, pos, dict(self.results))

    def calculate_images(self, images: Sequence[Any]):
        """One batched evaluation for a list of images of the SAME system; returns (E [K] eV, F [K,N,3] eV/A)."""
        if not images:
            raise ValueError("empty image list")
        eng = self._ensure(images[0])
        z0 = _numbers(images[0])
        for im in images[1:]:
            if not np.array_equal(_numbers(im), z0):
                raise ValueError("all images must share atom order and elements")
        pos = np.stack([np.asarray(im.get_positions(), dtype=np.float64) for im in images])
        e, f = eng.energy_forces(pos, forces=True)
        return e, np.asarray(f, dtype=np.float64)
