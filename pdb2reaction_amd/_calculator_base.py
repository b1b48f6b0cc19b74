"""pysisyphus ``Calculator`` base and physical constants -- the real ones when pysisyphus is importable,
otherwise a minimal stand-in with the same constructor contract.

The reference derives from ``pysisyphus.calculators.Calculator.Calculator`` and takes its constants
from ``pysisyphus.constants`` (reference ``pdb2reaction/uma_pysis.py:122-123``); pysisyphus builds
those from ``scipy.constants`` (CODATA), which is what the fallback below does as well.
"""
from __future__ import annotations

try:  # pragma: no cover - exercised only where pysisyphus is installed
    from pysisyphus.calculators.Calculator import Calculator  # type: ignore
    from pysisyphus.constants import ANG2BOHR, AU2EV, BOHR2ANG  # type: ignore

    HAVE_PYSISYPHUS = True
except Exception:
    HAVE_PYSISYPHUS = False
    try:
        from scipy import constants as _c

        BOHR2ANG = _c.value("Bohr radius") * 1e10
        AU2EV = _c.value("Hartree energy in eV")
    except Exception:  # CODATA 2022
        BOHR2ANG = 0.529177210544
        AU2EV = 27.211386245981
    ANG2BOHR = 1.0 / BOHR2ANG

    class Calculator:  # noqa: D401 - mirrors pysisyphus' constructor keywords
        """Stand-in for ``pysisyphus.calculators.Calculator.Calculator`` (bookkeeping only)."""

        def __init__(self, calc_number=0, charge=0, mult=1, base_name="calculator", pal=1, mem=1000,
                     check_mem=True, retry_calc=0, last_calc_cycle=None, clean_after=True, out_dir="qm_calcs",
                     force_num_hess_kwargs=None, **kwargs):
            self.calc_number = calc_number
            self.charge = int(charge)
            self.mult = int(mult)
            self.base_name = base_name
            self.pal = int(pal)
            self.mem = int(mem)
            self.retry_calc = int(retry_calc)
            self.out_dir = out_dir
            self.calc_counter = 0
            self.extra_kwargs = dict(kwargs)

        def get_energy(self, atoms, coords, **prepare_kwargs):
            raise NotImplementedError

        def get_forces(self, atoms, coords, **prepare_kwargs):
            raise NotImplementedError

        def get_hessian(self, atoms, coords, **prepare_kwargs):
            raise NotImplementedError
