"""MI355X-native UMA energy/force engine behind the pdb2reaction ``uma_pysis`` calculator API."""
from .uma_pysis import uma_pysis, CALC_KW, GEOM_KW_DEFAULT  # noqa: F401  (reference pdb2reaction/__init__.py:3)

__all__ = ["uma_pysis", "CALC_KW", "GEOM_KW_DEFAULT"]
__version__ = "0.1.0"
