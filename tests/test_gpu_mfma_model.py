"""The engine's bf16x3 forward GEMM IS the bit-exact model of the matrix core (round 6).

`tools/mfma_emul.c` reproduces `v_mfma_f32_32x32x16_bf16` bit for bit (tests/test_mfma_model.py, against raw hardware results).  Here the
radial fc3 GEMMs of a real evaluation are replayed on that model: the A operand exactly as the GEMM read it (`a2q.*`: float32 quad-row blocks,
odd rows negated), the weights, the bias-seeded accumulators, the plane split (aligned leading planes or the plain nearest-bf16 ones), the
order of the six plane products of `umx_gemm_q.h` (LS = 2 on the 256 x 128 tiles of the edge-degree MLP, LS = 1 on the 256 x 256 tiles of the
layers' MLPs) -- and every output BIT of the engine must come out.  That pins both ways: the kernel computes what its comments say, and what
tools/cpu_mfma_gemm_bias.py concludes from the model holds for the engine."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from pdb2reaction_amd import synth, weights as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import mfma_model as MM  # noqa: E402

pytestmark = pytest.mark.gpu


def _unblock(raw: np.ndarray, cols: int) -> np.ndarray:
    """float32 quad-row blocks (4 rows x 16 columns = 256 B, umx_gemm_q.h "QF") -> row-major [rows, cols]"""
    rows = raw.size // cols
    return raw.reshape(rows // 4, cols // 16, 4, 16).transpose(0, 2, 1, 3).reshape(rows, cols)


@pytest.mark.parametrize("align", ["1", "0"])
def test_fc3_gemm_is_the_matrix_core_model_bit_for_bit(align, monkeypatch):
    from pdb2reaction_amd.engine import Engine

    monkeypatch.setenv("UMX_ALIGN_PLANES", align)
    lib = MM.load_lib()
    w = W.make_synthetic_weights(1)
    z, pos = synth.make_cluster(90)
    eng = Engine(0, precision="bf16x3")
    da, dw, dw2 = (C.c_int.in_dll(lib, n) for n in ("gemm_dem_a", "gemm_dem_w", "gemm_dem_w2"))
    try:
        eng.load_weights(w)
        eng.set_system(z)
        eng.debug_keep(True)
        eng.energy_forces(pos.astype(np.float32), forces=False)
        da.value = dw.value = dw2.value = 12 if align == "1" else 0
        for tag, prefix, scheme in (("deg", "edge_degree_embedding.rad_func", "ls2"), ("0", "blocks.0.edge_wise.so2_conv_1.rad_func", "ls1"),
                                    ("3", "blocks.3.edge_wise.so2_conv_1.rad_func", "ls1")):
            w3, b3 = w[f"{prefix}.fc3.weight"], w[f"{prefix}.fc3.bias"]
            rad = eng.debug_fetch(f"rad.{tag}").reshape(-1, w3.shape[0])
            ne = rad.shape[0]
            a = _unblock(eng.debug_fetch(f"a2q.{tag}"), W.RADIAL_HIDDEN)[:ne]
            sg = np.where(np.arange(ne) % 2 == 1, -1.0, 1.0).astype(np.float32)
            model = MM.gemm_bf16x3(a * sg[:, None], w3, b3, scheme, alt_rows=True)          # (the model negates odd rows itself)
            same = model.view(np.uint32) == rad.view(np.uint32)
            print(f"[align {align}] fc3.{tag}: {ne} x {w3.shape[0]} outputs, {int((~same).sum())} differ from the model")
            assert same.all(), (tag, int((~same).sum()), float(np.abs(model - rad).max()))
    finally:
        da.value = dw.value = dw2.value = 0
        eng.close()
