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


def test_so2_conv_gemms_are_the_matrix_core_model_bit_for_bit():
    """The same for the SO(2) convolutions of a layer: conv-1 / conv-2 m = 0 (plain products: LS = 2 on the 256 x 128 tiles, aligned A planes) and the
    complex m = 1, 2 products (one accumulator, nearest-bf16 A planes, aligned weight planes; y_re = Xre.Wa - Xim.Wb, y_im = Xim.Wa + Xre.Wb joined by
    ONE float32 subtraction / addition in the epilogue) -- from the A operands exactly as the GEMMs read them (`y1q.*`, `hidq.*`) to every output bit
    of `hg` and `msg`."""
    from pdb2reaction_amd.engine import Engine

    lib = MM.load_lib()
    w = W.make_synthetic_weights(1)
    z, pos = synth.make_cluster(60)
    eng = Engine(0, precision="bf16x3")
    da, dw, dw2 = (C.c_int.in_dll(lib, n) for n in ("gemm_dem_a", "gemm_dem_w", "gemm_dem_w2"))
    try:
        eng.load_weights(w)
        eng.set_system(z)
        eng.debug_keep(True)
        eng.energy_forces(pos.astype(np.float32), forces=False)
        layer = 1
        b = f"blocks.{layer}.edge_wise"
        hg = eng.debug_fetch(f"hg.{layer}").reshape(-1, 1408)
        msg = eng.debug_fetch(f"msg.{layer}").reshape(-1, 1152)
        ne = hg.shape[0]
        sg = np.where(np.arange(ne) % 2 == 1, -1.0, 1.0).astype(np.float32)[:, None]
        y1 = _unblock(eng.debug_fetch(f"y1q.{layer}"), 2304)[:ne] * sg            # un-negated: the model negates odd rows itself
        hid = _unblock(eng.debug_fetch(f"hidq.{layer}"), 1152)[:ne] * sg
        dw.value = dw2.value = 12

        def plain(a, wt, bias):
            da.value = 12
            return MM.gemm_bf16x3(np.ascontiguousarray(a), wt, bias, "ls2", alt_rows=True)

        def cplx(a_re, a_im, wt, half):
            da.value = 0
            pr = [MM.gemm_bf16x3(np.ascontiguousarray(a), np.ascontiguousarray(wh), None, "plain", alt_rows=True)
                  for a in (a_re, a_im) for wh in (wt[:half], wt[half:])]           # [re.Wa, re.Wb, im.Wa, im.Wb]
            return pr[0] - pr[3], pr[2] + pr[1]

        checks = []
        checks.append(("conv-1 m0", plain(y1[:, :768], w[f"{b}.so2_conv_1.fc_m0.weight"], w[f"{b}.so2_conv_1.fc_m0.bias"]), hg[:, :640]))
        re, im = cplx(y1[:, 768:1280], y1[:, 1280:1792], w[f"{b}.so2_conv_1.so2_m_conv.0.fc.weight"], 256)
        checks += [("conv-1 m1 re", re, hg[:, 640:896]), ("conv-1 m1 im", im, hg[:, 896:1152])]
        re, im = cplx(y1[:, 1792:2048], y1[:, 2048:2304], w[f"{b}.so2_conv_1.so2_m_conv.1.fc.weight"], 128)
        checks += [("conv-1 m2 re", re, hg[:, 1152:1280]), ("conv-1 m2 im", im, hg[:, 1280:1408])]
        checks.append(("conv-2 m0", plain(hid[:, :384], w[f"{b}.so2_conv_2.fc_m0.weight"], w[f"{b}.so2_conv_2.fc_m0.bias"]), msg[:, :384]))
        re, im = cplx(hid[:, 384:640], hid[:, 640:896], w[f"{b}.so2_conv_2.so2_m_conv.0.fc.weight"], 256)
        checks += [("conv-2 m1 re", re, msg[:, 384:640]), ("conv-2 m1 im", im, msg[:, 640:896])]
        re, im = cplx(hid[:, 896:1024], hid[:, 1024:1152], w[f"{b}.so2_conv_2.so2_m_conv.1.fc.weight"], 128)
        checks += [("conv-2 m2 re", re, msg[:, 896:1024]), ("conv-2 m2 im", im, msg[:, 1024:1152])]
        bad = []
        for name, model, got in checks:
            same = np.ascontiguousarray(model).view(np.uint32) == np.ascontiguousarray(got).view(np.uint32)
            print(f"{name}: {model.shape[0]} x {model.shape[1]} outputs, {int((~same).sum())} differ from the model")
            if not same.all():
                bad.append((name, int((~same).sum()), float(np.abs(model - got).max()), np.argwhere(~same)[:4].tolist()))
        if bad and os.environ.get("UMX_MODEL_DUMP"):          # operands of the run for an offline replay (tools/mfma_model.py)
            np.savez_compressed(os.environ["UMX_MODEL_DUMP"], y1=y1, hid=hid, hg=hg, msg=msg)
        assert not bad, bad
    finally:
        da.value = dw.value = dw2.value = 0
        eng.close()
