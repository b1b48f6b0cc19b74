"""CPU tests of the drop-in boundary: C-ABI library exports, loud failure without a GPU, and the
host-side mirror of reference pdb2reaction/uma_pysis.py (units, freeze semantics, FD-Hessian
assembly, batched API) exercised through a toy analytic core -- no engine compute on the CPU."""
import os
import re

import numpy as np
import pytest
import torch

from pdb2reaction_amd import engine as E, weights as W
import importlib

U = importlib.import_module("pdb2reaction_amd.uma_pysis")   # the package attribute of that name is the class
from pdb2reaction_amd.string import perpendicular, reparametrize_equal, select_hei_index, string_step, tangents

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    txt = open(os.path.join(ROOT, "include", "umx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(umx_[a-z_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = E.load_library()
    names = header_functions()
    assert len(names) >= 23
    for n in names:
        assert hasattr(lib, n), f"libumx.so does not export {n} declared in include/umx.h"
    assert sorted(E.EXPORTED_SYMBOLS) == names
    assert lib.umx_abi_version() == 10


def test_missing_library_is_loud(tmp_path):
    with pytest.raises(ImportError, match="no CPU fallback"):
        E.load_library(str(tmp_path / "nope.so"))


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure path")
def test_engine_refuses_to_run_without_gpu():
    with pytest.raises(E.UmxError, match="no HIP device"):
        E.Engine(0)
    calc = U.uma_pysis(model="synthetic")
    with pytest.raises(E.UmxError):
        calc.get_energy(["H", "H"], [0, 0, 0, 0, 0, 1.4])


def test_unit_constants_match_reference_definitions():
    # reference uma_pysis.py:127-129 with CODATA-2022 (scipy 1.15) -- SURVEY.md 8c
    assert U.EV2AU == pytest.approx(0.03674932217566444, rel=1e-14)
    assert U.F_EVAA_2_AU == pytest.approx(0.01944690379830087, rel=1e-14)
    assert U.H_EVAA_2_AU == pytest.approx(0.010290858305702375, rel=1e-14)


def test_calc_kw_and_ctor_mirror_reference():
    assert list(U.CALC_KW) == ["charge", "spin", "model", "task_name", "device", "workers", "workers_per_node", "max_neigh",
                               "radius", "r_edges", "out_hess_torch", "freeze_atoms", "hessian_calc_mode",
                               "return_partial_hessian", "hessian_double"]
    assert U.CALC_KW["model"] == "uma-s-1p1" and U.CALC_KW["hessian_calc_mode"] == "FiniteDifference"
    assert U.GEOM_KW_DEFAULT == {"coord_type": "cart", "freeze_atoms": []}
    c = U.uma_pysis(**{**U.CALC_KW, "charge": -1, "spin": 2, "freeze_atoms": [5, 2, 5]})
    assert c.freeze_atoms == [2, 5] and c.charge == -1 and c.mult == 2
    assert c.implemented_properties == ["energy", "forces", "hessian"]
    assert c._core is None                                     # lazy model load (reference :482,502-504)
    with pytest.raises(TypeError):
        U.uma_pysis(0, 1)                                       # keyword-only, like the reference
    # the one extra key: the GEMM arithmetic of the engine (taken out of **kwargs, validated up front, used at the lazy load)
    c = U.uma_pysis(precision="split-bf16")
    assert c._core_kw["precision"] == "split-bf16" and U.uma_pysis()._core_kw["precision"] is None
    with pytest.raises(ValueError, match="precision"):
        U.uma_pysis(precision="fp64")


def test_missing_weights_are_loud(tmp_path, monkeypatch):
    """reference uma_pysis.py:246-250 raises when the checkpoint cannot be obtained; a drop-in must not hand out
    energies of random weights (VERDICT r1 'silent garbage physics')."""
    monkeypatch.delenv("UMX_ALLOW_SYNTHETIC", raising=False)
    monkeypatch.delenv("UMX_WEIGHTS_DIR", raising=False)
    with pytest.raises(FileNotFoundError, match=r"uma-s-1p1.*UMX_WEIGHTS_DIR"):
        U.resolve_weights("uma-s-1p1")
    monkeypatch.setenv("UMX_WEIGHTS_DIR", str(tmp_path))
    with pytest.raises(FileNotFoundError, match=str(tmp_path / "uma-s-1p1.umxw").replace("\\", "/")):
        U.resolve_weights("uma-s-1p1")
    with pytest.raises(FileNotFoundError):                     # the calculator fails at first use, before touching the GPU
        U.uma_pysis().get_energy(["H", "H"], [0, 0, 0, 0, 0, 1.4])
    # explicit opt-ins
    w0 = U.resolve_weights("synthetic")
    assert np.array_equal(w0["mix_csd.bias"], W.make_synthetic_weights(0)["mix_csd.bias"])
    assert not np.array_equal(U.resolve_weights("synthetic:3")["mix_csd.bias"], w0["mix_csd.bias"])
    monkeypatch.setenv("UMX_ALLOW_SYNTHETIC", "1")
    with pytest.warns(RuntimeWarning, match="RANDOM synthetic weights"):
        U.resolve_weights("uma-s-1p1")
    monkeypatch.delenv("UMX_ALLOW_SYNTHETIC")
    # a blob in $UMX_WEIGHTS_DIR or given by path is loaded as is
    W.save_weights(str(tmp_path / "uma-s-1p1.umxw"), W.make_synthetic_weights(5))
    got = U.resolve_weights("uma-s-1p1")
    assert np.array_equal(got["mix_csd.bias"], W.make_synthetic_weights(5)["mix_csd.bias"])
    assert np.array_equal(U.resolve_weights(str(tmp_path / "uma-s-1p1.umxw"))["mix_csd.bias"], got["mix_csd.bias"])


def test_device_mapping():
    assert U._device_index("cuda:3") == 3 and U._device_index("cuda") == 0
    with pytest.raises(RuntimeError, match="no CPU path"):
        U._device_index("cpu")


class ToyCore:
    """Quadratic potential E = 1/2 x^T A x (eV, Angstrom) standing in for UMAcore."""

    def __init__(self, n):
        rng = np.random.default_rng(0)
        m = rng.standard_normal((3 * n, 3 * n))
        self.A = m @ m.T / (3 * n) + np.eye(3 * n)
        self.parallel_predict, self.has_torch_model = False, False
        self.device = torch.device("cpu")
        self.calls = 0

    def compute_batch(self, coords, *, forces=True):
        c = np.asarray(coords, dtype=np.float64)
        k = c.shape[0]
        x = c.reshape(k, -1)
        self.calls += 1
        e = 0.5 * np.einsum("ki,ij,kj->k", x, self.A, x)
        f = -(x @ self.A).reshape(c.shape).astype(np.float32)
        return {"energy": e, "forces": f if forces else None}

    def compute(self, coord_ang, *, forces=False, hessian=False):
        r = self.compute_batch(np.asarray(coord_ang)[None], forces=forces)
        return {"energy": float(r["energy"][0]), "forces": r["forces"][0] if forces else None, "hessian": None}


def make_calc(n, **kw):
    c = U.uma_pysis(**kw)
    c._core = ToyCore(n)
    return c


def test_get_forces_units_and_freeze():
    n = 5
    c = make_calc(n, freeze_atoms=[1, 3])
    x_bohr = np.random.default_rng(1).standard_normal(3 * n)
    r = c.get_forces(["C"] * n, x_bohr)
    x_ang = x_bohr * U.BOHR2ANG
    e_ev = 0.5 * x_ang @ c._core.A @ x_ang
    f_ev = -(c._core.A @ x_ang)
    assert r["energy"] == pytest.approx(e_ev * U.EV2AU, rel=1e-12)
    f = r["forces"]
    assert f.shape == (3 * n,) and f.dtype == np.float64
    exp = f_ev.astype(np.float32).astype(np.float64).reshape(n, 3) * U.F_EVAA_2_AU
    exp[[1, 3]] = 0.0
    assert np.allclose(f, exp.reshape(-1), rtol=1e-12, atol=0)
    assert c.get_energy(["C"] * n, x_bohr.reshape(n, 3))["energy"] == pytest.approx(r["energy"])


def test_get_forces_batch_equals_loop():
    n, k = 4, 6
    c = make_calc(n, freeze_atoms=[0])
    xb = np.random.default_rng(2).standard_normal((k, 3 * n))
    rb = c.get_forces_batch(["O"] * n, xb)
    assert rb["forces"].shape == (k, 3 * n) and rb["energy"].shape == (k,)
    for i in range(k):
        r = c.get_forces(["O"] * n, xb[i])
        assert rb["energy"][i] == pytest.approx(r["energy"], rel=1e-13)
        assert np.array_equal(rb["forces"][i], r["forces"])
    assert np.allclose(c.get_energy_batch(["O"] * n, xb)["energy"], rb["energy"])


@pytest.mark.parametrize("partial", [False, True])
def test_fd_hessian_semantics(partial):
    """Central differences of a quadratic potential reproduce A on active columns; frozen columns are
    skipped, output symmetrised and converted (reference :515-551, :595-686)."""
    n = 4
    c = make_calc(n, freeze_atoms=[2], return_partial_hessian=partial, out_hess_torch=False)
    x = np.random.default_rng(3).standard_normal(3 * n)
    r = c.get_hessian(["N"] * n, x)
    h = r["hessian"]
    act = [i for i in range(3 * n) if i // 3 != 2]
    a = c._core.A
    if partial:
        assert h.shape == (9, 9)
        exp = a[np.ix_(act, act)] * U.H_EVAA_2_AU
    else:
        assert h.shape == (12, 12)
        full = np.zeros_like(a)
        full[:, act] = a[:, act]
        exp = 0.5 * (full + full.T) * U.H_EVAA_2_AU
    assert h.dtype == np.float64
    assert np.allclose(h, exp, rtol=0, atol=2e-5)             # float32 forces / 2e-3 A step noise floor
    assert np.allclose(r["forces"].reshape(n, 3)[2], 0.0)
    c2 = make_calc(n, out_hess_torch=True, hessian_double=False)
    h2 = c2.get_hessian(["N"] * n, x)["hessian"]
    assert isinstance(h2, torch.Tensor) and h2.dtype == torch.float32 and h2.shape == (12, 12)


def test_fd_hessian_is_batched(monkeypatch):
    n = 6
    monkeypatch.setattr(U, "FD_BATCH", 8)
    c = make_calc(n)
    c.get_hessian(["C"] * n, np.zeros(3 * n))
    assert c._core.calls == 1 + -(-18 // 4)                   # base point + ceil(18 DOF / 4 DOF per batch)


def test_analytical_request_falls_back_to_fd():
    c = make_calc(3, hessian_calc_mode="Analytical", out_hess_torch=False)
    h = c.get_hessian(["H"] * 3, np.ones(9))["hessian"]       # engine exposes no torch model -> FD, like workers>1
    assert h.shape == (9, 9)


def test_graph_parallel_and_hessian_sharding_refuse_each_other():
    """workers == world switches the graph-parallel mode on (every force call a collective on ONE geometry); dealing FD
    columns over the ranks on top of that would hang or mix geometries.  The combination is refused in both orders."""
    n = 3
    c = make_calc(n)
    c._core._gp = object()                                    # what UMAcore.enable_graph_parallel leaves behind
    with pytest.raises(RuntimeError, match="graph-parallel"):
        c.enable_hessian_sharding(True)
    c._hess_shard = True                                      # switched on before the core existed / before the mode was entered
    with pytest.raises(RuntimeError, match="cannot be combined"):
        c.get_hessian(["H"] * n, np.ones(3 * n))
    with pytest.raises(RuntimeError, match="cannot be combined"):
        c.enable_graph_parallel(["H"] * n, True)
    c._hess_shard = False
    c._core._gp = None
    assert c.get_hessian(["H"] * n, np.ones(3 * n))["hessian"].shape == (9, 9)


def test_symbols_to_z():
    from pdb2reaction_amd.synth import symbols_to_z

    assert symbols_to_z(["h", "C", "cl", "FE"]).tolist() == [1, 6, 17, 26]
    with pytest.raises(ValueError):
        symbols_to_z(["Xx"])


def test_hei_rule():
    assert select_hei_index([0, 3, 1, 5, 2]) == 3             # highest internal local maximum
    assert select_hei_index([0, 1, 2, 3]) == 2                # no local max -> max internal
    assert select_hei_index([5, 1]) == 0


def test_string_update_properties():
    g = torch.Generator().manual_seed(0)
    x = torch.cumsum(torch.rand(9, 30, dtype=torch.float64, generator=g), 0)
    t = tangents(x)
    assert torch.allclose(t.norm(dim=1), torch.ones(9, dtype=torch.float64))
    f = torch.randn(9, 30, dtype=torch.float64, generator=g)
    assert (perpendicular(f, t) * t).sum(1).abs().max() < 1e-12
    y = reparametrize_equal(x)
    assert torch.equal(y[0], x[0]) and torch.equal(y[-1], x[-1])
    s = (y[1:] - y[:-1]).norm(dim=1)
    assert (s.max() - s.min()) / s.mean() < 0.15
    z = string_step(x, f, max_step=0.05)
    assert z.shape == x.shape and torch.isfinite(z).all()
    z2 = string_step(x, torch.zeros_like(f))
    assert torch.allclose(z2, reparametrize_equal(x))


def test_build_dependency_list_covers_every_kernel_header(monkeypatch):
    """a hand-kept header list missed umx_gemm_q.h, so the stale (git-ignored) libumx.so shipped."""
    import glob

    from pdb2reaction_amd import build as B

    deps = B.dependencies()
    names = {os.path.basename(d) for d in deps}
    assert {"umx_gemm_q.h", "umx_generators.h", "umx_api.hip", "umx.h"} <= names
    assert {os.path.basename(h) for h in glob.glob(os.path.join(B.CSRC, "*.h"))} <= names
    assert all(os.path.exists(d) for d in deps)
    real = os.path.getmtime
    t_lib = real(B.OUT)
    for touched in ("umx_gemm_q.h", "umx_generators.h", "umx_kernels_pl.h", "umx.h"):
        monkeypatch.setattr(os.path, "getmtime", lambda p, t=touched: t_lib + 5.0 if os.path.basename(p) == t else min(real(p), t_lib))
        assert B.needs_build(), touched
    monkeypatch.setattr(os.path, "getmtime", lambda p: min(real(p), t_lib))
    assert not B.needs_build()


def test_prebuilt_library_without_sources_still_loads(monkeypatch, tmp_path):
    """a deployment that ships libumx.so without csrc/ and include/ must not fail at import -- the library is then
    held to the libumx.so.digest written next to it at build time (and refused when THAT disagrees)."""
    import shutil

    from pdb2reaction_amd import build as B

    so = tmp_path / "libumx.so"
    shutil.copy(E.LIB_PATH, so)
    good = E.load_library().umx_build_digest().decode()
    monkeypatch.setattr(B, "dependencies", lambda: [str(tmp_path / "csrc" / "umx_api.hip"), str(tmp_path / "include" / "umx.h")])
    monkeypatch.setattr(E, "LIB_PATH", str(so))
    monkeypatch.delenv("UMX_LIBRARY", raising=False)
    monkeypatch.delenv("UMX_ALLOW_STALE", raising=False)
    monkeypatch.setattr(E, "_lib", None)
    (tmp_path / "libumx.so.digest").write_text(good + "\n")
    assert E.load_library().umx_build_digest().decode() == good          # sources absent, digest file agrees
    monkeypatch.setattr(E, "_lib", None)
    (tmp_path / "libumx.so.digest").write_text("0" * 64 + "\n")
    with pytest.raises(ImportError, match="built from other sources"):
        E.load_library()
    (tmp_path / "libumx.so.digest").unlink()
    monkeypatch.setattr(E, "_lib", None)
    with pytest.warns(RuntimeWarning, match="cannot be checked"):
        E.load_library()
    monkeypatch.setattr(E, "_lib", None)


def test_library_has_no_packed_fp32(tmp_path):
    """Round 3: packed-fp32 VALU instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32, what hipcc's SLP vectoriser makes of adjacent
    float operations) give timing-dependent results on gfx950 when other kernels share the SIMDs -- the library is compiled with
    -fno-slp-vectorize (build.FLAGS, part of the source digest).  Compile the smallest kernel header to assembly with those flags and
    check that none is left (a future flag or compiler change must not bring them back unnoticed)."""
    import re
    import subprocess

    from pdb2reaction_amd import build

    assert "-fno-slp-vectorize" in build.FLAGS
    src = tmp_path / "probe.hip"
    src.write_text('#include "umx_kernels_pl.h"\n#include "umx_kernels.h"\n')
    out = tmp_path / "probe.s"
    subprocess.run([build.find_hipcc(), *build.FLAGS, "-I", build.CSRC, "-S", "--cuda-device-only", str(src), "-o", str(out)], check=True, timeout=600)
    text = out.read_text()
    assert "k_norm_bwd" in text and "global_load" in text                      # device code of the kernels is really in there
    assert not re.search(r"\bv_pk_(mul|add|fma)_f32\b", text)
    # ... and the SHIPPED code object as a whole (GEMM and radial kernels, loop vectoriser, explicit float2 arithmetic included): the
    # gfx950 ELF inside libumx.so is disassembled and searched (build.check_no_packed_fp32, which build_library runs after linking)
    build.build_library(force=False, verbose=False)
    assert build.check_no_packed_fp32(build.OUT) >= 60                         # kernels checked
    dis = build.device_disassembly(build.OUT)
    assert "k_norm_bwd" in dis and "umx_gemm_q_kernel" in dis and "k_radial_head" in dis
