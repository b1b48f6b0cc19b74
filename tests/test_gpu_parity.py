"""GPU parity tests (through the C ABI): HIP engine vs the float64 CPU oracle, the committed golden
fixtures, and size-independent invariants at the full BASELINE sizes.

Tolerances are the north-star's: |dE| <= 1e-4 eV, max|dF| <= 1e-3 eV/A (BASELINE.json)."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from pdb2reaction_amd import synth, weights as W

pytestmark = pytest.mark.gpu

TOL_E = 1e-4   # eV
TOL_F = 1e-3   # eV/Angstrom


def check(engine, oracle, z, imgs, **sys_kw):
    p32 = np.asarray(imgs, dtype=np.float32)
    engine.set_system(z, **sys_kw)
    e, f = engine.energy_forces(p32)
    for k in range(len(p32)):
        e_ref, f_ref = oracle.energy_forces(z, p32[k].astype(np.float64), **sys_kw)
        assert abs(e[k] - e_ref) <= TOL_E, (k, e[k], e_ref)
        assert np.abs(f[k] - f_ref).max() <= TOL_F
    return e, f


@pytest.mark.parametrize("n,k,seed", [(2, 1, 1), (9, 2, 2), (33, 3, 3), (64, 2, 4), (130, 1, 5)])
def test_engine_matches_oracle(engine, oracle, n, k, seed):
    z, imgs, _ = synth.make_images(n, k, seed=seed)
    check(engine, oracle, z, imgs)


@pytest.mark.parametrize("kw", [dict(charge=-1, spin=2, task="omat"), dict(charge=2, spin=3, task="oc20"), dict(charge=0, spin=1, task="omc")])
def test_charge_spin_task(engine, oracle, kw):
    z, imgs, _ = synth.make_images(24, 1, seed=11)
    check(engine, oracle, z, imgs, **kw)


def test_heavier_elements(engine, oracle):
    z, imgs, _ = synth.make_images(28, 1, seed=12)
    z = z.copy()
    z[::3] = [26, 17, 15, 30, 35, 12, 29, 34, 11, 9][: len(z[::3])]
    check(engine, oracle, z, imgs)


@pytest.mark.parametrize("name", ["small_n20_k3", "small_n20_charged", "c1_n50_k8", "c2_n500_k2"])
def test_golden_fixtures(engine, name):
    g = load_golden(name)
    engine.set_system(g["z"], charge=int(g["charge"]), spin=int(g["spin"]), task=str(g["task"]))
    e, f = engine.energy_forces(g["pos"])
    assert np.abs(e - g["energy"]).max() <= TOL_E
    assert np.abs(f - g["forces"]).max() <= TOL_F


def test_c3_energy_golden(engine):
    """2000-atom images (BASELINE c3 size) against float64 oracle energies (tools/make_golden_c3.py).
    Guards the systematic part of the error: anything shared by all atoms (e.g. the system embedding)
    must be exact, or the bias grows linearly with N (it was -2e-4 eV before sys_emb moved to float64)."""
    g = load_golden("c3c4_n2000")                       # forces at this size: tests/test_gpu_baseline_sizes.py
    engine.set_system(g["z"])
    e, _ = engine.energy_forces(g["c3_pos"], forces=False)
    assert np.abs(e - g["c3_energy"]).max() <= TOL_E


@pytest.mark.parametrize("mode", ["fp32", "split", "split-bf16", "bf16x3"])
def test_precision_modes(weights, oracle, mode, monkeypatch):
    """UMX_PRECISION: fp32-MFMA everywhere, or split planes on the large SO(2)/radial GEMMs (LDS-DMA GEMM; forward: two fp16 activation
    planes x three exact fp16 weight planes, 4 products -- or three bf16 planes, 6 products; reverse: two bf16 planes, 3 products --
    or, bf16x3, three bf16 planes and 6 products in BOTH passes: 24-bit products everywhere, the like-for-like arithmetic to the
    reference's float32) -- every mode must hold the north-star tolerances."""
    from pdb2reaction_amd.engine import Engine

    monkeypatch.setenv("UMX_PRECISION", mode)
    eng = Engine(0)
    try:
        eng.load_weights(weights)
        z, imgs, _ = synth.make_images(150, 2, seed=13)
        check(eng, oracle, z, imgs)
    finally:
        eng.close()


@pytest.fixture(scope="module")
def switch_case(oracle):
    z, imgs, _ = synth.make_images(150, 2, seed=13)
    p32 = np.asarray(imgs, dtype=np.float32)
    ref = [oracle.energy_forces(z, p32[k].astype(np.float64)) for k in range(len(p32))]
    return z, p32, np.array([r[0] for r in ref]), np.stack([r[1] for r in ref])


@pytest.mark.parametrize("env", [
    {"UMX_ALT_ROWS": "0"}, {"UMX_ALT_ROWS": "0", "UMX_PRECISION": "split"},     # without the sign-alternating operand rows
    {"UMX_NODE_F64": "0"},                                                       # node-level linears on the fp32 MFMA
    {"UMX_STREAMS": "2"}, {"UMX_STREAMS": "2", "UMX_PRECISION": "split"}, {"UMX_STREAMS": "1"},
    {"UMX_FORCE_PARTS": "3"}, {"UMX_MAX_CHUNK_IMAGES": "1"}, {"UMX_WS_EAGER": "1"},
    {"UMX_LOW_SEP": "0"}, {"UMX_LOW_SEP": "1"}, {"UMX_LOW_SEP": "2"},           # the small plane products: one accumulator / apart for fc3 only / for every forward product
    {"UMX_LANES_AUTO_EDGES": "1000"},                                            # two lanes chosen by the engine from the batch's edge count
    {"UMX_ALIGN_PLANES": "0"}, {"UMX_ALIGN_PLANES": "1"},                        # leading planes: nearest bf16 (rounds 4-5) / aligned in every forward product (default 2: the plain ones)
], ids=lambda e: ",".join(f"{k[4:]}={v}" for k, v in e.items()))
def test_documented_switches_hold_the_tolerances(weights, switch_case, env, monkeypatch):
    """Every run-time switch of README.md that touches the evaluation (round 5 pruned the settled development levers: what is listed is what
    exists); each combination must stay inside the north-star tolerances."""
    from pdb2reaction_amd.engine import Engine

    for k, v in env.items():
        monkeypatch.setenv(k, v)
    z, p32, e_ref, f_ref = switch_case
    eng = Engine(0)
    try:
        eng.load_weights(weights)
        eng.set_system(z)
        e, f = eng.energy_forces(p32)
        assert np.abs(e - e_ref).max() <= TOL_E and np.abs(f - f_ref).max() <= TOL_F
    finally:
        eng.close()


@pytest.mark.parametrize("mode,wider", [("split", "split-bf16")])
def test_fp16_operand_range_is_guarded(weights, mode, wider, monkeypatch):
    """The mode with fp16 forward planes (split) keeps the forward GEMM operands as fp16 planes of 16 x (activation): an activation beyond +-4094 converts to
    inf, the GEMM output to NaN and the host-buffer entry refuses the result (UMX_ERR_RANGE) instead of returning it.  The binding
    then re-loads the SAME engine with bf16 forward planes (float32's range; still the HIP path) and evaluates again -- the answer
    is bitwise what an engine created in split-bf16 gives for the same (absurd) weights."""
    from pdb2reaction_amd.engine import Engine, UmxError, UMX_ERR_RANGE

    big = dict(weights)
    key = "blocks.0.edge_wise.so2_conv_1.rad_func.fc3"
    big[key + ".weight"] = (np.asarray(weights[key + ".weight"]) * 3e4).astype(np.float32)
    z, imgs, _ = synth.make_images(40, 1, seed=2)
    monkeypatch.setenv("UMX_PRECISION", mode)
    monkeypatch.setenv("UMX_NO_WIDEN", "1")
    eng = Engine(0)
    try:
        eng.load_weights(big)
        eng.set_system(z)
        with pytest.raises(UmxError, match="non-finite energy.*fp16 operand range") as ei:
            eng.energy_forces(imgs)
        assert ei.value.status == UMX_ERR_RANGE
    finally:
        eng.close()
    monkeypatch.delenv("UMX_NO_WIDEN")
    ref = Engine(0, precision=wider)
    eng = Engine(0)
    try:
        ref.load_weights(big)
        ref.set_system(z)
        e0, f0 = ref.energy_forces(imgs)
        assert np.isfinite(e0).all() and np.isfinite(f0).all()
        eng.load_weights(big)
        eng.set_system(z)
        with pytest.warns(RuntimeWarning, match=wider):
            e, f = eng.energy_forces(imgs)
        assert eng.widened and np.array_equal(e, e0) and np.array_equal(f, f0)
        e2, f2 = eng.energy_forces(imgs)                       # stays widened, no second warning path
        assert np.array_equal(e2, e0) and np.array_equal(f2, f0)
        nanw = dict(weights)
        nanw[key + ".bias"] = np.full_like(np.asarray(weights[key + ".bias"]), np.nan)
        with pytest.raises(UmxError, match="non-finite value in " + key.replace(".", r"\.") + r"\.bias"):
            ref.load_weights(nanw)                              # refused at load, not discovered as a NaN energy later
        ref.load_weights(big)
        ref.set_system(z)
        bad = imgs.copy(); bad[0, 0, 0] = np.nan
        with pytest.raises(UmxError, match="non-finite position"):
            eng.energy_forces(bad)                              # (a NaN coordinate would otherwise just lose its edges)
    finally:
        eng.close()
        ref.close()


def test_device_pointer_entry_reports_range_violations(weights, monkeypatch):
    """the asynchronous device-pointer entry cannot look at its own result, so an activation beyond the fp16
    operand range used to hand NaN energies / forces to torch-resident callers with status 0 and no way to find out.  Now the kernel
    that writes the energies sets a sticky device flag: umx_synchronize returns UMX_ERR_RANGE (and clears it), and so does the next
    evaluation on the engine; ShardedImageEvaluator (the multi-GPU string path) checks the gathered energies, widens the engine to
    bf16 forward planes -- every rank sees the same energies, so all ranks do -- and evaluates again."""
    import torch

    from pdb2reaction_amd.engine import Engine, UmxError, UMX_ERR_RANGE
    from pdb2reaction_amd.parallel import ShardedImageEvaluator

    big = dict(weights)
    key = "blocks.0.edge_wise.so2_conv_1.rad_func.fc3"
    big[key + ".weight"] = (np.asarray(weights[key + ".weight"]) * 3e4).astype(np.float32)
    z, imgs, _ = synth.make_images(40, 2, seed=2)
    dev = torch.device("cuda", 0)
    pos = torch.as_tensor(imgs, dtype=torch.float32, device=dev)
    e_t = torch.zeros(2, dtype=torch.float64, device=dev)
    f_t = torch.zeros(2, 40, 3, dtype=torch.float32, device=dev)
    ref = Engine(0, precision="split-bf16")
    eng = Engine(0, precision="split")                          # the fast mode: fp16 forward planes
    try:
        ref.load_weights(big); ref.set_system(z)
        e0, f0 = ref.energy_forces(imgs)
        eng.load_weights(big); eng.set_system(z)
        assert eng.precision_mode() == "split-f16"
        st = torch.cuda.current_stream().cuda_stream
        eng.energy_forces_dev(2, pos.data_ptr(), e_t.data_ptr(), f_t.data_ptr(), stream=st)       # status 0: nothing has been looked at
        assert eng.take_range_error() is True and not torch.isfinite(e_t).all()
        assert eng.take_range_error() is False                  # cleared by the report
        eng.energy_forces_dev(2, pos.data_ptr(), e_t.data_ptr(), f_t.data_ptr(), stream=st)
        with pytest.raises(UmxError, match="previous device-pointer evaluation produced a non-finite energy") as ei:
            eng.energy_forces_dev(2, pos.data_ptr(), e_t.data_ptr(), f_t.data_ptr(), stream=st)  # the NEXT call refuses to run
        assert ei.value.status == UMX_ERR_RANGE
        eng.energy_forces_dev(2, pos.data_ptr(), e_t.data_ptr(), f_t.data_ptr(), stream=st)       # ... once: the flag is cleared
        with pytest.raises(UmxError, match="device-pointer evaluation produced a non-finite energy"):
            eng.synchronize()
        eng.synchronize()
        bad = pos.clone()
        bad[1, 7, 2] = float("nan")                             # a NaN coordinate in a DEVICE buffer: refused by the call itself
        with pytest.raises(UmxError, match=r"non-finite position \(device buffer\)") as ei:
            eng.energy_forces_dev(2, bad.data_ptr(), e_t.data_ptr(), f_t.data_ptr(), stream=st)
        assert ei.value.status == -1 and eng.take_range_error() is False

        def local(c):
            eng.energy_forces_dev(c.shape[0], c.contiguous().data_ptr(), e_t.data_ptr(), f_t.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
            return e_t[: c.shape[0]], f_t[: c.shape[0]]

        ev = ShardedImageEvaluator(local, 2, 40, dev, engine=eng)
        with pytest.warns(RuntimeWarning, match="split-bf16"):
            e, f = ev(pos)
        assert eng.widened and eng.precision_mode() == "split-bf16"
        assert np.array_equal(e.cpu().numpy(), e0) and np.array_equal(f.cpu().numpy(), f0.astype(np.float64))
        monkeypatch.setenv("UMX_NO_WIDEN", "1")
        eng2 = Engine(0, precision="split")
        try:
            eng2.load_weights(big); eng2.set_system(z)

            def local2(c):
                eng2.energy_forces_dev(c.shape[0], c.contiguous().data_ptr(), e_t.data_ptr(), f_t.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
                return e_t[: c.shape[0]], f_t[: c.shape[0]]

            with pytest.raises(RuntimeError, match=r"non-finite energy for image\(s\) \[0, 1\]"):
                ShardedImageEvaluator(local2, 2, 40, dev, engine=eng2)(pos)
            assert eng2.take_range_error() is False             # the evaluator collected the flag before raising
        finally:
            eng2.close()
    finally:
        eng.close()
        ref.close()


def test_auto_precision_is_the_like_for_like_mode(weights, monkeypatch):
    """UMX_PRECISION=auto (the default) resolves to bf16x3 -- every product of both passes with >= 24 significant bits, the like-for-like
    arithmetic to the reference's float32 (uma_pysis.py:229,246-250) -- at every system size, bitwise what the explicitly chosen mode
    computes; an explicit mode is never changed behind the caller's back."""
    from pdb2reaction_amd.engine import Engine

    monkeypatch.delenv("UMX_PRECISION", raising=False)
    z_s, img_s, _ = synth.make_images(40, 2, seed=3)
    z_l, img_l, _ = synth.make_images(90, 2, seed=4)
    auto, x3, f16 = Engine(0), Engine(0, precision="bf16x3"), Engine(0, precision="split")
    try:
        for e in (auto, x3, f16):
            e.load_weights(weights)
        assert (auto.precision_mode(), x3.precision_mode(), f16.precision_mode()) == ("bf16x3", "bf16x3", "split-f16")
        for z, img in ((z_s, img_s), (z_l, img_l)):
            auto.set_system(z); x3.set_system(z); f16.set_system(z)
            assert auto.precision_mode() == "bf16x3" and f16.precision_mode() == "split-f16"
            ea, fa = auto.energy_forces(img)
            er, fr = x3.energy_forces(img)
            assert np.array_equal(ea, er) and np.array_equal(fa, fr)
        assert not auto.widened
        alias = Engine(0, precision="split-exact")
        try:
            alias.load_weights(weights)
            assert alias.precision_mode() == "bf16x3"
        finally:
            alias.close()
    finally:
        for e in (auto, x3, f16):
            e.close()


def test_single_image_beyond_the_workspace_names_the_way_out(weights):
    """One image whose edge pipeline does not fit the workspace budget cannot be evaluated by one engine (every per-edge activation
    of the four layers is kept for the reverse pass: ~120 KB per directed edge, i.e. ~25 000 atoms on a 288 GB part).  The status is
    UMX_ERR_CAPACITY and the text says what to do: the graph-parallel mode (uma_pysis(workers=<ranks>) under torch.distributed),
    which partitions the edges of the ONE image over several GPUs -- the reference's workers > 1 (uma_pysis.py:220-242)."""
    from pdb2reaction_amd.engine import Engine, UmxError

    z, imgs, _ = synth.make_images(300, 2, seed=4)
    eng = Engine(0)
    try:
        eng.load_weights(weights)
        eng.set_system(z)
        eng.set_workspace_limit(200 << 20)                      # 200 MiB: one 300-atom image needs ~2 GiB
        with pytest.raises(UmxError, match=r"one image \(300 atoms, \d+ directed edges\) needs \d+ MiB.*graph-parallel.*workers=.*16 partitions") as ei:
            eng.energy_forces(imgs)
        assert ei.value.status == -4
        eng.set_workspace_limit(0)                              # back to the automatic budget: the engine is usable again
        e, f = eng.energy_forces(imgs)
        assert np.isfinite(e).all() and np.isfinite(f).all()
    finally:
        eng.close()


@pytest.mark.parametrize("parts", [2, 3, 5])
def test_one_image_in_target_node_partitions(weights, oracle, parts, monkeypatch):
    """an image whose edge pipeline does not fit the workspace in one piece is evaluated in target-node partitions on the
    SAME GPU -- the graph-parallel plan for `parts` virtual ranks run one after another, own per-edge activations, ONE shared region for
    the GEMM operands, exchange points summed locally (`eval_partitioned`).  Forced here on small systems (UMX_FORCE_PARTS): results
    against the oracle at the BASELINE tolerances and against the ordinary path (float32 summation order), several images per call,
    energy-only calls, a partition without edges (dilute system, more partitions than bonded atoms)."""
    from pdb2reaction_amd.engine import Engine

    z, imgs, _ = synth.make_images(97, 3, seed=11)
    ref = Engine(0)
    monkeypatch.setenv("UMX_FORCE_PARTS", str(parts))
    eng = Engine(0)
    try:
        for e_ in (ref, eng):
            e_.load_weights(weights)
            e_.set_system(z)
        e0, f0 = ref.energy_forces(imgs)
        e, f = eng.energy_forces(imgs)
        assert ref.last_partitions() == 0 and eng.last_partitions() == parts and eng.graph_stats() == ref.graph_stats()
        assert np.abs(e - e0).max() <= 2e-5 and np.abs(f - f0).max() <= 2e-5
        for k in range(3):
            e_ref, f_ref = oracle.energy_forces(z, imgs[k].astype(np.float32).astype(np.float64))
            assert abs(e[k] - e_ref) <= TOL_E and np.abs(f[k] - f_ref).max() <= TOL_F
        e2, _ = eng.energy_forces(imgs, forces=False)
        assert np.array_equal(e2, e)
        e3, f3 = eng.energy_forces(imgs)                                    # deterministic (fixed summation order over the partitions)
        assert np.array_equal(e3, e) and np.array_equal(f3, f)
        zd = np.array([8, 1, 1, 6, 7], dtype=np.int32)                       # two bonded groups 30 A apart + one isolated atom
        pd = np.array([[[0, 0, 0], [0.96, 0, 0], [-0.3, 0.9, 0], [30, 0, 0], [60, 0, 0]]], np.float32)
        pd[0, 4] = [31.2, 0.4, 0.1]
        pd = np.concatenate([pd, [[[90.0, 0, 0]]]], axis=1)
        zd = np.concatenate([zd, [1]]).astype(np.int32)
        for e_ in (ref, eng):
            e_.set_system(zd)
        ed0, fd0 = ref.energy_forces(pd)
        ed, fd = eng.energy_forces(pd)
        assert abs(ed[0] - ed0[0]) <= 2e-5 and np.abs(fd - fd0).max() <= 2e-5 and np.all(fd[0, 5] == 0.0)
    finally:
        ref.close()
        eng.close()


def test_oversized_image_falls_back_to_partitions(weights, oracle):
    """The automatic route: with a workspace budget that holds the image only in pieces the engine partitions it by itself."""
    from pdb2reaction_amd.engine import Engine

    z, imgs, _ = synth.make_images(300, 1, seed=4)
    eng = Engine(0)
    try:
        eng.load_weights(weights)
        eng.set_system(z)
        e0, f0 = eng.energy_forces(imgs)
        assert eng.last_partitions() == 0
        eng.set_workspace_limit(1800 << 20)                     # the whole image needs ~2.2 GiB; its persistent part ~1.4 GiB
        e, f = eng.energy_forces(imgs)
        assert 2 <= eng.last_partitions() <= 16
        assert np.abs(e - e0).max() <= 2e-5 and np.abs(f - f0).max() <= 2e-5
        e_ref, f_ref = oracle.energy_forces(z, imgs[0].astype(np.float32).astype(np.float64))
        assert abs(e[0] - e_ref) <= TOL_E and np.abs(f[0] - f_ref).max() <= TOL_F
        eng.set_workspace_limit(0)
        e2, f2 = eng.energy_forces(imgs)                        # back on the ordinary path, bitwise as before
        assert eng.last_partitions() == 0 and np.array_equal(e2, e0) and np.array_equal(f2, f0)
    finally:
        eng.close()


def test_no_edges_and_isolated_atoms(engine, oracle):
    """Empty / ragged graphs: a lone atom, two atoms beyond the cutoff, one isolated atom next to a cluster."""
    z = np.array([8], dtype=np.int32)
    engine.set_system(z)
    e, f = engine.energy_forces(np.zeros((1, 1, 3), np.float32))
    e_ref, _ = oracle.energy_forces(z, np.zeros((1, 3)))
    assert engine.graph_stats() == (0, 0) and abs(e[0] - e_ref) <= TOL_E and np.all(f == 0)
    z = np.array([1, 6], dtype=np.int32)
    far = np.array([[[0, 0, 0], [0, 0, 7.5]]], np.float32)
    engine.set_system(z)
    e, f = engine.energy_forces(far)
    assert engine.graph_stats()[0] == 0 and np.all(f == 0)
    assert abs(e[0] - oracle.energy_forces(z, far[0].astype(np.float64))[0]) <= TOL_E
    zc, pc = synth.make_cluster(12, seed=9)
    z = np.concatenate([zc, [7]]).astype(np.int32)
    p = np.concatenate([pc, [[40.0, 0, 0]]])[None]
    check(engine, oracle, z, p)


def test_pole_aligned_edges(engine, oracle):
    """Edges exactly along +y / -y exercise the detached-pole branch and the flipped frame."""
    z = np.array([6, 8, 1, 7], dtype=np.int32)
    p = np.array([[[0, 0, 0], [0, 1.4, 0], [0, -1.1, 0], [1.2, 0.3, -0.4]]], np.float64)
    check(engine, oracle, z, p)


@pytest.mark.parametrize("max_neigh", [3, 8, 20])
def test_max_neigh_truncation(engine, weights, max_neigh):
    """`max_neigh` keeps the nearest M sources per target (fairchem's per-centre cap, reference uma_pysis.py:301-318):
    the graph becomes asymmetric, so this also exercises the CSR-by-source reverse pass."""
    import torch
    from oracle.escn_md_oracle import Oracle, radius_graph

    z, imgs, _ = synth.make_images(40, 2, seed=3)
    p32 = imgs.astype(np.float32)
    orc = Oracle(weights, max_neigh=max_neigh)
    engine.set_system(z, max_neigh=max_neigh)
    engine.debug_keep(True)
    try:
        e, f = engine.energy_forces(p32[1:2])
        src, dst = radius_graph(torch.as_tensor(p32[1].astype(np.float64)), W.CUTOFF, max_neigh)
        assert np.array_equal(engine.debug_fetch("src", np.int32), src.numpy()) and np.array_equal(engine.debug_fetch("dst", np.int32), dst.numpy())
        assert engine.graph_stats()[1] == max_neigh
    finally:
        engine.debug_keep(False)
    e, f = engine.energy_forces(p32)
    for k in range(2):
        e_ref, f_ref = orc.energy_forces(z, p32[k].astype(np.float64))
        assert abs(e[k] - e_ref) <= TOL_E and np.abs(f[k] - f_ref).max() <= TOL_F
    engine.set_system(z, radius=3.0)
    e, _ = engine.energy_forces(imgs)
    assert np.isfinite(e).all()


def test_batched_equals_single_and_chunked(engine):
    """Images are independent units: batched, one-by-one and chunked evaluation agree bit for bit."""
    z, imgs, _ = synth.make_images(70, 5, seed=6)
    engine.set_system(z)
    e, f = engine.energy_forces(imgs)
    for k in range(5):
        e1, f1 = engine.energy_forces(imgs[k])
        assert e1[0] == e[k] and np.array_equal(f1[0], f[k])
    os.environ["UMX_MAX_CHUNK_IMAGES"] = "2"
    try:
        e2, f2 = engine.energy_forces(imgs)
    finally:
        del os.environ["UMX_MAX_CHUNK_IMAGES"]
    assert np.array_equal(e2, e) and np.array_equal(f2, f)
    e3, _ = engine.energy_forces(imgs, forces=False)
    assert np.array_equal(e3, e)


def test_deterministic(engine):
    z, imgs, _ = synth.make_images(90, 2, seed=8)
    engine.set_system(z)
    a = engine.energy_forces(imgs)
    b = engine.energy_forces(imgs)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])     # segmented sums, no float atomics


def test_device_pointer_entry(engine):
    z, imgs, _ = synth.make_images(30, 3, seed=2)
    engine.set_system(z)
    e, f = engine.energy_forces(imgs)
    dev = torch.device("cuda", 0)
    pos = torch.as_tensor(imgs, dtype=torch.float32, device=dev).contiguous()
    ed = torch.zeros(3, dtype=torch.float64, device=dev)
    fd = torch.zeros(3, 30, 3, dtype=torch.float32, device=dev)
    engine.energy_forces_dev(3, pos.data_ptr(), ed.data_ptr(), fd.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(ed.cpu().numpy(), e) and np.array_equal(fd.cpu().numpy(), f)


def test_device_pointer_entry_is_stream_ordered(engine):
    """The dev entry must be ordered with the CALLER's stream and nothing else (the work used to run on a private
    non-blocking stream when the handle was 0, so torch consumers read the output while it was still being computed).
    No device-wide synchronisation here: results are consumed by torch ops on the same stream, positions are overwritten
    right after the call, and the only host wait is the .cpu() of the CONSUMER's output (which waits for that stream only)."""
    n, k = 400, 4                      # large enough that the evaluation takes several ms
    z, imgs, _ = synth.make_images(n, k, seed=3)
    engine.set_system(z)
    e_ref, f_ref = engine.energy_forces(imgs)
    imgs2 = imgs[::-1] + np.random.default_rng(9).normal(0.0, 0.05, imgs.shape)   # same atoms, other geometries
    e_ref2, f_ref2 = engine.energy_forces(imgs2)
    dev = torch.device("cuda", 0)
    host_a = torch.as_tensor(imgs, dtype=torch.float32).pin_memory()
    host_b = torch.as_tensor(imgs2, dtype=torch.float32).pin_memory()
    for stream in (torch.cuda.current_stream(), torch.cuda.Stream()):
        with torch.cuda.stream(stream):
            pos = torch.empty(k, n, 3, dtype=torch.float32, device=dev)
            ed = torch.full((k,), float("nan"), dtype=torch.float64, device=dev)
            fd = torch.full((k, n, 3), float("nan"), dtype=torch.float32, device=dev)
            pos.copy_(host_a, non_blocking=True)                                  # producer on the caller's stream
            engine.energy_forces_dev(k, pos.data_ptr(), ed.data_ptr(), fd.data_ptr(), stream=stream.cuda_stream)
            e1, f1 = ed.clone(), fd.to(torch.float64)                             # consumers on the caller's stream
            pos.copy_(host_b, non_blocking=True)                                  # overwrite the input right away
            engine.energy_forces_dev(k, pos.data_ptr(), ed.data_ptr(), fd.data_ptr(), stream=stream.cuda_stream)
            e2, f2 = ed.clone(), fd.to(torch.float64)
            out = [t.cpu().numpy() for t in (e1, f1, e2, f2)]
        assert np.array_equal(out[0], e_ref) and np.array_equal(out[1], f_ref.astype(np.float64))
        assert np.array_equal(out[2], e_ref2) and np.array_equal(out[3], f_ref2.astype(np.float64))


@pytest.mark.parametrize("mode", ["fp32", "split", "split-bf16", "bf16x3"])
def test_stage_by_stage_against_staged_oracle(weights, mode, monkeypatch):
    """Every intermediate of the forward AND of the analytic reverse pass vs oracle/staged.py, in both precision modes
    (the split-bf16 path keeps its GEMM operands as bf16 planes, so fewer fp32 intermediates exist there)."""
    from oracle.staged import Staged
    from pdb2reaction_amd.engine import Engine

    z, pos = synth.make_cluster(26, seed=4)
    p32 = pos.astype(np.float32)
    st = Staged(weights)
    st.forward(z, p32.astype(np.float64))
    st.backward()
    t = {k: v.numpy() for k, v in st.t.items() if torch.is_tensor(v)}
    ne = len(t["src"])
    monkeypatch.setenv("UMX_PRECISION", mode)
    engine = Engine(0)
    try:
        engine.load_weights(weights)
        engine.set_system(z)
        engine.debug_keep(True)
        engine.energy_forces(p32)
        assert np.array_equal(engine.debug_fetch("src", np.int32), t["src"])
        assert np.array_equal(engine.debug_fetch("dst", np.int32), t["dst"])
        out_ptr, out_edge = engine.debug_fetch("out_ptr", np.int32), engine.debug_fetch("out_edge", np.int32)
        assert out_ptr[-1] == ne and np.array_equal(np.sort(out_edge), np.arange(ne))            # CSR by source is a permutation
        for n in range(len(z)):
            row = out_edge[out_ptr[n]:out_ptr[n + 1]]
            assert np.all(t["src"][row] == n) and np.all(np.diff(row) > 0)                      # right rows, deterministic order
        names = ["x0", "rad.deg", "e_node", "g_xfinal", "dedd"]
        per_layer = ["xn", "rad", "msg", "xmid", "xn2", "gspre", "ffh", "x", "g_xmid", "g_hid", "g_xn", "g_xin"]
        if mode == "fp32":          # split mode keeps these in PL planes / registers (k_modrot_bwd_pl never writes g_xrot)
            per_layer += ["xrot", "hid", "g_msg", "g_rad", "g_xrot"]
        for i in range(W.NUM_LAYERS):
            names += [f"{s}.{i}" for s in per_layer]
        # reverse pass: 16-bit products (2 x 2 bf16 planes) ~1e-5 relative per GEMM; bf16x3 (3 x 3 planes, 24-bit products) is held to
        # the fp32 mode's bound
        tol = 2e-5 if mode in ("fp32", "bf16x3") else 1e-4
        for nm in names:
            a = engine.debug_fetch(nm)
            r = t[nm].reshape(-1)
            assert a.size == r.size, nm
            assert np.abs(a - r).max() <= tol * max(np.abs(r).max(), 1.0), nm
        # the FORWARD tensors alone, at float32 level in every mode: the split modes' forward products are fp32-equivalent
        # (4 fp16 / 6 bf16 plane products); only the reverse pass runs on 16-bit products
        fwd = ["x0", "rad.deg", "e_node"] + [f"{s}.{i}" for i in range(W.NUM_LAYERS) for s in ("xn", "rad", "msg", "xmid", "xn2", "gspre", "ffh", "x")]
        worst = max(np.abs(engine.debug_fetch(nm) - t[nm].reshape(-1)).max() / max(np.abs(t[nm]).max(), 1.0) for nm in fwd)
        print(f"[stages {mode}] worst forward relative error {worst:.2e}")
        assert worst <= 3e-6, (mode, worst)              # measured: fp32 1.3e-6, split 7.9e-7, split-bf16 1.1e-6
        tau = engine.debug_fetch("tau").reshape(ne, 4)[:, :3]
        assert np.abs(tau - t["tau"]).max() <= tol and np.abs(tau[:, 1]).max() <= tol     # gauge: no torque about the edge
    finally:
        engine.close()


# ---- BASELINE sizes: size-independent properties (the oracle is too slow there) -----------------
@pytest.fixture(scope="module")
def c3(engine):
    z, imgs, frozen = synth.make_images(2000, 2)
    engine.set_system(z)
    e, f = engine.energy_forces(imgs)
    return z, imgs, e, f


def pole_atoms(pos, cutoff=W.CUTOFF, tol=2e-5):
    """Atoms on an edge whose direction is within the reference's isclose(nhat_y, 1) pole mask (rtol 1e-5):
    fairchem detaches the frame gradient there, so forces on those atoms are not rotation covariant."""
    p = np.asarray(pos, dtype=np.float64)
    bad = np.zeros(len(p), dtype=bool)
    for s0 in range(0, len(p), 500):
        d = p[None, :, :] - p[s0:s0 + 500, None, :]
        r = np.linalg.norm(d, axis=-1)
        m = (r > 0) & (r <= cutoff + 1e-3) & (np.abs(d[..., 1] / np.maximum(r, 1e-30) - 1.0) <= tol)
        i, j = np.nonzero(m)
        bad[i + s0] = True
        bad[j] = True
    return bad


def test_c3_newton_third_law(c3):
    _, _, e, f = c3
    assert np.isfinite(e).all() and np.isfinite(f).all()
    assert np.abs(f.astype(np.float64).sum(axis=1)).max() <= 5e-4        # sum of 2000 float32 forces


def test_c3_rotation_translation_invariance(engine, c3):
    from scipy.spatial.transform import Rotation

    z, imgs, e, f = c3
    rm = Rotation.random(random_state=3).as_matrix()
    rot = imgs @ rm.T + np.array([1.0, -2.0, 0.5])
    e2, f2 = engine.energy_forces(rot)
    assert np.abs(e2 - e).max() <= 2e-3                                   # fp32 round-off of rotated float32 inputs
    diff = np.abs(f2 - f @ rm.T.astype(np.float32)).max(axis=2)
    for k in range(len(imgs)):
        ok = ~(pole_atoms(imgs[k]) | pole_atoms(rot[k]))
        assert ok.sum() >= len(z) - 40
        assert diff[k][ok].max() <= TOL_F


def test_c3_permutation_invariance(engine, c3):
    z, imgs, e, f = c3
    perm = np.random.default_rng(0).permutation(len(z))
    engine.set_system(z[perm])
    e2, f2 = engine.energy_forces(imgs[:, perm])
    engine.set_system(z)
    assert np.abs(e2 - e).max() <= 1e-3
    assert np.abs(f2 - f[:, perm]).max() <= TOL_F                          # same frames, different summation order


def test_c3_forces_are_energy_gradient(engine, c3):
    z, imgs, e, f = c3
    h = 2e-2
    rng = np.random.default_rng(1)
    d = rng.standard_normal(imgs[0].shape)
    d /= np.linalg.norm(d)
    ep, _ = engine.energy_forces(imgs[0] + h * d, forces=False)
    em, _ = engine.energy_forces(imgs[0] - h * d, forces=False)
    fd = -(ep[0] - em[0]) / (2 * h)
    assert abs(fd - float((f[0].astype(np.float64) * d).sum())) <= 2e-2 * max(1.0, abs(fd))


def test_c5_size_invariants(engine):
    """Largest BASELINE config (c5: ~20 000 atoms, 1.6 M directed edges per image): runs inside the HBM workspace budget,
    chunked == single bit for bit, Newton's third law."""
    z, imgs, _ = synth.make_images(20000, 2)
    engine.set_system(z)
    e1, f1 = engine.energy_forces(imgs[:1])
    ne, maxdeg = engine.graph_stats()
    assert ne > 1_500_000 and maxdeg < 300
    assert np.isfinite(e1).all() and np.isfinite(f1).all()
    assert np.abs(f1.astype(np.float64).sum(axis=1)).max() <= 2e-3
    e2, f2 = engine.energy_forces(imgs)
    assert e2[0] == e1[0] and np.array_equal(f2[0], f1[0])


def test_two_lane_execution_is_bitwise_identical(weights, monkeypatch):
    """UMX_STREAMS=2 (two chunks in flight, matrix segments alternating through an event token, capped grids of the grid-stride
    stream kernels) must not change a single bit: same kernels, same per-item arithmetic, only the issue order differs -- with the two
    lanes really overlapping and with their segments issued on one stream (UMX_LANES_ONE_STREAM=1).
    Round 3: this test failed about once in ten suite runs (forces of one image 1e-6...1e-4 eV/A off).  Cause: the packed-fp32
    instructions hipcc's SLP vectoriser emits (v_pk_mul/add/fma_f32) are timing-sensitive on gfx950 -- beside other kernels on the same
    SIMDs single waves come out 0.1-1 % off (csrc/norm_bwd_repro.hip); the library is built with -fno-slp-vectorize since
    (build.py, NOTES.md section 5 item 14).  test_library_has_no_packed_fp32 keeps it that way."""
    from pdb2reaction_amd.engine import Engine

    z, imgs, _ = synth.make_images(260, 5, seed=21)
    res = {}
    for lanes, cap, one_stream in (("1", "512", False), ("2", "512", False), ("2", "512", True)):
        monkeypatch.setenv("UMX_STREAMS", lanes)
        if one_stream:
            monkeypatch.setenv("UMX_LANES_ONE_STREAM", "1")
        else:
            monkeypatch.delenv("UMX_LANES_ONE_STREAM", raising=False)
        eng = Engine(0)
        try:
            eng.load_weights(weights)
            eng.set_system(z)
            res[(lanes, cap, one_stream)] = eng.energy_forces(imgs)
        finally:
            eng.close()
    e0, f0 = res[("1", "512", False)]
    for key, (e, f) in res.items():
        assert np.array_equal(e, e0) and np.array_equal(f, f0), key


def test_reserve_images_allocates_the_workspace_once(weights):
    """umx_reserve_images (ABI v8): with the hint the first evaluation sizes the workspace for the announced batch, and batches up to that
    size no longer re-allocate it (device memory in use stays put); without it every larger batch grows it.  Results are unaffected."""
    import torch
    from pdb2reaction_amd.engine import Engine

    z, imgs, _ = synth.make_images(150, 6, seed=4)

    def used():
        torch.cuda.synchronize()
        fr, tot = torch.cuda.mem_get_info(0)
        return tot - fr

    plain, hinted = Engine(0), Engine(0)
    try:
        for e_ in (plain, hinted):
            e_.load_weights(weights)
            e_.set_system(z)
        hinted.reserve_images(6)
        base = used()
        e2, f2 = hinted.energy_forces(imgs[:2])
        after_first = used()
        e4, f4 = hinted.energy_forces(imgs[:4])
        e6, f6 = hinted.energy_forces(imgs)
        ws_bytes, n_alloc = hinted.workspace_stats()
        assert n_alloc == 1 and ws_bytes > 0                    # ONE allocation served the 2-, 4- and 6-image batches
        assert used() - after_first < 0.05 * ws_bytes           # only the small per-batch I/O buffers grew
        assert after_first - base >= ws_bytes
        p2, _ = plain.energy_forces(imgs[:2])
        mid = used()
        p6, pf6 = plain.energy_forces(imgs)
        assert plain.workspace_stats()[1] == 2 and plain.workspace_stats()[0] <= ws_bytes     # without the hint: one allocation per growth
        assert np.array_equal(p6, e6) and np.array_equal(pf6, f6) and np.array_equal(p2, e2) and np.array_equal(e4, e6[:4])
        hinted.reserve_images(0)
        with pytest.raises(Exception):
            hinted.reserve_images(-1)
    finally:
        plain.close()
        hinted.close()
