"""GPU parity of the MODEL VARIANTS SURVEY.md lists as possible for the real checkpoint (section 2.4 K8 "grid or spectral", Appendix A
header ``ff_type=grid|spectral (unsure)``, A.5 charge / spin embedding): the grid feed-forward, the pos_emb / lin_emb charge-spin
embeddings, a dataset embedding in the checkpoint's own order or absent -- each through the C ABI against the float64 oracle
(autograd) and, for the grid block, stage by stage against oracle/staged.py's hand-derived reverse.  [3P-UNVERIFIED]: like the rest of the
model oracle these forms are restated from SURVEY.md Appendix A, not pinned to fairchem.

Tolerances are the north-star's: |dE| <= 1e-4 eV, max|dF| <= 1e-3 eV/A (BASELINE.json)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from pdb2reaction_amd import synth, weights as W

pytestmark = pytest.mark.gpu

TOL_E = 1e-4   # eV
TOL_F = 1e-3   # eV/Angstrom

VARIANTS = {
    "grid": dict(ff_type="grid"),
    "grid_bias": dict(ff_type="grid", grid_bias=True),
    "pos_emb": dict(chg_spin_emb_type="pos_emb"),
    "lin_emb": dict(chg_spin_emb_type="lin_emb"),
    "grid_pos_emb_own_datasets": dict(ff_type="grid", chg_spin_emb_type="pos_emb", dataset_list=("omol", "omat", "oc20")),
    "no_dataset_embedding": dict(dataset_list=()),
}


def make(variant, mode, monkeypatch):
    from oracle.escn_md_oracle import Oracle
    from pdb2reaction_amd.engine import Engine

    w = W.make_synthetic_weights(0, **VARIANTS[variant])
    monkeypatch.setenv("UMX_PRECISION", mode)
    eng = Engine(0)
    eng.load_weights(w)
    return w, eng, Oracle(w)


@pytest.mark.parametrize("mode", ["bf16x3", "fp32"])
@pytest.mark.parametrize("variant", sorted(VARIANTS))
def test_variant_matches_oracle(variant, mode, monkeypatch):
    w, eng, orc = make(variant, mode, monkeypatch)
    try:
        v = W.variant_of(w)
        assert eng.model_variant() == (f"ff={'grid(G=42)' if v['ff_type'] == 'grid' else 'spectral'};emb={v['chg_spin_emb_type']};datasets={v['n_datasets']}")
        tasks = list(eng.dataset_list)
        for n, k, seed, kw in ((33, 2, 3, dict(charge=0, spin=1)), (64, 1, 4, dict(charge=-1, spin=2)), (130, 1, 5, dict(charge=2, spin=0))):
            kw["task"] = tasks[seed % len(tasks)] if v["n_datasets"] else "omol"
            z, imgs, _ = synth.make_images(n, k, seed=seed)
            p32 = imgs.astype(np.float32)
            eng.set_system(z, **kw)
            e, f = eng.energy_forces(p32)
            for i in range(k):
                e_ref, f_ref = orc.energy_forces(z, p32[i].astype(np.float64), **kw)
                assert abs(e[i] - e_ref) <= TOL_E, (variant, mode, n, e[i], e_ref)
                assert np.abs(f[i] - f_ref).max() <= TOL_F, (variant, mode, n)
    finally:
        eng.close()


def test_task_names_follow_the_blob_s_dataset_list(monkeypatch):
    """A checkpoint whose dataset_list has another order is an index remap, not a refusal: the same embedding ROW must be picked by NAME."""
    from pdb2reaction_amd.engine import Engine

    z, imgs, _ = synth.make_images(20, 1, seed=9)
    w_a = W.make_synthetic_weights(0)
    order = ("omc", "omol", "oc20", "odac", "omat")
    w_b = W.WeightSet(w_a, meta={"model": {"dataset_list": list(order)}})
    w_b["dataset_embedding.weight"] = np.stack([w_a["dataset_embedding.weight"][W.DATASET_LIST.index(t)] for t in order])
    out = []
    for w in (w_a, w_b):
        eng = Engine(0)
        try:
            eng.load_weights(W.pack_blob(w))                       # through the blob: the trailer carries the list
            assert eng.dataset_list == (tuple(order) if w is w_b else tuple(W.DATASET_LIST))
            eng.set_system(z, task="omat")
            out.append(eng.energy_forces(imgs))
            with pytest.raises(ValueError):
                eng.set_system(z, task="not-a-task")
        finally:
            eng.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    # a dataset table whose row count is not the list's length cannot be mapped by name: refused at load, not silently mis-indexed
    w_c = W.WeightSet(w_a, meta={"model": {"dataset_list": ["omol", "omat", "oc20"]}})          # 5 rows, 3 names
    eng = Engine(0)
    try:
        with pytest.raises(Exception, match="dataset"):
            eng.load_weights(W.pack_blob(w_c))
    finally:
        eng.close()


@pytest.mark.parametrize("mode", ["bf16x3", "fp32"])
def test_grid_block_stage_by_stage(mode, monkeypatch):
    """The grid feed-forward's forward activations (both hidden pre-activations on the 42-point grid, the block output) and the
    gradients that leave its hand-derived reverse (g_xmid, and everything downstream) vs oracle/staged.py."""
    from oracle.staged import Staged

    w, eng, _ = make("grid_bias", mode, monkeypatch)
    try:
        z, pos = synth.make_cluster(26, seed=4)
        p32 = pos.astype(np.float32)
        st = Staged(w)
        st.forward(z, p32.astype(np.float64))
        st.backward()
        t = {k: v.numpy() for k, v in st.t.items() if torch.is_tensor(v)}
        eng.set_system(z)
        eng.debug_keep(True)
        eng.energy_forces(p32)
        names = ["x0", "e_node", "g_xfinal", "dedd"]
        for i in range(W.NUM_LAYERS):
            names += [f"{s}.{i}" for s in ("xn", "msg", "xmid", "xn2", "ffg1", "ffg2", "x", "g_xmid", "g_xn", "g_xin")]
        for nm in names:
            a, r = eng.debug_fetch(nm), t[nm].reshape(-1)
            assert a.size == r.size, nm
            assert np.abs(a - r).max() <= 2e-5 * max(np.abs(r).max(), 1.0), nm
        fwd = [f"{s}.{i}" for i in range(W.NUM_LAYERS) for s in ("xn2", "ffg1", "ffg2", "x")]
        worst = max(np.abs(eng.debug_fetch(nm) - t[nm].reshape(-1)).max() / max(np.abs(t[nm]).max(), 1.0) for nm in fwd)
        print(f"[grid stages {mode}] worst forward relative error {worst:.2e}")
        assert worst <= 3e-6, worst
    finally:
        eng.close()


@pytest.mark.parametrize("name", ["small_n20_k3_grid", "small_n20_charged_grid_pos_emb", "c2_n500_k1_grid", "c3_n2000_k1_grid"])
def test_golden_fixtures_of_the_variants(name, monkeypatch):
    """... up to the headline size: a 2000-atom image with the grid feed-forward (84 000 grid rows per layer, every atom's residual stream
    through the float64-accumulated grid MLP) must hold 1e-4 eV / 1e-3 eV/A like the spectral form does."""
    g = load_golden(name)
    kw = {k[len("variant_"):]: (tuple(str(x) for x in g[k]) if k.endswith("dataset_list") else g[k].item()) for k in g if k.startswith("variant_")}
    from pdb2reaction_amd.engine import Engine

    eng = Engine(0)
    try:
        eng.load_weights(W.make_synthetic_weights(int(g["weights_seed"]), **kw))
        eng.set_system(g["z"], charge=int(g["charge"]), spin=int(g["spin"]), task=str(g["task"]))
        e, f = eng.energy_forces(g["pos"])
        print(f"[{name}] max |dE| = {np.abs(e - g['energy']).max():.2e} eV, max |dF| = {np.abs(f - g['forces']).max():.2e} eV/A")
        assert np.abs(e - g["energy"]).max() <= TOL_E
        assert np.abs(f - g["forces"]).max() <= TOL_F
    finally:
        eng.close()


def test_grid_variant_batched_equals_single_and_is_reproducible(monkeypatch):
    w, eng, _ = make("grid", "bf16x3", monkeypatch)
    try:
        z, imgs, _ = synth.make_images(40, 3, seed=21)
        eng.set_system(z)
        e, f = eng.energy_forces(imgs)
        e2, f2 = eng.energy_forces(imgs)
        assert np.array_equal(e, e2) and np.array_equal(f, f2)
        for k in range(3):
            ek, fk = eng.energy_forces(imgs[k])
            assert np.array_equal(ek[0], e[k]) and np.array_equal(fk[0], f[k])
        assert np.abs(f.astype(np.float64).sum(axis=1)).max() <= 1e-4
    finally:
        eng.close()
