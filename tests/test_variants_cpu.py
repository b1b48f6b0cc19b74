"""CPU checks of the model variants (``ff_type = grid | spectral``, the charge / spin embedding
forms, the dataset list): the oracle's three forms agree (autograd == hand-derived staged reverse == chunked), invariants that survive a
grid non-linearity hold, the weight-set plumbing round-trips, and the committed variant fixtures are what the oracle gives."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from pdb2reaction_amd import synth, weights as W
from oracle.chunked import ChunkedForces
from oracle.escn_md_oracle import Oracle
from oracle.staged import Staged

VARIANTS = [dict(ff_type="grid"), dict(ff_type="grid", grid_bias=True), dict(chg_spin_emb_type="pos_emb"), dict(chg_spin_emb_type="lin_emb"),
            dict(ff_type="grid", chg_spin_emb_type="pos_emb", dataset_list=("omol", "omat", "oc20")), dict(dataset_list=())]


def test_default_weight_set_is_unchanged_by_the_variant_plumbing():
    """The default arguments still give the weight set every golden fixture of rounds 1-4 was generated with."""
    import hashlib

    w = W.make_synthetic_weights(0)
    h = hashlib.sha256()
    for k, v in w.items():
        h.update(k.encode()); h.update(v.tobytes())
    assert h.hexdigest()[:16] == "942a5fa82e349d17" and len(w) == 140
    assert W.variant_of(w) == {"ff_type": "spectral", "chg_spin_emb_type": "rand_emb", "n_datasets": 5, "grid_points": 0, "grid_bias": False}
    g = W.make_synthetic_weights(0, ff_type="grid")
    shared = [k for k in w if k in g]
    assert len(shared) == len(w) - 6 * W.NUM_LAYERS and all(np.array_equal(w[k], g[k]) for k in shared)     # only the feed-forward differs


@pytest.mark.parametrize("kw", VARIANTS)
def test_oracle_forms_agree_and_blob_round_trips(kw):
    w = W.make_synthetic_weights(0, **kw)
    back = W.unpack_blob(W.pack_blob(w))
    assert list(back) == list(w) and all(np.array_equal(back[k], w[k]) for k in w) and back.meta == w.meta
    assert W.blob_meta(W.pack_blob(w)) == w.meta
    z, imgs, _ = synth.make_images(22, 1, seed=5)
    pos = imgs[0]
    task = "omat" if "omat" in (kw.get("dataset_list") or W.DATASET_LIST) else "omol"
    e, f = Oracle(w).energy_forces(z, pos, charge=-1, spin=2, task=task)
    st = Staged(w)
    st.forward(z, pos, charge=-1, spin=2, task=task)
    g = st.backward().numpy()
    rmsd = float(w["normalizer.rmsd"][0])
    assert np.abs(-g * rmsd - f).max() <= 1e-12                                 # hand-derived reverse (incl. the grid block) == autograd
    e3, f3 = ChunkedForces(w, chunk=97).energy_forces(z, pos, charge=-1, spin=2, task=task)
    assert abs(e3 - e) <= 1e-10 and np.abs(f3 - f).max() <= 1e-12
    assert np.abs(f.sum(0)).max() <= 1e-10                                      # translation invariance survives every variant
    perm = np.random.default_rng(0).permutation(len(z))
    e_p, f_p = Oracle(w).energy_forces(z[perm], pos[perm], charge=-1, spin=2, task=task)
    assert abs(e_p - e) <= 1e-9 and np.abs(f_p - f[perm]).max() <= 1e-10
    # F = -dE/dx by central differences
    o = Oracle(w)
    for a, c in ((3, 0), (11, 2)):
        h = 1e-5
        pp, pm = pos.copy(), pos.copy()
        pp[a, c] += h; pm[a, c] -= h
        fd = -(o.energy_forces(z, pp, charge=-1, spin=2, task=task, forces=False)[0] - o.energy_forces(z, pm, charge=-1, spin=2, task=task, forces=False)[0]) / (2 * h)
        assert abs(fd - f[a, c]) <= 2e-6 * max(1.0, abs(f[a, c]))


def test_grid_matrices_are_data_and_their_round_trip_is_the_identity():
    tg, fg = W.synthetic_grid_matrices()
    assert tg.shape == fg.shape == (42, 9)
    np.testing.assert_allclose(fg.astype(np.float64).T @ tg.astype(np.float64), np.eye(9), atol=2e-6)
    # a linear "MLP" makes the grid block the identity map times the product of the three matrices: to-grid / from-grid carry no hidden scale
    w = W.make_synthetic_weights(0, ff_type="grid")
    x = torch.as_tensor(np.random.default_rng(1).standard_normal((5, 9, 128)))
    p = {k: torch.as_tensor(np.asarray(v), dtype=torch.float64) for k, v in w.items()}
    back = torch.einsum("gi,ngc->nic", p["so3_grid.from_grid_mat"], torch.einsum("gi,nic->ngc", p["so3_grid.to_grid_mat"], x))
    assert (back - x).abs().max() <= 1e-5


def test_embedding_forms():
    z, imgs, _ = synth.make_images(8, 1, seed=2)
    for emb in ("pos_emb", "lin_emb"):
        o = Oracle(W.make_synthetic_weights(0, chg_spin_emb_type=emb))
        a = o.system_embedding(0, 1, "omol")
        assert a.shape == (128,) and not torch.allclose(a, o.system_embedding(1, 1, "omol")) and not torch.allclose(a, o.system_embedding(0, 0, "omol"))
    o = Oracle(W.make_synthetic_weights(0, chg_spin_emb_type="pos_emb"))
    wv = o.p["spin_embedding.W"]
    assert torch.equal(o.charge_spin_embedding("spin", 0), torch.zeros(128, dtype=torch.float64))          # the null spin embeds to zero
    torch.testing.assert_close(o.charge_spin_embedding("spin", 3), torch.cat([torch.sin(6 * np.pi * wv), torch.cos(6 * np.pi * wv)]))
    # another dataset order = another row, same answer by NAME
    w_a = W.make_synthetic_weights(0)
    order = ("omc", "omol", "oc20", "odac", "omat")
    w_b = W.WeightSet(w_a, meta={"model": {"dataset_list": list(order)}})
    w_b["dataset_embedding.weight"] = np.stack([w_a["dataset_embedding.weight"][W.DATASET_LIST.index(t)] for t in order])
    assert torch.equal(Oracle(w_a).system_embedding(0, 1, "odac"), Oracle(w_b).system_embedding(0, 1, "odac"))


@pytest.mark.parametrize("name", ["small_n20_k3_grid", "small_n20_charged_grid_pos_emb"])
def test_variant_fixtures_are_oracle_outputs(name):
    g = load_golden(name)
    kw = {k[len("variant_"):]: (tuple(str(x) for x in g[k]) if k.endswith("dataset_list") else g[k].item()) for k in g if k.startswith("variant_")}
    o = Oracle(W.make_synthetic_weights(int(g["weights_seed"]), **kw))
    e, f = o.energy_forces(g["z"], g["pos"][0].astype(np.float64), charge=int(g["charge"]), spin=int(g["spin"]), task=str(g["task"]))
    assert abs(e - g["energy"][0]) <= 1e-9 and np.abs(f - g["forces"][0]).max() <= 1e-10
