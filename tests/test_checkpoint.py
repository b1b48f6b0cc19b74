"""Row f4 (first half): state-dict conversion with MoLE merging -- CPU checks on a synthetic fairchem-style state dict."""
import importlib

import numpy as np
import pytest
import torch

from pdb2reaction_amd import weights as W

CK = importlib.import_module("pdb2reaction_amd.checkpoint")

# a model config as a fairchem eSCN-MD backbone would carry it ([3P-UNVERIFIED] names)
UMA_S_CONFIG = {"model": "escnmd_backbone", "sphere_channels": 128, "hidden_channels": 128, "edge_channels": 128, "lmax": 2, "mmax": 2,
                "num_layers": 4, "num_distance_basis": 64, "distance_function": "gaussian", "norm_type": "rms_norm_sh", "act_type": "gate",
                "ff_type": "spectral", "chg_spin_emb_type": "rand_emb", "cutoff": 6.0, "max_neighbors": 300, "max_num_elements": 100,
                "otf_graph": True, "direct_forces": False, "regress_stress": False, "always_use_pbc": False,
                "dataset_list": ["oc20", "omol", "omat", "odac", "omc"], "num_experts": 4, "use_dataset_embedding": True}


def _fake_state(w, n_exp=4, seed=3):
    """Split every SO(2) linear into n_exp random experts whose alpha-weighted sum is the merged weight."""
    rng = np.random.default_rng(seed)
    alpha = rng.random(n_exp); alpha /= alpha.sum()
    state = {}
    for name, arr in w.items():
        if name in ("normalizer.rmsd", "element_refs"):
            continue
        if ".so2_conv_" in name and name.endswith(".weight") and ".rad_func." not in name:
            ex = rng.normal(size=(n_exp,) + arr.shape)
            ex[-1] = (arr.astype(np.float64) - np.tensordot(alpha[:-1], ex[:-1], axes=(0, 0))) / alpha[-1]
            state["backbone." + name[:-len(".weight")] + ".weights"] = torch.tensor(ex)          # torch tensors, float64
        else:
            state["backbone." + name] = torch.tensor(arr)
    extra = {"normalizer.rmsd": w["normalizer.rmsd"], "element_refs": w["element_refs"]}
    return state, alpha, extra


def test_merge_and_roundtrip():
    w = W.make_synthetic_weights(0)
    state, alpha, extra = _fake_state(w)
    got = CK.from_state_dict(state, coefficients=alpha, extra=extra)
    assert set(got) == set(W.param_shapes())
    for k in w:
        assert got[k].dtype == np.float32
        np.testing.assert_allclose(got[k], w[k], rtol=0, atol=2e-6 * max(1.0, np.abs(w[k]).max()))
    with pytest.raises(ValueError, match="merged_for"):
        CK.convert(state, coefficients=alpha, extra=extra, model_config=CK.ASSUME_UMA_S)   # a merge without its system record is refused
    rec = W.system_record([1, 1, 6, 8, 1], 0, 1, "omol")
    with pytest.raises(ValueError, match="model_config"):
        CK.convert(state, coefficients=alpha, extra=extra, merged_for=rec)  # no config, no stated assumption: refused
    blob = CK.convert(state, coefficients=alpha, extra=extra, merged_for=rec, model_config=CK.ASSUME_UMA_S)
    back = W.unpack_blob(blob)
    assert all(np.array_equal(back[k], got[k]) for k in got)
    # the blob remembers the system the experts were merged for and refuses any other
    assert back.meta["merged_for"] == {"composition": {"1": 3, "6": 1, "8": 1}, "charge": 0, "spin": 1, "task": "omol"}
    W.check_merged_for(back, [8, 1, 1, 1, 6], 0, 1, "omol")                 # same multiset, other order: fine
    for z, q, s, t, what in [([1, 1, 6, 8], 0, 1, "omol", "composition"), ([1, 1, 6, 8, 1], 1, 1, "omol", "charge"),
                             ([1, 1, 6, 8, 1], 0, 3, "omol", "spin"), ([1, 1, 6, 8, 1], 0, 1, "omat", "task")]:
        with pytest.raises(ValueError, match=what):
            W.check_merged_for(back, z, q, s, t)
    W.check_merged_for(w, [1, 2, 3], 5, 2, "oc20")                          # synthetic weights carry no record: any system
    m = CK.merge_mole(np.arange(12.0).reshape(3, 2, 2), [0.5, 0.25, 0.25])
    np.testing.assert_allclose(m, [[2.5 + 0.5, 3.5 + 0.5], [4.5 + 0.5, 5.5 + 0.5]])


def test_errors_and_renaming():
    w = W.make_synthetic_weights(0)
    state, alpha, extra = _fake_state(w)
    with pytest.raises(ValueError, match="needs MoLE coefficients"):
        CK.from_state_dict(state, extra=extra)
    with pytest.raises(ValueError, match="experts but"):
        CK.from_state_dict(state, coefficients=alpha[:-1], extra=extra)
    with pytest.raises(KeyError, match="missing"):
        CK.from_state_dict(state, coefficients=alpha)                       # normalizer / element refs not supplied
    bad = dict(state); bad["backbone.unknown.weight"] = torch.zeros(3)
    with pytest.raises(KeyError, match="not a parameter"):
        CK.from_state_dict(bad, coefficients=alpha, extra=extra)
    assert "unknown.weight" not in CK.from_state_dict(bad, coefficients=alpha, extra=extra, strict=False)
    shp = dict(state); shp["backbone.mix_csd.bias"] = torch.zeros(7)
    with pytest.raises(ValueError, match="shape"):
        CK.from_state_dict(shp, coefficients=alpha, extra=extra)
    # renaming: dict and callable forms, dropping keys with None
    ren = {k.replace("backbone.norm.", "backbone.final_norm."): v for k, v in state.items()}
    got = CK.from_state_dict(ren, coefficients=alpha, extra=extra, rename=lambda n: n.replace("final_norm.", "norm."))
    assert np.array_equal(got["norm.affine_bias"], w["norm.affine_bias"])
    ren["backbone.routing_mlp.0.weight"] = torch.zeros(2, 2)
    got = CK.from_state_dict(ren, coefficients=alpha, extra=extra,
                             rename=lambda n: None if n.startswith("routing_mlp.") else n.replace("final_norm.", "norm."))
    assert set(got) == set(W.param_shapes())


def test_mole_routing_and_convert_for_system():
    """Row f4: routing network restated from SURVEY.md App. A.7 ([3P-UNVERIFIED]) -> alpha -> merged blob bound to one system."""
    w = W.make_synthetic_weights(0)
    state, _, extra = _fake_state(w)
    rng = np.random.default_rng(11)
    n_exp = next(v.shape[0] for k, v in state.items() if k.endswith(".weights"))
    c = W.SPHERE_CHANNELS
    state["backbone.composition_embedding.weight"] = torch.tensor(rng.standard_normal((W.MAX_NUM_ELEMENTS, c)))
    state["backbone.routing_mlp.0.weight"] = torch.tensor(rng.standard_normal((64, 2 * c)) / np.sqrt(2 * c))
    state["backbone.routing_mlp.0.bias"] = torch.tensor(0.1 * rng.standard_normal(64))
    state["backbone.routing_mlp.2.weight"] = torch.tensor(rng.standard_normal((n_exp, 64)) / 8.0)
    state["backbone.routing_mlp.2.bias"] = torch.tensor(0.1 * rng.standard_normal(n_exp))
    z = [8, 1, 1, 6, 1, 1, 1, 7]
    a = CK.mole_coefficients(state, z, 0, 1, "omol")
    assert a.shape == (n_exp,) and np.all(a > 0) and abs(a.sum() - 1.0) < 1e-14
    np.testing.assert_allclose(CK.mole_coefficients(state, z[::-1], 0, 1, "omol"), a, rtol=1e-13)      # composition, not order
    assert np.abs(CK.mole_coefficients(state, z, -1, 2, "omol") - a).max() > 1e-6                       # charge / spin matter
    assert np.abs(CK.mole_coefficients(state, z, 0, 1, "omat") - a).max() > 1e-6                        # so does the task
    assert np.abs(CK.mole_coefficients(state, z + [26], 0, 1, "omol") - a).max() > 1e-6                 # and the composition
    # independent restatement of the formula
    sd = {k[len("backbone."):]: v.double().numpy() for k, v in state.items() if k.startswith("backbone.")}
    silu = lambda t: t / (1 + np.exp(-t))
    v = np.concatenate([sd["charge_embedding.weight"][100], sd["spin_embedding.weight"][1], sd["dataset_embedding.weight"][1]])
    x = np.concatenate([sd["composition_embedding.weight"][z].mean(0), silu(sd["mix_csd.weight"] @ v + sd["mix_csd.bias"])])
    h = sd["routing_mlp.2.weight"] @ silu(sd["routing_mlp.0.weight"] @ x + sd["routing_mlp.0.bias"]) + sd["routing_mlp.2.bias"]
    np.testing.assert_allclose(a, np.exp(h - h.max()) / np.exp(h - h.max()).sum(), rtol=1e-12)
    # full conversion: merged with alpha, routing tensors dropped, bound to the system
    blob = CK.convert_for_system(state, z, 0, 1, "omol", extra=extra, model_config=UMA_S_CONFIG)
    back = W.unpack_blob(blob)
    assert set(back) == set(W.param_shapes()) and back.meta["merged_for"] == W.system_record(z, 0, 1, "omol")
    key = "blocks.0.edge_wise.so2_conv_1.fc_m0.weight"
    want = CK.merge_mole(state["backbone." + key[:-len(".weight")] + ".weights"].double().numpy(), a).astype(np.float32)
    assert np.array_equal(back[key], want)
    with pytest.raises(ValueError, match="composition"):
        W.check_merged_for(back, z + [1], 0, 1, "omol")
    with pytest.raises(KeyError, match="routing"):
        CK.mole_coefficients({k: v for k, v in state.items() if "routing_mlp" not in k}, z, 0, 1, "omol")


def test_model_config_is_read_and_held_against_the_engine():
    """the loader reads the checkpoint's OWN model config, takes cutoff / max_neighbors from it and refuses
    everything libumx does not implement -- one synthetic checkpoint per mismatch."""
    model = CK.validate_model_config(UMA_S_CONFIG)
    assert model["cutoff"] == 6.0 and model["max_neighbors"] == 300 and model["num_experts"] == 4
    assert "ff_type" in model["checked"] and "lmax" in model["checked"] and model["unknown_keys"] == []
    assert "use_pbc" in model["unchecked_engine_keys"]                      # not mentioned by this config: reported, not assumed
    for key, val, word in [("ff_type", "s2_mlp", "ff_type='s2_mlp'"), ("lmax", 3, "lmax=3"), ("mmax", 1, "mmax=1"), ("sphere_channels", 256, "sphere_channels=256"),
                           ("hidden_channels", 64, "hidden_channels"), ("edge_channels", 64, "edge_channels"), ("num_layers", 6, "num_layers=6"),
                           ("num_distance_basis", 128, "num_distance_basis=128"), ("distance_function", "bessel", "distance_function"),
                           ("norm_type", "layer_norm_sh", "norm_type"), ("act_type", "s2", "act_type"), ("chg_spin_emb_type", "fourier", "chg_spin_emb_type"),
                           ("direct_forces", True, "direct_forces"), ("regress_stress", True, "regress_stress"), ("always_use_pbc", True, "always_use_pbc"),
                           ("max_num_elements", 118, "max_num_elements"), ("cutoff", -1.0, "cutoff"), ("max_neighbors", 0, "max_neighbors"),
                           ("dataset_list", ["omol", "omol"], "dataset_list"), ("grid_resolution", 14, "grid_resolution")]:
        with pytest.raises(CK.UnsupportedCheckpoint, match=word.split("=")[0]) as ei:
            CK.validate_model_config({**UMA_S_CONFIG, key: val})
        assert word.split("=")[0] in str(ei.value)
    with pytest.raises(CK.UnsupportedCheckpoint) as ei:                      # every mismatch is named, not just the first
        CK.validate_model_config({**UMA_S_CONFIG, "act_type": "s2", "lmax": 4})
    assert "act_type" in str(ei.value) and "lmax" in str(ei.value)
    # round 5: the variants SURVEY.md lists as possible are SELECTED, not refused -- and recorded for the loader
    m_g = CK.validate_model_config({**UMA_S_CONFIG, "ff_type": "grid", "chg_spin_emb_type": "pos_emb", "dataset_list": ["omol", "oc20", "omat", "odac", "omc"],
                                    "grid_resolution": None})
    assert m_g["ff_type"] == "grid" and m_g["chg_spin_emb_type"] == "pos_emb" and m_g["dataset_list"] == ["omol", "oc20", "omat", "odac", "omc"]
    assert CK.validate_model_config({**UMA_S_CONFIG, "use_dataset_embedding": False})["dataset_list"] == []
    # free parameters: taken over, aliases understood
    m2 = CK.validate_model_config({**{k: v for k, v in UMA_S_CONFIG.items() if k not in ("cutoff", "max_neighbors")}, "radius": 5.0, "max_neigh": 40})
    assert m2["cutoff"] == 5.0 and m2["max_neighbors"] == 40
    # unknown keys: reported, or refused on request
    with pytest.warns(RuntimeWarning, match="does not know"):
        m3 = CK.validate_model_config({**UMA_S_CONFIG, "so2_attention": True})
    assert m3["unknown_keys"] == ["so2_attention"]
    with pytest.raises(CK.UnsupportedCheckpoint, match="so2_attention"):
        CK.validate_model_config({**UMA_S_CONFIG, "so2_attention": True}, strict_unknown=True)


def test_convert_checkpoint_finds_the_config_and_the_blob_carries_the_graph_defaults():
    w = W.make_synthetic_weights(0)
    state, _, extra = _fake_state(w)
    rng = np.random.default_rng(11)
    n_exp = next(v.shape[0] for k, v in state.items() if k.endswith(".weights"))
    c = W.SPHERE_CHANNELS
    state["backbone.composition_embedding.weight"] = torch.tensor(rng.standard_normal((W.MAX_NUM_ELEMENTS, c)))
    state["backbone.routing_mlp.0.weight"] = torch.tensor(rng.standard_normal((n_exp, 2 * c)) / np.sqrt(2 * c))
    state["backbone.routing_mlp.0.bias"] = torch.tensor(0.1 * rng.standard_normal(n_exp))
    z = [8, 1, 1]
    cfg = {**UMA_S_CONFIG, "cutoff": 5.5, "max_neighbors": 120}
    ckpt = {"epoch": 3, "config": {"optim": {"lr": 1e-3}, "model": {"name": "hydra", "backbone": cfg, "heads": {"energy": {"module": "mlp_efs"}}}},
            "ema_state_dict": state}
    assert CK.find_model_config(ckpt) == cfg and CK.find_model_config({"state_dict": state}) is None
    blob = CK.convert_checkpoint(ckpt, z, 0, 1, "omol", extra=extra)
    back = W.unpack_blob(blob)
    assert back.meta["model"]["cutoff"] == 5.5 and back.meta["model"]["max_neighbors"] == 120 and back.meta["merged_for"] == W.system_record(z, 0, 1, "omol")
    with pytest.raises(CK.UnsupportedCheckpoint, match="no model config"):
        CK.convert_checkpoint({"state_dict": state}, z, 0, 1, "omol", extra=extra)
    with pytest.raises(CK.UnsupportedCheckpoint, match="ff_type"):
        CK.convert_checkpoint({**ckpt, "config": {"model": {"backbone": {**cfg, "ff_type": "grid"}}}}, z, 0, 1, "omol", extra=extra)
    with pytest.raises(KeyError, match="normalizer.rmsd"):
        CK.convert_checkpoint(ckpt, z, 0, 1, "omol")                        # normaliser / element references are mandatory
    with pytest.raises(KeyError, match="no state dict"):
        CK.convert_checkpoint({"config": ckpt["config"]}, z, 0, 1, "omol", extra=extra)
    # the calculator takes the graph defaults from the blob (reference: backbone.cutoff / backbone.max_neighbors, uma_pysis.py:301-309)
    import importlib as _il
    U = _il.import_module("pdb2reaction_amd.uma_pysis")
    seen = {}

    class FakeEngine:
        def __init__(self, *a, **k): pass
        def load_weights(self, wts): seen["meta"] = getattr(wts, "meta", {})
        def set_system(self, zz, charge=0, spin=1, task="omol", radius=None, max_neigh=None): seen.update(radius=radius, max_neigh=max_neigh)

    import pdb2reaction_amd.engine as E
    import tempfile, os as _os
    with tempfile.TemporaryDirectory() as td:
        path = _os.path.join(td, "m.umxw")
        with open(path, "wb") as f:
            f.write(blob)
        real = E.Engine
        E.Engine = FakeEngine
        try:
            core = U.UMAcore(["O", "H", "H"], model=path)
            assert seen["radius"] == 5.5 and seen["max_neigh"] == 120 and core.model_record["cutoff"] == 5.5
            U.UMAcore(["O", "H", "H"], model=path, radius=4.0, max_neigh=10)          # the caller's values win, as in the reference
            assert seen["radius"] == 4.0 and seen["max_neigh"] == 10
            with pytest.warns(RuntimeWarning, match="r_edges"):
                U.UMAcore(["O", "H", "H"], model=path, r_edges=True)
            with pytest.warns(RuntimeWarning, match="workers_per_node"):
                U.UMAcore(["O", "H", "H"], model=path, workers_per_node=4)
        finally:
            E.Engine = real


def _fairchem_style(w, dataset_list):
    """The variant weight set under the module names fairchem gives them [3P-UNVERIFIED]: lookup tables under ``.rand_emb``, one (1, C)
    dataset table per NAME, (lat, long, 9) grid buffers."""
    state = {}
    for name, arr in w.items():
        if name in ("normalizer.rmsd", "element_refs"):
            continue
        if name in ("charge_embedding.weight", "spin_embedding.weight"):
            state["backbone." + name.replace(".weight", ".rand_emb.weight")] = torch.tensor(arr)
        elif name == "dataset_embedding.weight":
            for i, d in enumerate(dataset_list):
                state[f"backbone.dataset_embedding.dataset_emb_dict.{d}.weight"] = torch.tensor(arr[i:i + 1])
        elif name.startswith("so3_grid."):
            state["backbone.SO3_grid.lmax_lmax." + name[len("so3_grid."):]] = torch.tensor(arr.reshape(6, 7, 9))
        else:
            state["backbone." + name] = torch.tensor(arr)
    state["backbone.SO3_grid.lmax_mmax.to_grid_mat"] = torch.zeros(6, 7, 9)           # the other grid of the module tree: dropped
    return state


@pytest.mark.parametrize("variant", [dict(ff_type="grid"), dict(chg_spin_emb_type="pos_emb"), dict(chg_spin_emb_type="lin_emb"),
                                     dict(ff_type="grid", chg_spin_emb_type="pos_emb", dataset_list=("omol", "omat", "oc20")), dict(dataset_list=())])
def test_model_variants_are_converted_not_refused(variant):
    """grid feed-forward (its S2-grid matrices taken from the checkpoint), pos_emb / lin_emb charge-spin embedding,
    a dataset_list in another order or absent -- state dict under fairchem-style names -> blob == the variant's own weight set, the
    blob trailer names the variant and the dataset order, and the blob packs / unpacks."""
    dl = tuple(variant.get("dataset_list", W.DATASET_LIST))
    w = W.make_synthetic_weights(0, **variant)
    state = _fairchem_style(w, dl)
    cfg = {**UMA_S_CONFIG, "ff_type": variant.get("ff_type", "spectral"), "chg_spin_emb_type": variant.get("chg_spin_emb_type", "rand_emb"),
           "dataset_list": list(dl) or ["omol"], "use_dataset_embedding": bool(dl), "num_experts": 1}
    extra = {"normalizer.rmsd": w["normalizer.rmsd"], "atom_refs": {"omol_elem_refs": w["element_refs"], "omat_elem_refs": np.zeros(100)},
             "form_elem_refs": {"omol": np.ones(100)}}
    blob = CK.convert(state, extra=extra, model_config=cfg, task="omol")
    back = W.unpack_blob(blob)
    assert list(back) == list(w) and all(np.array_equal(back[k], w[k]) for k in w)          # atom_refs[omol] -> element_refs; form refs not applied
    v = W.variant_of(back)
    assert back.meta["model"]["ff_type"] == v["ff_type"] == variant.get("ff_type", "spectral")
    assert back.meta["model"]["chg_spin_emb_type"] == v["chg_spin_emb_type"] == variant.get("chg_spin_emb_type", "rand_emb")
    assert back.meta["model"]["dataset_list"] == list(dl) and v["n_datasets"] == len(dl)
    # the config and the tensors must tell the same story
    with pytest.raises(CK.UnsupportedCheckpoint, match="ff_type"):
        CK.convert(state, extra=extra, model_config={**cfg, "ff_type": "grid" if v["ff_type"] == "spectral" else "spectral"}, task="omol")
    if v["ff_type"] == "grid":
        no_grid = {k: t for k, t in state.items() if "SO3_grid" not in k}
        with pytest.raises(KeyError, match="dumped from the loaded model"):             # never re-derived: must come from the checkpoint
            CK.convert(no_grid, extra=extra, model_config=cfg, task="omol")
        got = CK.from_state_dict(no_grid, extra={**extra, "so3_grid.to_grid_mat": w["so3_grid.to_grid_mat"].reshape(6, 7, 9),
                                                 "so3_grid.from_grid_mat": w["so3_grid.from_grid_mat"]}, task="omol", dataset_list=dl)
        assert np.array_equal(got["so3_grid.to_grid_mat"], w["so3_grid.to_grid_mat"])


def test_element_references_as_the_reference_obtains_them():
    """uma_pysis.py:231-239 hands fairchem ``atom_refs`` and ``form_elem_refs``; both are accepted here."""
    a = {"omol_elem_refs": np.arange(100.0), "omat": np.arange(100.0) * 2}
    f = {"omol": np.full(100, 0.5)}
    assert np.array_equal(CK.element_refs_from(a, f, task="omol"), np.arange(100.0))
    assert np.array_equal(CK.element_refs_from(a, f, task="omol", formation_energy=True), np.arange(100.0) - 0.5)
    assert np.array_equal(CK.element_refs_from(a, None, task="omat"), np.arange(100.0) * 2)
    assert np.array_equal(CK.element_refs_from(np.arange(120.0) * (np.arange(120) < 100), None, task="omol"), np.arange(100.0))
    with pytest.raises(KeyError, match="odac"):
        CK.element_refs_from(a, None, task="odac")
    with pytest.raises(ValueError, match="not representable"):
        CK.element_refs_from(np.arange(120.0), None, task="omol")


def test_mole_routing_sees_the_variant_s_system_embedding():
    """alpha depends on sys_emb: the pos_emb / per-name-dataset forms must feed the routing network the same vector the engine adds."""
    from oracle.escn_md_oracle import Oracle

    dl = ("omol", "omat", "oc20")
    w = W.make_synthetic_weights(0, chg_spin_emb_type="pos_emb", dataset_list=dl)
    state = _fairchem_style(w, dl)
    sd = {k[len("backbone."):]: v.double().numpy() for k, v in state.items()}
    se = CK.system_embedding(lambda n: sd[n], lambda n: n in sd, -1, 2, "omat", dl)
    want = Oracle(w).system_embedding(-1, 2, "omat").numpy()
    np.testing.assert_allclose(se, want, rtol=1e-13, atol=1e-15)
    assert np.abs(se - CK.system_embedding(lambda n: sd[n], lambda n: n in sd, -1, 0, "omat", dl)).max() > 1e-3       # null spin differs


def test_per_name_dataset_tables_keep_their_names():
    """per-name tables without a dataset_list were stacked alphabetically while the trailer said UMA's order, so
    'omol' got odac's row.  Row <-> name identity must hold for every way the order can be known, and an unknowable order is refused."""
    w = W.make_synthetic_weights(0)
    state = _fairchem_style(w, W.DATASET_LIST)
    extra = {"normalizer.rmsd": w["normalizer.rmsd"], "element_refs": w["element_refs"]}
    per_name = {d: state[f"backbone.dataset_embedding.dataset_emb_dict.{d}.weight"].numpy().reshape(-1) for d in W.DATASET_LIST}

    def rows_match(blob):
        back = W.unpack_blob(blob)
        names = back.meta["model"]["dataset_list"]
        assert sorted(names) == sorted(W.DATASET_LIST)
        for i, d in enumerate(names):
            assert np.array_equal(back["dataset_embedding.weight"][i], per_name[d].astype(np.float32)), (i, d)
        return names

    assert rows_match(CK.convert(state, extra=extra, model_config=CK.ASSUME_UMA_S)) == list(W.DATASET_LIST)     # no list: UMA's own order
    other = ["omc", "omol", "oc20", "odac", "omat"]
    assert rows_match(CK.convert(state, extra=extra, model_config=CK.ASSUME_UMA_S, dataset_list=other)) == other
    assert rows_match(CK.convert(state, extra=extra, model_config={**UMA_S_CONFIG, "dataset_list": other, "num_experts": 1})) == other
    # names the list does not cover are not dropped silently; an unknown name set without a list has no knowable order
    with pytest.raises(KeyError, match="refusing to drop"):
        CK.convert(state, extra=extra, model_config=CK.ASSUME_UMA_S, dataset_list=other[:4])
    renamed = {k.replace(".omc.", ".mine."): v for k, v in state.items()}
    with pytest.raises(KeyError, match="without a dataset_list"):
        CK.convert(renamed, extra=extra, model_config=CK.ASSUME_UMA_S)
    # use_dataset_embedding = False with dataset tables in the state dict: config and tensors disagree
    with pytest.raises(CK.UnsupportedCheckpoint, match="use_dataset_embedding"):
        CK.convert(state, extra=extra, model_config={**UMA_S_CONFIG, "use_dataset_embedding": False, "num_experts": 1})
    # MoLE routing refuses a task the engine's list does not have even when a per-name table for it exists
    sd = {k[len("backbone."):]: v.double().numpy() for k, v in state.items()}
    with pytest.raises(ValueError, match="not in"):
        CK.system_embedding(lambda n: sd[n], lambda n: n in sd, 0, 1, "omc", ("omol", "omat"))


def test_blob_trailer_is_located_by_the_header_not_by_search():
    """the magic of the JSON trailer may occur inside tensor data; the trailer sits behind the header's tensor extent."""
    w = W.make_synthetic_weights(0)
    blob = W.pack_blob(w, meta={"model": {"dataset_list": ["omol", "omat", "oc20", "odac", "omc"]}})
    assert W.blob_meta(blob)["model"]["dataset_list"][1] == "omat"
    plain = W.pack_blob(dict(w))                 # (a plain dict: no meta, no trailer)
    assert W.blob_meta(plain) == {} and W.blob_meta(plain + b"junk") == {}
    poisoned = bytearray(plain)
    at = len(poisoned) - 4096                    # magic bytes in the middle of the last tensor
    poisoned[at:at + 8] = W.META_MAGIC
    assert W.blob_meta(bytes(poisoned)) == {}
    assert W.blob_meta(bytes(poisoned) + blob[len(plain):])["model"]["dataset_list"][0] == "omol"
