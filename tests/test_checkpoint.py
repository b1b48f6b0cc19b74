"""Row f4 (first half): state-dict conversion with MoLE merging -- CPU checks on a synthetic fairchem-style state dict."""
import importlib

import numpy as np
import pytest
import torch

from pdb2reaction_amd import weights as W

CK = importlib.import_module("pdb2reaction_amd.checkpoint")


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
        CK.convert(state, coefficients=alpha, extra=extra)                  # a merge without its system record is refused
    rec = W.system_record([1, 1, 6, 8, 1], 0, 1, "omol")
    blob = CK.convert(state, coefficients=alpha, extra=extra, merged_for=rec)
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
    blob = CK.convert_for_system(state, z, 0, 1, "omol", extra=extra)
    back = W.unpack_blob(blob)
    assert set(back) == set(W.param_shapes()) and back.meta["merged_for"] == W.system_record(z, 0, 1, "omol")
    key = "blocks.0.edge_wise.so2_conv_1.fc_m0.weight"
    want = CK.merge_mole(state["backbone." + key[:-len(".weight")] + ".weights"].double().numpy(), a).astype(np.float32)
    assert np.array_equal(back[key], want)
    with pytest.raises(ValueError, match="composition"):
        W.check_merged_for(back, z + [1], 0, 1, "omol")
    with pytest.raises(KeyError, match="routing"):
        CK.mole_coefficients({k: v for k, v in state.items() if "routing_mlp" not in k}, z, 0, 1, "omol")
