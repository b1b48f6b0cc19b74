"""UMA-S (eSCN-MD) parameter inventory, synthetic initialisation and the on-disk weight blob.

The reference loads a gated fairchem checkpoint through
``pretrained_mlip.get_predict_unit(model, device)`` (reference ``pdb2reaction/uma_pysis.py:246-250``).
No checkpoint exists in this environment, so the engine consumes a flat
"merged" parameter set (MoLE experts already merged) in the shapes listed
by :func:`param_shapes`.  :func:`make_synthetic_weights` fills those shapes deterministically from
a seed; the same arrays feed the HIP engine (through :func:`pack_blob`) and the CPU oracle.

Blob layout (little endian), parsed by ``csrc/umx_api.hip``::

    char  magic[8] = "UMXW0001"
    u32   n_tensors, u32 reserved
    n_tensors x { char name[96]; u32 ndim; u32 dims[4]; u64 offset; u64 nbytes }
    f32 data section (each tensor 64-byte aligned, offsets relative to the data section)
    optional trailer (ignored by the C parser): char "UMXMETA1", u32 nbytes, JSON -- e.g.
    {"merged_for": {"composition": {"1": 12, "6": 4}, "charge": 0, "spin": 1, "task": "omol"}} for a
    parameter set whose MoLE experts were merged for ONE system (checkpoint.py); consumers must refuse
    to bind such a blob to any other system (:func:`check_merged_for`).
"""
from __future__ import annotations

import json
import struct
from collections import OrderedDict
from typing import Any, Dict, Optional, Sequence, Tuple

import numpy as np

# ---- UMA-S hyper-parameters (fairchem eSCN-MD "K4L2") -------------------
LMAX = 2
MMAX = 2
NUM_SPH = (LMAX + 1) ** 2          # 9
SPHERE_CHANNELS = 128              # C
HIDDEN_CHANNELS = 128              # H
EDGE_CHANNELS = 128
NUM_LAYERS = 4
NUM_DISTANCE_BASIS = 64
CUTOFF = 6.0                       # Angstrom
MAX_NEIGHBORS = 300
MAX_NUM_ELEMENTS = 100
EDGE_FEAT = NUM_DISTANCE_BASIS + 2 * EDGE_CHANNELS   # 320
RADIAL_HIDDEN = 128
DEG_RESCALE = 5.0
CHARGE_OFFSET = 100                # charge index = charge + 100 (rand_emb table)
NUM_CHARGE = 201
NUM_SPIN = 101
DATASET_LIST = ("oc20", "omol", "omat", "odac", "omc")
NORM_EPS = 1e-5
LN_EPS = 1e-5

# m-primary row r holds the l-primary coefficient TO_M[r] (index l*l+l+m); SURVEY.md Appendix A.3
TO_M = (0, 2, 6, 3, 7, 1, 5, 8, 4)
# degree l of every l-primary coefficient and of every m-primary row
L_OF_LP = (0, 1, 1, 1, 2, 2, 2, 2, 2)
L_OF_MP = tuple(L_OF_LP[i] for i in TO_M)

MAGIC = b"UMXW0001"
META_MAGIC = b"UMXMETA1"


class WeightSet(OrderedDict):
    """name -> float32 array, plus ``meta`` (the blob trailer: provenance, the system a MoLE merge was made for)."""

    meta: Dict[str, Any]

    def __init__(self, *a, meta: Optional[Dict[str, Any]] = None, **kw):
        super().__init__(*a, **kw)
        self.meta = dict(meta or {})


def system_record(atomic_numbers: Sequence[int], charge: int, spin: int, task: str) -> Dict[str, Any]:
    """What MoLE routing depends on: the element multiset, total charge, spin multiplicity, task."""
    zs, counts = np.unique(np.asarray(atomic_numbers, dtype=np.int64), return_counts=True)
    return {"composition": {str(int(z)): int(c) for z, c in zip(zs, counts)}, "charge": int(charge), "spin": int(spin), "task": str(task)}


def check_merged_for(weights: Dict[str, np.ndarray], atomic_numbers: Sequence[int], charge: int, spin: int, task: str) -> None:
    """Raise ValueError when `weights` carries a ``merged_for`` record that does not match the system about to be bound.
    Parameter sets without the record (synthetic weights, single-expert checkpoints) fit every system."""
    want = (getattr(weights, "meta", None) or {}).get("merged_for")
    if not want:
        return
    have = system_record(atomic_numbers, charge, spin, task)
    if have != want:
        diff = [k for k in ("composition", "charge", "spin", "task") if have.get(k) != want.get(k)]
        raise ValueError(
            f"these weights were MoLE-merged for another system (differs in {', '.join(diff)}: blob {{{', '.join(f'{k}={want.get(k)}' for k in diff)}}}, "
            f"requested {{{', '.join(f'{k}={have.get(k)}' for k in diff)}}}); re-run checkpoint.convert for this system")


def _radial_shapes(prefix: str, out_dim: int) -> "OrderedDict[str, Tuple[int, ...]]":
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    s[f"{prefix}.fc1.weight"] = (RADIAL_HIDDEN, EDGE_FEAT)
    s[f"{prefix}.fc1.bias"] = (RADIAL_HIDDEN,)
    s[f"{prefix}.ln1.weight"] = (RADIAL_HIDDEN,)
    s[f"{prefix}.ln1.bias"] = (RADIAL_HIDDEN,)
    s[f"{prefix}.fc2.weight"] = (RADIAL_HIDDEN, RADIAL_HIDDEN)
    s[f"{prefix}.fc2.bias"] = (RADIAL_HIDDEN,)
    s[f"{prefix}.ln2.weight"] = (RADIAL_HIDDEN,)
    s[f"{prefix}.ln2.bias"] = (RADIAL_HIDDEN,)
    s[f"{prefix}.fc3.weight"] = (out_dim, RADIAL_HIDDEN)
    s[f"{prefix}.fc3.bias"] = (out_dim,)
    return s


def param_shapes() -> "OrderedDict[str, Tuple[int, ...]]":
    """Name -> shape of every parameter the engine consumes (nn.Linear layout: [out, in])."""
    C, H, L1 = SPHERE_CHANNELS, HIDDEN_CHANNELS, LMAX + 1
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    s["sphere_embedding.weight"] = (MAX_NUM_ELEMENTS, C)
    s["charge_embedding.weight"] = (NUM_CHARGE, C)
    s["spin_embedding.weight"] = (NUM_SPIN, C)
    s["dataset_embedding.weight"] = (len(DATASET_LIST), C)
    s["mix_csd.weight"] = (C, 3 * C)
    s["mix_csd.bias"] = (C,)
    s["source_embedding.weight"] = (MAX_NUM_ELEMENTS, EDGE_CHANNELS)
    s["target_embedding.weight"] = (MAX_NUM_ELEMENTS, EDGE_CHANNELS)
    s.update(_radial_shapes("edge_degree_embedding.rad_func", L1 * C))
    for i in range(NUM_LAYERS):
        b = f"blocks.{i}"
        s[f"{b}.norm_1.affine_weight"] = (L1, C)
        s[f"{b}.norm_1.affine_bias"] = (C,)
        # SO(2) conv 1: input 2C channels per coefficient, extra lmax*H gate scalars on m=0
        s[f"{b}.edge_wise.so2_conv_1.fc_m0.weight"] = (LMAX * H + L1 * H, L1 * 2 * C)      # (640, 768)
        s[f"{b}.edge_wise.so2_conv_1.fc_m0.bias"] = (LMAX * H + L1 * H,)
        s[f"{b}.edge_wise.so2_conv_1.so2_m_conv.0.fc.weight"] = (2 * 2 * H, 2 * 2 * C)    # (512, 512)
        s[f"{b}.edge_wise.so2_conv_1.so2_m_conv.1.fc.weight"] = (2 * 1 * H, 1 * 2 * C)    # (256, 256)
        s.update(_radial_shapes(f"{b}.edge_wise.so2_conv_1.rad_func",
                                L1 * 2 * C + 2 * 2 * C + 2 * C))                           # 1536
        s[f"{b}.edge_wise.so2_conv_2.fc_m0.weight"] = (L1 * C, L1 * H)                     # (384, 384)
        s[f"{b}.edge_wise.so2_conv_2.fc_m0.bias"] = (L1 * C,)
        s[f"{b}.edge_wise.so2_conv_2.so2_m_conv.0.fc.weight"] = (2 * 2 * C, 2 * H)         # (512, 256)
        s[f"{b}.edge_wise.so2_conv_2.so2_m_conv.1.fc.weight"] = (2 * 1 * C, 1 * H)         # (256, 128)
        s[f"{b}.norm_2.affine_weight"] = (L1, C)
        s[f"{b}.norm_2.affine_bias"] = (C,)
        s[f"{b}.atom_wise.scalar_mlp.weight"] = (LMAX * H, C)
        s[f"{b}.atom_wise.scalar_mlp.bias"] = (LMAX * H,)
        s[f"{b}.atom_wise.so3_linear_1.weight"] = (L1, H, C)
        s[f"{b}.atom_wise.so3_linear_1.bias"] = (H,)
        s[f"{b}.atom_wise.so3_linear_2.weight"] = (L1, C, H)
        s[f"{b}.atom_wise.so3_linear_2.bias"] = (C,)
    s["norm.affine_weight"] = (L1, C)
    s["norm.affine_bias"] = (C,)
    s["energy_block.0.weight"] = (H, C)
    s["energy_block.0.bias"] = (H,)
    s["energy_block.2.weight"] = (H, H)
    s["energy_block.2.bias"] = (H,)
    s["energy_block.4.weight"] = (1, H)
    s["energy_block.4.bias"] = (1,)
    s["normalizer.rmsd"] = (1,)
    s["element_refs"] = (MAX_NUM_ELEMENTS,)
    return s


def make_synthetic_weights(seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    """Deterministic stand-in for the gated UMA checkpoint (fan-in scaled normal, float32).

    Affine/LayerNorm scales are 1 + 0.1 N(0,1), biases 0.1 N(0,1), so that every parameter
    participates non-trivially in parity tests.  SO(2) m>0 weights carry the 1/sqrt(2) factor.
    """
    rng = np.random.default_rng(seed)
    out = WeightSet(meta={"source": f"synthetic(seed={int(seed)})"})
    for name, shape in param_shapes().items():
        leaf = name.split(".")[-1]
        if name == "normalizer.rmsd":
            a = np.array([1.5])
        elif name == "element_refs":
            z = np.arange(MAX_NUM_ELEMENTS, dtype=np.float64)
            a = -13.6 * z ** 1.2 + rng.standard_normal(MAX_NUM_ELEMENTS)
        elif "embedding" in name and leaf == "weight" and "rad_func" not in name:
            a = rng.standard_normal(shape)
        elif leaf == "affine_weight" or (leaf == "weight" and (".ln1." in name or ".ln2." in name)):
            a = 1.0 + 0.1 * rng.standard_normal(shape)
        elif leaf in ("bias", "affine_bias"):
            a = 0.1 * rng.standard_normal(shape)
        else:
            fan_in = shape[-1]
            a = rng.standard_normal(shape) / np.sqrt(fan_in)
            if ".so2_m_conv." in name:
                a = a / np.sqrt(2.0)
        out[name] = np.ascontiguousarray(a, dtype=np.float32)
    return out


def pack_blob(weights: Dict[str, np.ndarray], meta: Optional[Dict[str, Any]] = None) -> bytes:
    """Serialise a name->array dict into the UMXW0001 blob read by ``umx_load_weights`` (``meta`` or ``weights.meta``
    goes into the JSON trailer)."""
    shapes = param_shapes()
    missing = [k for k in shapes if k not in weights]
    if missing:
        raise KeyError(f"weights missing {len(missing)} tensors, e.g. {missing[:3]}")
    entries = []
    chunks = []
    off = 0
    for name, shape in shapes.items():
        a = np.ascontiguousarray(weights[name], dtype=np.float32)
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"{name}: shape {a.shape} != expected {shape}")
        raw = a.tobytes()
        pad = (-len(raw)) % 64
        dims = list(shape) + [1] * (4 - len(shape))
        entries.append(struct.pack("<96sI4IQQ", name.encode(), len(shape), *dims, off, len(raw)))
        chunks.append(raw + b"\0" * pad)
        off += len(raw) + pad
    head = MAGIC + struct.pack("<II", len(entries), 0)
    table = b"".join(entries)
    pad = (-(len(head) + len(table))) % 64
    meta = meta if meta is not None else getattr(weights, "meta", None)
    trailer = b""
    if meta:
        js = json.dumps(meta, sort_keys=True).encode()
        trailer = META_MAGIC + struct.pack("<I", len(js)) + js
    return head + table + b"\0" * pad + b"".join(chunks) + trailer


def unpack_blob(blob: bytes) -> "WeightSet":
    if blob[:8] != MAGIC:
        raise ValueError("not a UMXW0001 weight blob")
    n, _ = struct.unpack_from("<II", blob, 8)
    esz = struct.calcsize("<96sI4IQQ")
    pos = 16
    ents = []
    for _ in range(n):
        name, ndim, d0, d1, d2, d3, off, nb = struct.unpack_from("<96sI4IQQ", blob, pos)
        ents.append((name.rstrip(b"\0").decode(), (d0, d1, d2, d3)[:ndim], off, nb))
        pos += esz
    data0 = pos + ((-pos) % 64)
    out = WeightSet()
    end = data0
    for name, shape, off, nb in ents:
        out[name] = np.frombuffer(blob, dtype=np.float32, count=nb // 4, offset=data0 + off).reshape(shape).copy()
        end = max(end, data0 + off + nb + ((-nb) % 64))
    if blob[end:end + 8] == META_MAGIC:
        (n_js,) = struct.unpack_from("<I", blob, end + 8)
        out.meta = json.loads(blob[end + 12:end + 12 + n_js].decode())
    return out


def save_weights(path: str, weights: Dict[str, np.ndarray]) -> None:
    with open(path, "wb") as f:
        f.write(pack_blob(weights))


def load_weights(path: str) -> "WeightSet":
    with open(path, "rb") as f:
        return unpack_blob(f.read())
