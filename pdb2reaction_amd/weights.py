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
# model variants SURVEY.md marks "unsure" (section 2.4 K8, Appendix A header): the feed-forward block and the charge / spin embedding
FF_TYPES = ("spectral", "grid")
EMB_TYPES = ("rand_emb", "pos_emb", "lin_emb")
GRID_RESOLUTION = (2 * (LMAX + 1), 2 * (LMAX + 1) + 1)     # fairchem SO3_Grid default for lmax == mmax: (lat, long) = (6, 7) [3P-UNVERIFIED]
GRID_POINTS_MAX = 128                                      # the engine's grid kernels take any G <= 128
MAX_DATASETS = 32

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


def variant_of(weights: Dict[str, Any]) -> Dict[str, Any]:
    """The model variant a weight set IS, read off the tensors it carries (the same rule ``umx_load_weights`` applies):
    ``{"ff_type", "chg_spin_emb_type", "n_datasets", "grid_points", "grid_bias"}``."""
    if "charge_embedding.W" in weights:
        emb = "pos_emb"
    elif "charge_embedding.lin_emb.weight" in weights:
        emb = "lin_emb"
    else:
        emb = "rand_emb"
    grid = "blocks.0.atom_wise.grid_mlp.0.weight" in weights
    de = weights.get("dataset_embedding.weight")
    return {"ff_type": "grid" if grid else "spectral", "chg_spin_emb_type": emb,
            "n_datasets": int(np.asarray(de).shape[0]) if de is not None else 0,
            "grid_points": int(np.asarray(weights["so3_grid.to_grid_mat"]).shape[0]) if grid and "so3_grid.to_grid_mat" in weights else 0,
            "grid_bias": bool(grid and "blocks.0.atom_wise.grid_mlp.0.bias" in weights)}


def param_shapes(ff_type: str = "spectral", chg_spin_emb_type: str = "rand_emb", n_datasets: int = len(DATASET_LIST),
                 grid_points: int = GRID_RESOLUTION[0] * GRID_RESOLUTION[1], grid_bias: bool = False) -> "OrderedDict[str, Tuple[int, ...]]":
    """Name -> shape of every parameter the engine consumes (nn.Linear layout: [out, in]) for one model variant
    (``param_shapes(**variant_of(weights))``).  Defaults = the variant every earlier round built: spectral feed-forward, lookup
    charge / spin tables, the five UMA datasets.

    * ``ff_type="grid"``: ``blocks.<i>.atom_wise.grid_mlp.{0,2,4}.weight`` (+ ``.bias`` with ``grid_bias``) and the S2-grid matrices
      ``so3_grid.to_grid_mat`` / ``so3_grid.from_grid_mat`` (G, 9) -- data of the checkpoint, in the coefficient order l*l+l+m;
    * ``chg_spin_emb_type="pos_emb"``: frequency vectors ``{charge,spin}_embedding.W`` (C/2); ``"lin_emb"``: ``….lin_emb.weight`` (C, 1)
      + ``.bias`` (C);
    * ``n_datasets``: rows of ``dataset_embedding.weight`` (0: ``use_dataset_embedding = False``, ``mix_csd`` then takes 2C inputs)."""
    if ff_type not in FF_TYPES or chg_spin_emb_type not in EMB_TYPES:
        raise ValueError(f"ff_type must be one of {FF_TYPES}, chg_spin_emb_type one of {EMB_TYPES}")
    if not (0 <= int(n_datasets) <= MAX_DATASETS) or (ff_type == "grid" and not (1 <= int(grid_points) <= GRID_POINTS_MAX)):
        raise ValueError(f"n_datasets must be in [0, {MAX_DATASETS}], grid_points in [1, {GRID_POINTS_MAX}]")
    C, H, L1 = SPHERE_CHANNELS, HIDDEN_CHANNELS, LMAX + 1
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    s["sphere_embedding.weight"] = (MAX_NUM_ELEMENTS, C)
    if chg_spin_emb_type == "rand_emb":
        s["charge_embedding.weight"] = (NUM_CHARGE, C)
        s["spin_embedding.weight"] = (NUM_SPIN, C)
    elif chg_spin_emb_type == "pos_emb":
        s["charge_embedding.W"] = (C // 2,)
        s["spin_embedding.W"] = (C // 2,)
    else:
        for which in ("charge", "spin"):
            s[f"{which}_embedding.lin_emb.weight"] = (C, 1)
            s[f"{which}_embedding.lin_emb.bias"] = (C,)
    if n_datasets:
        s["dataset_embedding.weight"] = (int(n_datasets), C)
    s["mix_csd.weight"] = (C, (3 if n_datasets else 2) * C)
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
        if ff_type == "grid":
            for li, shp in ((0, (H, C)), (2, (H, H)), (4, (C, H))):
                s[f"{b}.atom_wise.grid_mlp.{li}.weight"] = shp
                if grid_bias:
                    s[f"{b}.atom_wise.grid_mlp.{li}.bias"] = (shp[0],)
        else:
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
    if ff_type == "grid":
        s["so3_grid.to_grid_mat"] = (int(grid_points), NUM_SPH)
        s["so3_grid.from_grid_mat"] = (int(grid_points), NUM_SPH)
    return s


def synthetic_grid_matrices(resolution: Tuple[int, int] = GRID_RESOLUTION) -> Tuple[np.ndarray, np.ndarray]:
    """Stand-ins for the S2-grid buffers of a checkpoint's ``SO3_Grid`` -- OWN construction for synthetic weight sets, NOT e3nn's
    ``ToS2Grid`` / ``FromS2Grid`` matrices (a real checkpoint brings its own; the engine treats both as opaque (G, 9) data).
    Grid: ``lat`` polar angles beta_a = (a + 1/2) pi / lat about the y axis (the model's polar axis) x ``long`` azimuths
    alpha_b = 2 pi b / long; ``to_grid[g, i]`` = the model's real harmonic i (l = 1: x, y, z; l = 2: the five traceless quadratic
    forms of the oracle header) at direction g times sqrt(2l + 1); ``from_grid`` = ``to_grid (to_grid^T to_grid)^-1``, so that
    from-grid of to-grid is the identity on the 9 coefficients."""
    lat, lon = int(resolution[0]), int(resolution[1])
    beta = (np.arange(lat) + 0.5) * np.pi / lat
    alpha = np.arange(lon) * 2.0 * np.pi / lon
    bb, aa = np.meshgrid(beta, alpha, indexing="ij")
    x, y, z = (np.sin(bb) * np.sin(aa)).ravel(), np.cos(bb).ravel(), (np.sin(bb) * np.cos(aa)).ravel()
    s3 = np.sqrt(3.0)
    sh = np.stack([np.ones_like(x), x, y, z, s3 * x * z, s3 * x * y, y * y - 0.5 * (x * x + z * z), s3 * y * z, 0.5 * s3 * (z * z - x * x)], axis=1)
    to_grid = sh * np.sqrt(2.0 * np.asarray(L_OF_LP, dtype=np.float64) + 1.0)[None, :]
    from_grid = to_grid @ np.linalg.inv(to_grid.T @ to_grid)
    return to_grid.astype(np.float32), from_grid.astype(np.float32)


def make_synthetic_weights(seed: int = 0, ff_type: str = "spectral", chg_spin_emb_type: str = "rand_emb",
                           dataset_list: Optional[Sequence[str]] = None, grid_bias: bool = False) -> "OrderedDict[str, np.ndarray]":
    """Deterministic stand-in for the gated UMA checkpoint (fan-in scaled normal, float32).

    Affine/LayerNorm scales are 1 + 0.1 N(0,1), biases 0.1 N(0,1), so that every parameter
    participates non-trivially in parity tests.  SO(2) m>0 weights carry the 1/sqrt(2) factor.
    The defaults give the weight set of every earlier round bit for bit (the variant tensors draw from their OWN generators);
    ``ff_type="grid"`` / ``chg_spin_emb_type="pos_emb" | "lin_emb"`` / another ``dataset_list`` give the model variants SURVEY.md
    marks as possible for the real checkpoint.
    """
    rng = np.random.default_rng(seed)
    dl = tuple(dataset_list) if dataset_list is not None else tuple(DATASET_LIST)
    default = ff_type == "spectral" and chg_spin_emb_type == "rand_emb" and dl == tuple(DATASET_LIST)
    out = WeightSet(meta={"source": f"synthetic(seed={int(seed)})" + ("" if default else f"[ff={ff_type},emb={chg_spin_emb_type},datasets={','.join(dl)}]")})
    if not default:
        out.meta["model"] = {"ff_type": ff_type, "chg_spin_emb_type": chg_spin_emb_type, "dataset_list": list(dl)}
    base = param_shapes()
    shapes = param_shapes(ff_type, chg_spin_emb_type, len(dl), grid_bias=grid_bias)
    # tensors shared with the default variant keep the default variant's values: draw the default set first, in its own order
    drawn: Dict[str, np.ndarray] = {}
    for name, shape in list(base.items()) + [(n, sh) for n, sh in shapes.items() if n not in base or tuple(sh) != tuple(base[n])]:
        if name in base and name in drawn:                       # a variant tensor with a default-variant name but another shape
            rng_v = np.random.default_rng([seed, sum(name.encode())])
            drawn[name] = _draw(rng_v, name, shape)
        elif name in base:
            drawn[name] = _draw(rng, name, shape)
        elif name in ("so3_grid.to_grid_mat", "so3_grid.from_grid_mat"):
            tg, fg = synthetic_grid_matrices()
            drawn[name] = tg if name.endswith("to_grid_mat") else fg
        else:
            drawn[name] = _draw(np.random.default_rng([seed, sum(name.encode())]), name, shape)
    for name in shapes:
        out[name] = drawn[name]
    return out


def _draw(rng, name: str, shape) -> np.ndarray:
    leaf = name.split(".")[-1]
    if name == "normalizer.rmsd":
        a = np.array([1.5])
    elif name == "element_refs":
        z = np.arange(MAX_NUM_ELEMENTS, dtype=np.float64)
        a = -13.6 * z ** 1.2 + rng.standard_normal(MAX_NUM_ELEMENTS)
    elif "embedding" in name and leaf in ("weight", "W") and "rad_func" not in name:
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
    return np.ascontiguousarray(a, dtype=np.float32)


def pack_blob(weights: Dict[str, np.ndarray], meta: Optional[Dict[str, Any]] = None) -> bytes:
    """Serialise a name->array dict into the UMXW0001 blob read by ``umx_load_weights`` (``meta`` or ``weights.meta``
    goes into the JSON trailer)."""
    shapes = param_shapes(**variant_of(weights))
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


def _blob_data_end(blob: bytes) -> int:
    """Offset just behind the tensor data section, computed from the header's entry table (where a JSON trailer starts if there is one)."""
    if blob[:8] != MAGIC:
        raise ValueError("not a UMXW0001 weight blob")
    n, _ = struct.unpack_from("<II", blob, 8)
    esz = struct.calcsize("<96sI4IQQ")
    pos = 16 + n * esz
    data0 = pos + ((-pos) % 64)
    end = data0
    for i in range(n):
        off, nb = struct.unpack_from("<QQ", blob, 16 + i * esz + 96 + 20)
        end = max(end, data0 + off + nb + ((-nb) % 64))
    return end


def blob_meta(blob: bytes) -> Dict[str, Any]:
    """The JSON trailer of a blob ({} when it has none) without copying the tensors.  The trailer is located from the header's tensor
    extent -- a search for the magic could match bytes inside the tensor data."""
    try:
        at = _blob_data_end(blob)
    except Exception:
        return {}
    if blob[at:at + 8] != META_MAGIC:
        return {}
    try:
        (n_js,) = struct.unpack_from("<I", blob, at + 8)
        return json.loads(blob[at + 12:at + 12 + n_js].decode())
    except Exception:
        return {}


def save_weights(path: str, weights: Dict[str, np.ndarray]) -> None:
    with open(path, "wb") as f:
        f.write(pack_blob(weights))


def load_weights(path: str) -> "WeightSet":
    with open(path, "rb") as f:
        return unpack_blob(f.read())
