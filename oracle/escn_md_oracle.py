"""CPU oracle: UMA-S (eSCN-MD) energy + forces, restated in plain PyTorch (float64 by default).

TEST INFRASTRUCTURE ONLY.  Nothing under ``pdb2reaction_amd/`` may import this module; it is the
checker for the HIP engine (tests/, ``__graft_entry__.smoke()``, ``bench.py``'s cpu_baseline leg).

PARITY UNPINNED.  The arithmetic of the reference's hot path lives in the third-party package
``fairchem-core`` (unpinned, ``pyproject.toml:14`` of the reference), reached through
``self.predict.predict(batch)`` at reference ``pdb2reaction/uma_pysis.py:373,385``.  fairchem is not
installed here, the UMA checkpoint is a gated download, and the reference ships no tests or golden
vectors (SURVEY.md sections 4 and 8c).  This file therefore restates the *published* eSCN-MD
algorithm (fairchem-core 2.x ``models/uma/escn_md.py`` and friends, as summarised in SURVEY.md
Appendix A) and is pinned only by physical invariants (tests/test_oracle.py): rotation /
translation / permutation invariance of E, F = -dE/dx against central differences, sum(F) = 0,
and independence from the choice of edge-frame roll angle.

Stage map:  K1 radius graph -> :func:`radius_graph`;  K2 edge frames /
Wigner-D -> :func:`edge_rotation`, :func:`wigner_blocks`;  K3 edge scalars -> :func:`edge_scalars`;
K4 node init;  K5 edge-degree embedding;  K6 RMS-norm-SH -> :func:`rms_norm_sh`;  K7 Edgewise ->
:func:`so2_conv`, :func:`edgewise`;  K8 atom-wise FF (spectral or grid) -> :func:`atomwise`, :func:`grid_atomwise`;  K9 energy
readout;  K10 forces by autograd;  K11 normaliser + element references.

Conventions (ours; any consistent real-SH convention yields the same E/F because the SO(2)
convolution commutes with rotations about the edge axis when mmax == lmax):
  * l=1 basis (m=-1,0,1) = (x, y, z); the polar axis is y, so D^1(R) = R.
  * l=2 basis (m=-2..2) = sqrt3*xz, sqrt3*xy, y^2-(x^2+z^2)/2, sqrt3*yz, (sqrt3/2)(z^2-x^2).
  * edge e = (j -> i): source j, target i, vec = pos[j] - pos[i]; R_e maps vec/|vec| onto +y.
  * m-primary rows: [ (l0,m0) (l1,m0) (l2,m0) | (l1,+1) (l2,+1) (l1,-1) (l2,-1) | (l2,+2) (l2,-2) ].
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch

from . import tables as W    # literal constants of the oracle (NOT the product's tables: see tables.py)

C = W.SPHERE_CHANNELS
H = W.HIDDEN_CHANNELS
S = W.NUM_SPH
TO_M = list(W.TO_M)

_S3 = math.sqrt(3.0)
# symmetric traceless quadratic forms of the l=2 real harmonics, Y_a(r) = r^T A_a r, <A_a,A_b> = 1.5 delta
_A2 = np.zeros((5, 3, 3))
_A2[0, 0, 2] = _A2[0, 2, 0] = _S3 / 2          # sqrt3 x z
_A2[1, 0, 1] = _A2[1, 1, 0] = _S3 / 2          # sqrt3 x y
_A2[2] = np.diag([-0.5, 1.0, -0.5])            # y^2 - (x^2+z^2)/2
_A2[3, 1, 2] = _A2[3, 2, 1] = _S3 / 2          # sqrt3 y z
_A2[4] = np.diag([-_S3 / 2, 0.0, _S3 / 2])     # (sqrt3/2)(z^2 - x^2)


# ------------------------------------------------------------------------------------------------
# K1  radius graph (fairchem generate_graph, otf_graph=True, no PBC; reference uma_pysis.py:313-322)
# ------------------------------------------------------------------------------------------------
def radius_graph(pos: torch.Tensor, cutoff: float, max_neigh: Optional[int] = None):
    """All ordered pairs (j -> i), 0 < |r_j - r_i| <= cutoff, sorted by (target i, source j).

    Returns (src [E], dst [E]).  ``max_neigh`` keeps the nearest M sources per target.
    """
    with torch.no_grad():
        n = pos.shape[0]
        p = pos.detach().to(torch.float64)
        d2 = ((p[:, None, :] - p[None, :, :]) ** 2).sum(-1)
        mask = (d2 <= cutoff * cutoff) & ~torch.eye(n, dtype=torch.bool)
        if max_neigh is not None:
            d2m = torch.where(mask, d2, torch.full_like(d2, float("inf")))
            order = torch.argsort(d2m, dim=1, stable=True)
            rank = torch.empty_like(order)
            rank.scatter_(1, order, torch.arange(n).expand(n, n))
            mask = mask & (rank < max_neigh)
        dst, src = torch.nonzero(mask, as_tuple=True)      # row = target i, col = source j
    return src, dst


# ------------------------------------------------------------------------------------------------
# K2  edge frames and Wigner-D blocks
# ------------------------------------------------------------------------------------------------
def edge_rotation(nhat: torch.Tensor, roll: Optional[torch.Tensor] = None) -> torch.Tensor:
    """R (E,3,3) with R @ nhat = +y.  Minimal rotation about nhat x y; for nhat_y < -0.9 the
    vector is first flipped by F = diag(1,-1,-1).  ``roll`` (E,) adds a rotation about y (gauge)."""
    e = nhat.shape[0]
    flip = nhat[:, 1] < -0.9
    sgn = torch.where(flip, -torch.ones_like(nhat[:, 1]), torch.ones_like(nhat[:, 1]))
    n = torch.stack([nhat[:, 0], nhat[:, 1] * sgn, nhat[:, 2] * sgn], dim=1)     # F nhat
    nx, ny, nz = n[:, 0], n[:, 1], n[:, 2]
    k = 1.0 / (1.0 + ny)
    # rows of the minimal rotation taking n to y
    r0 = torch.stack([1.0 - k * nx * nx, -nx, -k * nx * nz], dim=1)
    r1 = torch.stack([nx, ny, nz], dim=1)
    r2 = torch.stack([-k * nx * nz, -nz, 1.0 - k * nz * nz], dim=1)
    rm = torch.stack([r0, r1, r2], dim=1)                                          # (E,3,3)
    fdiag = torch.stack([torch.ones_like(sgn), sgn, sgn], dim=1)                   # F
    rm = rm * fdiag[:, None, :]                                                    # R = R' F
    if roll is not None:
        c, s = torch.cos(roll), torch.sin(roll)
        z, o = torch.zeros_like(c), torch.ones_like(c)
        ry = torch.stack([torch.stack([c, z, s], 1), torch.stack([z, o, z], 1), torch.stack([-s, z, c], 1)], 1)
        rm = ry @ rm
    assert rm.shape == (e, 3, 3)
    return rm


def wigner_blocks(rm: torch.Tensor):
    """D^1 = R (E,3,3) and D^2 (E,5,5) with D2[a,b] = (2/3) <A_a, R A_b R^T>."""
    a2 = torch.as_tensor(_A2, dtype=rm.dtype)
    m = torch.einsum("eik,bkl,ejl->ebij", rm, a2, rm)          # R A_b R^T
    d2 = (2.0 / 3.0) * torch.einsum("aij,ebij->eab", a2, m)
    return rm, d2


def wigner_m_primary(rm: torch.Tensor) -> torch.Tensor:
    """(E,9,9) block-diagonal Wigner matrix with rows permuted to m-primary order."""
    e = rm.shape[0]
    d1, d2 = wigner_blocks(rm)
    wig = torch.zeros(e, S, S, dtype=rm.dtype)
    wig[:, 0, 0] = 1.0
    wig[:, 1:4, 1:4] = d1
    wig[:, 4:9, 4:9] = d2
    return wig[:, TO_M, :]


def envelope(u: torch.Tensor) -> torch.Tensor:
    """PolynomialEnvelope(exponent=5): 1 - 21 u^5 + 35 u^6 - 15 u^7 for u < 1, else 0."""
    env = 1.0 - 21.0 * u ** 5 + 35.0 * u ** 6 - 15.0 * u ** 7
    return torch.where(u < 1.0, env, torch.zeros_like(u))


# ------------------------------------------------------------------------------------------------
# small building blocks
# ------------------------------------------------------------------------------------------------
def silu(x):
    return x * torch.sigmoid(x)


def layer_norm(x, w, b):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + W.LN_EPS) * w + b


def radial_mlp(p: Dict[str, torch.Tensor], prefix: str, x_edge: torch.Tensor) -> torch.Tensor:
    """RadialMLP: Linear -> LayerNorm -> SiLU -> Linear -> LayerNorm -> SiLU -> Linear."""
    h = x_edge @ p[f"{prefix}.fc1.weight"].T + p[f"{prefix}.fc1.bias"]
    h = silu(layer_norm(h, p[f"{prefix}.ln1.weight"], p[f"{prefix}.ln1.bias"]))
    h = h @ p[f"{prefix}.fc2.weight"].T + p[f"{prefix}.fc2.bias"]
    h = silu(layer_norm(h, p[f"{prefix}.ln2.weight"], p[f"{prefix}.ln2.bias"]))
    return h @ p[f"{prefix}.fc3.weight"].T + p[f"{prefix}.fc3.bias"]


def rms_norm_sh(x: torch.Tensor, aw: torch.Tensor, ab: torch.Tensor) -> torch.Tensor:
    """K6: EquivariantRMSNormArraySphericalHarmonicsV2 (component norm, centering, balanced degrees)."""
    l_of = torch.tensor(W.L_OF_LP)
    x0 = x[:, 0:1, :] - x[:, 0:1, :].mean(dim=2, keepdim=True)
    feat = torch.cat([x0, x[:, 1:, :]], dim=1)
    bal = 1.0 / ((2.0 * l_of.to(x.dtype) + 1.0) * (W.LMAX + 1))           # (9,)
    fn = (feat ** 2 * bal[None, :, None]).sum(dim=1, keepdim=True)         # (N,1,C)
    fn = fn.mean(dim=2, keepdim=True)                                      # (N,1,1)
    fn = (fn + W.NORM_EPS) ** -0.5
    out = feat * fn * aw[l_of][None, :, :]
    out = torch.cat([out[:, 0:1, :] + ab[None, None, :], out[:, 1:, :]], dim=1)
    return out


def so2_conv(p, prefix: str, x: torch.Tensor, rad: Optional[torch.Tensor], c_in: int, c_out: int, extra: int):
    """SO2_Convolution on m-primary (E,9,c_in) input; returns ((E,9,c_out), gate scalars or None)."""
    e = x.shape[0]
    x0 = x[:, 0:3, :].reshape(e, 3 * c_in)
    if rad is not None:
        x0 = x0 * rad[:, : 3 * c_in]
    y0 = x0 @ p[f"{prefix}.fc_m0.weight"].T + p[f"{prefix}.fc_m0.bias"]
    gate = None
    if extra:
        gate = y0[:, :extra]
        y0 = y0[:, extra:]
    out = [y0.reshape(e, 3, c_out)]
    off, off_rad = 3, 3 * c_in
    for m in (1, 2):
        nl = W.LMAX - m + 1
        xm = x[:, off: off + 2 * nl, :].reshape(e, 2, nl * c_in)           # row 0 = +m (real), row 1 = -m (imag)
        if rad is not None:
            xm = xm * rad[:, off_rad: off_rad + nl * c_in][:, None, :]
        ym = xm @ p[f"{prefix}.so2_m_conv.{m - 1}.fc.weight"].T            # (E,2,2*nl*c_out)
        half = nl * c_out
        yr, yi = ym[:, :, :half], ym[:, :, half:]
        y_real = yr[:, 0] - yi[:, 1]
        y_imag = yr[:, 1] + yi[:, 0]
        out.append(torch.stack([y_real, y_imag], dim=1).reshape(e, 2 * nl, c_out))
        off += 2 * nl
        off_rad += nl * c_in
    return torch.cat(out, dim=1), gate


def gate_m_primary(gate: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """GateActivation on m-primary rows: row 0 -> SiLU, rows of degree l>0 * sigmoid(gate_l)."""
    e = x.shape[0]
    g = torch.sigmoid(gate).reshape(e, W.LMAX, H)
    l_of = torch.tensor(W.L_OF_MP[1:]) - 1
    return torch.cat([silu(x[:, 0:1, :]), x[:, 1:, :] * g[:, l_of, :]], dim=1)


def grid_atomwise(p, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """K8, ``ff_type = "grid"`` ([3P-UNVERIFIED]): GridAtomwise = project the coefficients
    onto the S2 grid (``to_grid_mat`` (G,9): x_grid[n,g,c] = sum_i T[g,i] x[n,i,c]), a point-wise 3-layer SiLU MLP over the channels
    (C -> H -> H -> C, bias-free in fairchem; a bias tensor is honoured when the weight set carries one), project back
    (``from_grid_mat`` (G,9): y[n,i,c] = sum_g F[g,i] o[n,g,c]).  The two matrices are DATA of the weight set (buffers of the
    checkpoint's SO3_Grid), never restated here."""
    tg, fg = p["so3_grid.to_grid_mat"], p["so3_grid.from_grid_mat"]
    h = torch.einsum("gi,nic->ngc", tg, x)
    for k, li in enumerate((0, 2, 4)):
        h = h @ p[f"{prefix}.grid_mlp.{li}.weight"].T
        if f"{prefix}.grid_mlp.{li}.bias" in p:
            h = h + p[f"{prefix}.grid_mlp.{li}.bias"]
        if k < 2:
            h = silu(h)
    return torch.einsum("gi,ngc->nic", fg, h)


def atomwise(p, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """K8: the atom-wise feed-forward of the weight set -- GridAtomwise when it carries ``grid_mlp`` tensors, else
    SpectralAtomwise = scalar MLP gates, SO3_Linear -> gate -> SO3_Linear (l-primary)."""
    if f"{prefix}.grid_mlp.0.weight" in p:
        return grid_atomwise(p, prefix, x)
    n = x.shape[0]
    l_of = torch.tensor(W.L_OF_LP)
    gs = silu(x[:, 0, :] @ p[f"{prefix}.scalar_mlp.weight"].T + p[f"{prefix}.scalar_mlp.bias"])
    w1 = p[f"{prefix}.so3_linear_1.weight"][l_of]                          # (9,H,C)
    h = torch.einsum("nmi,moi->nmo", x, w1)
    h = torch.cat([h[:, 0:1, :] + p[f"{prefix}.so3_linear_1.bias"][None, None, :], h[:, 1:, :]], dim=1)
    g = torch.sigmoid(gs).reshape(n, W.LMAX, H)
    h = torch.cat([silu(h[:, 0:1, :]), h[:, 1:, :] * g[:, l_of[1:] - 1, :]], dim=1)
    w2 = p[f"{prefix}.so3_linear_2.weight"][l_of]
    o = torch.einsum("nmi,moi->nmo", h, w2)
    o = torch.cat([o[:, 0:1, :] + p[f"{prefix}.so3_linear_2.bias"][None, None, :], o[:, 1:, :]], dim=1)
    return o


# ------------------------------------------------------------------------------------------------
# the model
# ------------------------------------------------------------------------------------------------
class Oracle:
    """UMA-S forward / forces for ONE system (the reference evaluates one image per call)."""

    def __init__(self, weights: Dict[str, np.ndarray], dtype=torch.float64, cutoff: float = W.CUTOFF,
                 max_neigh: Optional[int] = W.MAX_NEIGHBORS, dataset_list=None):
        """``dataset_list``: the order of the rows of ``dataset_embedding.weight`` (the checkpoint's ``dataset_list``); default: the
        weight set's own record (``weights.meta["model"]["dataset_list"]``) or the UMA order of tables.py."""
        self.dtype = dtype
        meta = (getattr(weights, "meta", None) or {}).get("model") or {}
        self.dataset_list = tuple(dataset_list or meta.get("dataset_list") or W.DATASET_LIST)
        self.cutoff = float(cutoff)
        self.max_neigh = max_neigh
        self.p = {k: torch.as_tensor(np.asarray(v), dtype=dtype) for k, v in weights.items()}
        self.refs64 = torch.as_tensor(np.asarray(weights["element_refs"]), dtype=torch.float64)
        self.debug: Dict[str, torch.Tensor] = {}

    # -- K3 ---------------------------------------------------------------------------------------
    def edge_scalars(self, dist, z_src, z_dst):
        mu = torch.linspace(0.0, self.cutoff, W.NUM_DISTANCE_BASIS, dtype=torch.float64).to(self.dtype)
        coeff = -0.5 / (2.0 * (self.cutoff / (W.NUM_DISTANCE_BASIS - 1))) ** 2
        gauss = torch.exp(coeff * (dist[:, None] - mu[None, :]) ** 2)
        return torch.cat([gauss, self.p["source_embedding.weight"][z_src], self.p["target_embedding.weight"][z_dst]], dim=1)

    def charge_spin_embedding(self, which: str, value: int):
        """ChgSpinEmbedding (fairchem ``chg_spin_emb_type`` [3P-UNVERIFIED]) -- the form is read off the
        tensors the weight set carries: ``rand_emb`` = a lookup table indexed by charge + 100 / by the multiplicity;
        ``pos_emb`` = [sin(2 pi v W), cos(2 pi v W)] with a fixed frequency vector W (C/2), the null spin 0 embedding to zero;
        ``lin_emb`` = Linear(1 -> C) of the value (null spin 0 -> -100)."""
        p = self.p
        if f"{which}_embedding.W" in p:
            ang = 2.0 * math.pi * float(value) * p[f"{which}_embedding.W"]
            emb = torch.cat([torch.sin(ang), torch.cos(ang)])
            return torch.zeros_like(emb) if (which == "spin" and value == 0) else emb
        if f"{which}_embedding.lin_emb.weight" in p:
            v = -100.0 if (which == "spin" and value == 0) else float(value)
            return p[f"{which}_embedding.lin_emb.weight"][:, 0] * v + p[f"{which}_embedding.lin_emb.bias"]
        return p[f"{which}_embedding.weight"][value + (W.CHARGE_OFFSET if which == "charge" else 0)]

    def system_embedding(self, charge: int, spin: int, task: str):
        p = self.p
        parts = [self.charge_spin_embedding("charge", charge), self.charge_spin_embedding("spin", spin)]
        if "dataset_embedding.weight" in p:          # (use_dataset_embedding = False: mix_csd takes [charge | spin] only)
            parts.append(p["dataset_embedding.weight"][self.dataset_list.index(task)])
        return silu(p["mix_csd.weight"] @ torch.cat(parts) + p["mix_csd.bias"])

    # -- forward ----------------------------------------------------------------------------------
    def model_energy(self, z: torch.Tensor, pos: torch.Tensor, charge=0, spin=1, task="omol",
                     roll: Optional[torch.Tensor] = None, graph=None, keep: bool = False) -> torch.Tensor:
        """Un-normalised model energy (scalar tensor) of one system; ``pos`` (N,3) Angstrom."""
        p = self.p
        n = pos.shape[0]
        dbg = self.debug if keep else None
        src, dst = graph if graph is not None else radius_graph(pos, self.cutoff, self.max_neigh)
        vec = pos[src] - pos[dst]
        dist = vec.norm(dim=1)
        nhat = vec / dist[:, None]
        rm = edge_rotation(nhat, roll)
        # pole edges: fairchem detaches the frame angles when nhat_y is numerically 1
        pole = torch.isclose(nhat[:, 1], torch.ones_like(nhat[:, 1]))
        if bool(pole.any()):
            rm = torch.where(pole[:, None, None], rm.detach(), rm)
        wig = wigner_m_primary(rm)                         # (E,9,9)
        wig_inv = wig.transpose(1, 2)
        env = envelope(dist / self.cutoff)
        x_edge = self.edge_scalars(dist, z[src], z[dst])
        sys_emb = self.system_embedding(charge, spin, task)

        # K4 node init
        x = torch.zeros(n, S, C, dtype=self.dtype)
        x[:, 0, :] = p["sphere_embedding.weight"][z] + sys_emb[None, :]

        # K5 edge-degree embedding
        rad0 = radial_mlp(p, "edge_degree_embedding.rad_func", x_edge).reshape(-1, 3, C)
        emb = torch.cat([rad0, torch.zeros(len(src), S - 3, C, dtype=self.dtype)], dim=1)
        emb = torch.bmm(wig_inv, emb) * env[:, None, None] / W.DEG_RESCALE
        x = x.index_add(0, dst, emb)
        if dbg is not None:
            dbg.update(src=src, dst=dst, vec=vec, dist=dist, wig=wig, env=env, x0=x, sys_emb=sys_emb)

        for i in range(W.NUM_LAYERS):
            b = f"blocks.{i}"
            xn = rms_norm_sh(x, p[f"{b}.norm_1.affine_weight"], p[f"{b}.norm_1.affine_bias"])
            xn = torch.cat([xn[:, 0:1, :] + sys_emb[None, None, :], xn[:, 1:, :]], dim=1)
            # K7 edgewise
            msg = torch.cat([xn[src], xn[dst]], dim=2)                         # (E,9,2C)
            msg = torch.bmm(wig, msg)
            rad = radial_mlp(p, f"{b}.edge_wise.so2_conv_1.rad_func", x_edge)
            hpre, gate = so2_conv(p, f"{b}.edge_wise.so2_conv_1", msg, rad, 2 * C, H, W.LMAX * H)
            hid = gate_m_primary(gate, hpre)
            out, _ = so2_conv(p, f"{b}.edge_wise.so2_conv_2", hid, None, H, C, 0)
            if dbg is not None:
                dbg[f"xn.{i}"], dbg[f"xrot.{i}"], dbg[f"rad.{i}"] = xn, msg, rad
                dbg[f"hpre.{i}"], dbg[f"gate.{i}"], dbg[f"msg.{i}"] = hpre, gate, out
            out = torch.bmm(wig_inv, out * env[:, None, None])
            x = x + torch.zeros_like(x).index_add(0, dst, out)
            if dbg is not None:
                dbg[f"xmid.{i}"] = x
            # K8 atomwise
            xn2 = rms_norm_sh(x, p[f"{b}.norm_2.affine_weight"], p[f"{b}.norm_2.affine_bias"])
            x = x + atomwise(p, f"{b}.atom_wise", xn2)
            if dbg is not None:
                dbg[f"x.{i}"] = x

        # K9 readout
        xf = rms_norm_sh(x, p["norm.affine_weight"], p["norm.affine_bias"])
        h = silu(xf[:, 0, :] @ p["energy_block.0.weight"].T + p["energy_block.0.bias"])
        h = silu(h @ p["energy_block.2.weight"].T + p["energy_block.2.bias"])
        e_node = (h @ p["energy_block.4.weight"].T + p["energy_block.4.bias"]).reshape(-1)
        if dbg is not None:
            dbg["e_node"] = e_node
        return e_node.sum()

    # -- K10 + K11 --------------------------------------------------------------------------------
    def energy_forces(self, z, pos, charge=0, spin=1, task="omol", forces=True, roll=None, keep=False):
        """Return (E_total eV as python float in f64, F (N,3) eV/A numpy or None).

        E = E_model * rmsd + sum_i ref[Z_i];  F = -dE_model/dpos * rmsd   (SURVEY.md Appendix A.8).
        """
        z = torch.as_tensor(np.asarray(z), dtype=torch.long)
        pos = torch.as_tensor(np.asarray(pos), dtype=self.dtype).clone().requires_grad_(forces)
        e = self.model_energy(z, pos, charge, spin, task, roll=roll, keep=keep)
        rmsd = float(self.p["normalizer.rmsd"][0])
        f = None
        if forces:
            (g,) = torch.autograd.grad(e, pos)
            f = (-g * rmsd).detach().numpy()
        e_tot = float(e.detach().to(torch.float64)) * rmsd + float(self.refs64[z].sum())
        return e_tot, f


def reference_energy_sum(weights: Dict[str, np.ndarray], z) -> float:
    return float(np.asarray(weights["element_refs"], dtype=np.float64)[np.asarray(z)].sum())
