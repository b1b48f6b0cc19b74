"""Staged CPU restatement with a HAND-DERIVED backward (no autograd) -- the kernel-level spec.

TEST INFRASTRUCTURE ONLY (see oracle/escn_md_oracle.py for the parity-unpinned statement).

``Staged.forward`` / ``Staged.backward`` mirror the HIP engine stage by stage (same buffers, same
names as ``umx_debug_fetch``) so that every GPU intermediate -- forward activations *and* the
analytic reverse pass -- has a float64 counterpart.  The reverse pass never
differentiates Wigner matrices: a frame perturbation W -> (1 + w.L) W contributes the per-edge
"torque" tau_k = <g_a, L_k a> - <g_o, L_k o> (a = rotated input, o = local-frame output), and
dE/dvec = dE/dd * nhat + (1/d) R^T (tau_z, 0, -tau_x).  tests/test_oracle.py checks this against
autograd through the explicit Wigner construction of escn_md_oracle.py.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch

from . import tables as W    # literal constants of the oracle (NOT the product's tables: see tables.py)
from . import escn_md_oracle as O

C, H, S = O.C, O.H, O.S
S3 = math.sqrt(3.0)
L_OF_MP = torch.tensor(W.L_OF_MP)
L_OF_LP = torch.tensor(W.L_OF_LP)


def torque(g: torch.Tensor, a: torch.Tensor) -> torch.Tensor:
    """tau_k = sum_ch <g, L_k a> for m-primary (E,9,ch) tensors; returns (E,3) = (x,y,z)."""
    def d(r, c):
        return (g[:, r, :] * a[:, c, :]).sum(-1)
    tx = -d(1, 3) + d(3, 1) + S3 * (d(4, 2) - d(2, 4)) - d(4, 7) + d(7, 4) - d(6, 8) + d(8, 6)
    ty = -d(3, 5) + d(5, 3) - d(4, 6) + d(6, 4) + 2.0 * (d(8, 7) - d(7, 8))
    tz = d(1, 5) - d(5, 1) + S3 * (d(2, 6) - d(6, 2)) + d(4, 8) - d(8, 4) - d(6, 7) + d(7, 6)
    return torch.stack([tx, ty, tz], dim=1)


def silu_grad(x):
    s = torch.sigmoid(x)
    return s * (1.0 + x * (1.0 - s))


def ln_silu_fwd(x, w, b):
    mu = x.mean(-1, keepdim=True)
    rstd = 1.0 / torch.sqrt(((x - mu) ** 2).mean(-1, keepdim=True) + W.LN_EPS)
    y = (x - mu) * rstd * w + b
    return O.silu(y)


def ln_silu_bwd(g_out, x, w, b):
    mu = x.mean(-1, keepdim=True)
    rstd = 1.0 / torch.sqrt(((x - mu) ** 2).mean(-1, keepdim=True) + W.LN_EPS)
    xh = (x - mu) * rstd
    gy = g_out * silu_grad(xh * w + b)
    gw = gy * w
    return rstd * (gw - gw.mean(-1, keepdim=True) - xh * (gw * xh).mean(-1, keepdim=True))


def norm_bwd(g_y, x, aw):
    """Backward of rms_norm_sh w.r.t. x (the l=0 bias and the sys_emb add carry no x-gradient)."""
    bal = 1.0 / ((2.0 * L_OF_LP.to(x.dtype) + 1.0) * (W.LMAX + 1))
    feat = torch.cat([x[:, 0:1, :] - x[:, 0:1, :].mean(2, keepdim=True), x[:, 1:, :]], dim=1)
    q = (feat ** 2 * bal[None, :, None]).sum(dim=(1, 2), keepdim=True) / C
    s = (q + W.NORM_EPS) ** -0.5
    gw = g_y * aw[L_OF_LP][None]
    dot = (gw * feat).sum(dim=(1, 2), keepdim=True)
    g_feat = gw * s - (s ** 3) * dot * bal[None, :, None] * feat / C
    g0 = g_feat[:, 0:1, :] - g_feat[:, 0:1, :].mean(2, keepdim=True)
    return torch.cat([g0, g_feat[:, 1:, :]], dim=1)


# ---- K8 atom-wise feed-forward with its hand-derived reverse (spectral = SO3_Linear -> gate -> SO3_Linear; grid = to-grid ->
#      point-wise SiLU MLP -> from-grid).  Shared by Staged and oracle/chunked.py; `saved` holds what the reverse pass needs.
def atomwise_fwd(p, pa: str, xn2: torch.Tensor):
    n = xn2.shape[0]
    if f"{pa}.grid_mlp.0.weight" in p:
        tg = p["so3_grid.to_grid_mat"]
        bias = [p.get(f"{pa}.grid_mlp.{li}.bias") for li in (0, 2, 4)]
        g1 = torch.einsum("gi,nic->ngc", tg, xn2) @ p[f"{pa}.grid_mlp.0.weight"].T
        g1 = g1 if bias[0] is None else g1 + bias[0]
        g2 = O.silu(g1) @ p[f"{pa}.grid_mlp.2.weight"].T
        g2 = g2 if bias[1] is None else g2 + bias[1]
        g3 = O.silu(g2) @ p[f"{pa}.grid_mlp.4.weight"].T
        g3 = g3 if bias[2] is None else g3 + bias[2]
        return torch.einsum("gi,ngc->nic", p["so3_grid.from_grid_mat"], g3), dict(ffg1=g1, ffg2=g2)
    gs_pre = xn2[:, 0, :] @ p[f"{pa}.scalar_mlp.weight"].T + p[f"{pa}.scalar_mlp.bias"]
    h1 = torch.einsum("nmi,moi->nmo", xn2, p[f"{pa}.so3_linear_1.weight"][L_OF_LP])
    h1 = torch.cat([h1[:, 0:1] + p[f"{pa}.so3_linear_1.bias"][None, None], h1[:, 1:]], dim=1)
    sg = torch.sigmoid(O.silu(gs_pre)).reshape(n, W.LMAX, H)
    hg = torch.cat([O.silu(h1[:, 0:1]), h1[:, 1:] * sg[:, L_OF_LP[1:] - 1]], dim=1)
    o2 = torch.einsum("nmi,moi->nmo", hg, p[f"{pa}.so3_linear_2.weight"][L_OF_LP])
    o2 = torch.cat([o2[:, 0:1] + p[f"{pa}.so3_linear_2.bias"][None, None], o2[:, 1:]], dim=1)
    return o2, dict(gspre=gs_pre, ffh=h1, ffhg=hg)


def atomwise_bwd(p, pa: str, g_o: torch.Tensor, saved) -> torch.Tensor:
    """dE/d(xn2) from dE/d(atom-wise output) (N,9,C)."""
    n = g_o.shape[0]
    if "ffg1" in saved:
        g3 = torch.einsum("gi,nic->ngc", p["so3_grid.from_grid_mat"], g_o)
        g2 = (g3 @ p[f"{pa}.grid_mlp.4.weight"]) * silu_grad(saved["ffg2"])
        g1 = (g2 @ p[f"{pa}.grid_mlp.2.weight"]) * silu_grad(saved["ffg1"])
        return torch.einsum("gi,ngc->nic", p["so3_grid.to_grid_mat"], g1 @ p[f"{pa}.grid_mlp.0.weight"])
    g_hg = torch.einsum("nmo,moi->nmi", g_o, p[f"{pa}.so3_linear_2.weight"][L_OF_LP])
    h1, gs_pre = saved["ffh"], saved["gspre"]
    gs = O.silu(gs_pre)
    sg = torch.sigmoid(gs)
    sgx = sg.reshape(n, W.LMAX, H)[:, L_OF_LP[1:] - 1]
    g_h1 = torch.cat([g_hg[:, 0:1] * silu_grad(h1[:, 0:1]), g_hg[:, 1:] * sgx], dim=1)
    prod = g_hg[:, 1:] * h1[:, 1:]
    g_sg = torch.stack([prod[:, 0:3].sum(1), prod[:, 3:8].sum(1)], dim=1).reshape(n, W.LMAX * H)
    g_gspre = g_sg * sg * (1 - sg) * silu_grad(gs_pre)
    g_xn2 = torch.einsum("nmo,moi->nmi", g_h1, p[f"{pa}.so3_linear_1.weight"][L_OF_LP])
    g_xn2[:, 0, :] = g_xn2[:, 0, :] + g_gspre @ p[f"{pa}.scalar_mlp.weight"]
    return g_xn2


class Staged:
    def __init__(self, weights: Dict[str, np.ndarray], dtype=torch.float64, cutoff=W.CUTOFF):
        self.o = O.Oracle(weights, dtype=dtype, cutoff=cutoff)
        self.p = self.o.p
        self.dtype = dtype
        self.cutoff = cutoff
        self.t: Dict[str, torch.Tensor] = {}
        mu = torch.linspace(0.0, cutoff, W.NUM_DISTANCE_BASIS, dtype=torch.float64).to(dtype)
        self.mu = mu
        self.gcoef = -0.5 / (2.0 * (cutoff / (W.NUM_DISTANCE_BASIS - 1))) ** 2

    # ---- radial MLP with the first layer split into gaussian GEMM + per-element tables -----------
    def radial_fwd(self, prefix, tag):
        p, t = self.p, self.t
        w1 = p[f"{prefix}.fc1.weight"]
        nb = W.NUM_DISTANCE_BASIS
        ts = p["source_embedding.weight"] @ w1[:, nb: nb + W.EDGE_CHANNELS].T
        tt = p["target_embedding.weight"] @ w1[:, nb + W.EDGE_CHANNELS:].T + p[f"{prefix}.fc1.bias"]
        h1 = t["gauss"] @ w1[:, :nb].T + ts[t["zsrc"]] + tt[t["zdst"]]
        a1 = ln_silu_fwd(h1, p[f"{prefix}.ln1.weight"], p[f"{prefix}.ln1.bias"])
        h2 = a1 @ p[f"{prefix}.fc2.weight"].T + p[f"{prefix}.fc2.bias"]
        a2 = ln_silu_fwd(h2, p[f"{prefix}.ln2.weight"], p[f"{prefix}.ln2.bias"])
        t[f"h1pre.{tag}"], t[f"h2pre.{tag}"] = h1, h2
        return a2 @ p[f"{prefix}.fc3.weight"].T + p[f"{prefix}.fc3.bias"]

    def radial_bwd(self, prefix, tag, g_rad):
        """Returns dE/dd contribution (E,) through the gaussian basis."""
        p, t = self.p, self.t
        g_a2 = g_rad @ p[f"{prefix}.fc3.weight"]
        g_h2 = ln_silu_bwd(g_a2, t[f"h2pre.{tag}"], p[f"{prefix}.ln2.weight"], p[f"{prefix}.ln2.bias"])
        t[f"g_a2.{tag}"], t[f"g_h2pre.{tag}"] = g_a2, g_h2
        g_a1 = g_h2 @ p[f"{prefix}.fc2.weight"]
        g_h1 = ln_silu_bwd(g_a1, t[f"h1pre.{tag}"], p[f"{prefix}.ln1.weight"], p[f"{prefix}.ln1.bias"])
        t[f"g_h1pre.{tag}"] = g_h1
        g_gauss = g_h1 @ p[f"{prefix}.fc1.weight"][:, : W.NUM_DISTANCE_BASIS]
        dgauss = t["gauss"] * (2.0 * self.gcoef) * (t["dist"][:, None] - self.mu[None, :])
        return (g_gauss * dgauss).sum(-1)

    # ---- forward ---------------------------------------------------------------------------------
    def forward(self, z, pos, charge=0, spin=1, task="omol"):
        p, t = self.p, self.t
        t.clear()
        z = torch.as_tensor(np.asarray(z), dtype=torch.long)
        pos = torch.as_tensor(np.asarray(pos), dtype=self.dtype)
        n = pos.shape[0]
        src, dst = O.radius_graph(pos, self.cutoff, None)
        vec = pos[src] - pos[dst]
        dist = vec.norm(dim=1)
        nhat = vec / dist[:, None]
        rm = O.edge_rotation(nhat)
        wig = O.wigner_m_primary(rm)
        u = dist / self.cutoff
        env = O.envelope(u)
        denv = torch.where(u < 1.0, (-105.0 * u ** 4 + 210.0 * u ** 5 - 105.0 * u ** 6) / self.cutoff, torch.zeros_like(u))
        t.update(z=z, src=src, dst=dst, vec=vec, dist=dist, nhat=nhat, rm=rm, wig=wig, env=env, denv=denv,
                 zsrc=z[src], zdst=z[dst])
        t["gauss"] = torch.exp(self.gcoef * (dist[:, None] - self.mu[None, :]) ** 2)
        sys_emb = self.o.system_embedding(charge, spin, task)
        t["sys_emb"] = sys_emb
        # node init + edge degree
        x = torch.zeros(n, S, C, dtype=self.dtype)
        x[:, 0, :] = p["sphere_embedding.weight"][z] + sys_emb[None]
        rad0 = self.radial_fwd("edge_degree_embedding.rad_func", "deg").reshape(-1, 3, C)
        t["rad.deg"] = rad0
        emb = torch.cat([rad0, torch.zeros(len(src), S - 3, C, dtype=self.dtype)], dim=1)
        x = x.index_add(0, dst, torch.bmm(wig.transpose(1, 2), emb) * (env / W.DEG_RESCALE)[:, None, None])
        t["x0"] = x
        for i in range(W.NUM_LAYERS):
            b = f"blocks.{i}"
            t[f"xin.{i}"] = x
            xn = O.rms_norm_sh(x, p[f"{b}.norm_1.affine_weight"], p[f"{b}.norm_1.affine_bias"])
            xn = torch.cat([xn[:, 0:1, :] + sys_emb[None, None, :], xn[:, 1:, :]], dim=1)
            xrot = torch.bmm(wig, torch.cat([xn[src], xn[dst]], dim=2))
            rad = self.radial_fwd(f"{b}.edge_wise.so2_conv_1.rad_func", str(i))
            hpre, gate = O.so2_conv(p, f"{b}.edge_wise.so2_conv_1", xrot, rad, 2 * C, H, W.LMAX * H)
            hid = O.gate_m_primary(gate, hpre)
            msg, _ = O.so2_conv(p, f"{b}.edge_wise.so2_conv_2", hid, None, H, C, 0)
            x = x + torch.zeros_like(x).index_add(0, dst, torch.bmm(wig.transpose(1, 2), msg * env[:, None, None]))
            t[f"xn.{i}"], t[f"xrot.{i}"], t[f"rad.{i}"], t[f"hpre.{i}"], t[f"gate.{i}"] = xn, xrot, rad, hpre, gate
            t[f"hid.{i}"], t[f"msg.{i}"], t[f"xmid.{i}"] = hid, msg, x
            # atomwise
            pa = f"{b}.atom_wise"
            xn2 = O.rms_norm_sh(x, p[f"{b}.norm_2.affine_weight"], p[f"{b}.norm_2.affine_bias"])
            o2, saved = atomwise_fwd(p, pa, xn2)
            x = x + o2
            t[f"xn2.{i}"], t[f"x.{i}"], t[f"ffsaved.{i}"] = xn2, x, saved
            for kk, vv in saved.items():
                t[f"{kk}.{i}"] = vv
        xf = O.rms_norm_sh(x, p["norm.affine_weight"], p["norm.affine_bias"])
        pre1 = xf[:, 0, :] @ p["energy_block.0.weight"].T + p["energy_block.0.bias"]
        pre2 = O.silu(pre1) @ p["energy_block.2.weight"].T + p["energy_block.2.bias"]
        e_node = (O.silu(pre2) @ p["energy_block.4.weight"].T + p["energy_block.4.bias"]).reshape(-1)
        t.update(xf=xf, pre1=pre1, pre2=pre2, e_node=e_node)
        return e_node.sum()

    # ---- backward (hand derived) -----------------------------------------------------------------
    def so2_conv_bwd(self, prefix, g_out, c_in, c_out, extra, g_gate=None):
        """g wrt the conv input (E,9,c_in) given g wrt output (E,9,c_out) (and gate scalars)."""
        p = self.p
        e = g_out.shape[0]
        g0 = g_out[:, 0:3, :].reshape(e, 3 * c_out)
        if extra:
            g0 = torch.cat([g_gate, g0], dim=1)
        parts = [(g0 @ p[f"{prefix}.fc_m0.weight"]).reshape(e, 3, c_in)]
        off = 3
        for m in (1, 2):
            nl = W.LMAX - m + 1
            w = p[f"{prefix}.so2_m_conv.{m - 1}.fc.weight"]
            half = nl * c_out
            wa, wb = w[:half], w[half:]
            gr = g_out[:, off: off + nl, :].reshape(e, half)
            gi = g_out[:, off + nl: off + 2 * nl, :].reshape(e, half)
            gxr = gr @ wa + gi @ wb
            gxi = gi @ wa - gr @ wb
            parts.append(torch.stack([gxr, gxi], dim=1).reshape(e, 2 * nl, c_in))
            off += 2 * nl
        return torch.cat(parts, dim=1)

    def backward(self):
        """Returns dE_model/dpos (N,3)."""
        p, t = self.p, self.t
        src, dst, wig, env, denv = t["src"], t["dst"], t["wig"], t["env"], t["denv"]
        n = t["x0"].shape[0]
        ne = len(src)
        dedd = torch.zeros(ne, dtype=self.dtype)
        tau = torch.zeros(ne, 3, dtype=self.dtype)
        # readout
        g_pre2 = p["energy_block.4.weight"].expand(n, H) * silu_grad(t["pre2"])
        g_pre1 = (g_pre2 @ p["energy_block.2.weight"]) * silu_grad(t["pre1"])
        g_xf = torch.zeros(n, S, C, dtype=self.dtype)
        g_xf[:, 0, :] = g_pre1 @ p["energy_block.0.weight"]
        g_x = norm_bwd(g_xf, t[f"x.{W.NUM_LAYERS - 1}"], p["norm.affine_weight"])
        t["g_xfinal"] = g_x
        for i in reversed(range(W.NUM_LAYERS)):
            b = f"blocks.{i}"
            pa = f"{b}.atom_wise"
            # ---- atomwise backward
            g_xn2 = atomwise_bwd(p, pa, g_x, t[f"ffsaved.{i}"])
            g_xmid = g_x + norm_bwd(g_xn2, t[f"xmid.{i}"], p[f"{b}.norm_2.affine_weight"])
            t[f"g_xmid.{i}"] = g_xmid
            # ---- edgewise backward
            gl = torch.bmm(wig, g_xmid[dst])                     # (E,9,C) local frame, pre-env
            msg = t[f"msg.{i}"]
            dedd = dedd + denv * (gl * msg).sum(dim=(1, 2))
            g_msg = gl * env[:, None, None]
            tau = tau - torque(g_msg, msg)
            g_hid = self.so2_conv_bwd(f"{b}.edge_wise.so2_conv_2", g_msg, H, C, 0)
            hpre, gate = t[f"hpre.{i}"], t[f"gate.{i}"]
            sgt = torch.sigmoid(gate)
            sgm = sgt.reshape(ne, W.LMAX, H)[:, L_OF_MP[1:] - 1]
            g_hpre = torch.cat([g_hid[:, 0:1] * silu_grad(hpre[:, 0:1]), g_hid[:, 1:] * sgm], dim=1)
            pr = g_hid[:, 1:] * hpre[:, 1:]
            l1 = (L_OF_MP[1:] == 1)
            g_gate = torch.stack([pr[:, l1].sum(1), pr[:, ~l1].sum(1)], dim=1).reshape(ne, W.LMAX * H) * sgt * (1 - sgt)
            g_y1 = self.so2_conv_bwd(f"{b}.edge_wise.so2_conv_1", g_hpre, 2 * C, H, W.LMAX * H, g_gate)
            xrot, rad = t[f"xrot.{i}"], t[f"rad.{i}"]
            c2 = 2 * C
            radx = torch.cat([rad[:, : 3 * c2].reshape(ne, 3, c2), rad[:, 3 * c2: 5 * c2].reshape(ne, 2, c2),
                              rad[:, 3 * c2: 5 * c2].reshape(ne, 2, c2), rad[:, 5 * c2:].reshape(ne, 1, c2),
                              rad[:, 5 * c2:].reshape(ne, 1, c2)], dim=1)
            gx = g_y1 * xrot
            g_rad = torch.cat([gx[:, 0:3].reshape(ne, -1), (gx[:, 3:5] + gx[:, 5:7]).reshape(ne, -1),
                               (gx[:, 7:8] + gx[:, 8:9]).reshape(ne, -1)], dim=1)
            g_xrot = g_y1 * radx
            tau = tau + torque(g_xrot, xrot)
            dedd = dedd + self.radial_bwd(f"{b}.edge_wise.so2_conv_1.rad_func", str(i), g_rad)
            gb = torch.bmm(wig.transpose(1, 2), g_xrot)            # (E,9,2C) global frame
            g_xn = torch.zeros(n, S, C, dtype=self.dtype).index_add(0, src, gb[:, :, :C]).index_add(0, dst, gb[:, :, C:])
            g_x = g_xmid + norm_bwd(g_xn, t[f"xin.{i}"], p[f"{b}.norm_1.affine_weight"])
            t[f"g_msg.{i}"], t[f"g_hid.{i}"], t[f"g_hpre.{i}"], t[f"g_gate.{i}"] = g_msg, g_hid, g_hpre, g_gate
            t[f"g_xrot.{i}"], t[f"g_rad.{i}"], t[f"g_xn.{i}"], t[f"g_xin.{i}"] = g_xrot, g_rad, g_xn, g_x
        # ---- edge-degree embedding backward (g_x is now dE/dx0)
        gl = torch.bmm(wig, g_x[dst])
        emb = torch.cat([t["rad.deg"], torch.zeros(ne, S - 3, C, dtype=self.dtype)], dim=1)
        dedd = dedd + denv * (gl * emb).sum(dim=(1, 2)) / W.DEG_RESCALE
        g_emb = gl * (env / W.DEG_RESCALE)[:, None, None]
        tau = tau - torque(g_emb, emb)
        t["g_rad.deg"] = g_emb[:, 0:3]
        dedd = dedd + self.radial_bwd("edge_degree_embedding.rad_func", "deg", g_emb[:, 0:3].reshape(ne, 3 * C))
        # ---- assemble dE/dvec and scatter
        pole = torch.isclose(t["nhat"][:, 1], torch.ones_like(t["nhat"][:, 1]))
        tloc = torch.stack([tau[:, 2], torch.zeros_like(tau[:, 0]), -tau[:, 0]], dim=1)
        tloc = torch.where(pole[:, None], torch.zeros_like(tloc), tloc)
        gvec = dedd[:, None] * t["nhat"] + torch.bmm(t["rm"].transpose(1, 2), tloc[:, :, None])[:, :, 0] / t["dist"][:, None]
        t["dedd"], t["tau"], t["gvec"] = dedd, tau, gvec
        return torch.zeros(n, 3, dtype=self.dtype).index_add(0, src, gvec).index_add(0, dst, -gvec)
