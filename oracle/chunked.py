"""Memory-lean float64 energy + forces for LARGE systems: the arithmetic of oracle/staged.py, edge tensors in chunks.

TEST INFRASTRUCTURE ONLY (see oracle/escn_md_oracle.py for the parity-unpinned statement).

``Staged`` keeps every edge-level activation and gradient of all four layers (about 18 GB per layer in float64 at the
c3 size of 142 k directed edges); autograd through ``Oracle`` needs several times that.  ``ChunkedForces`` keeps only
NODE-level tensors across layers and walks the edges in chunks: the forward pass accumulates the aggregated message of a
chunk and drops its edge tensors; the reverse pass re-derives the chunk's forward tensors from the stored node features and
applies the same hand-derived gradient formulas (torque form, no differentiation of Wigner matrices) -- so a 2000-atom
image needs a few GB.  ``tests/test_oracle.py`` holds it to ``Staged`` / autograd at 1e-12 on small systems; it is what
``tools/make_golden_c3.py`` uses to produce the committed c3 / c4 force fixtures.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from . import tables as W
from . import escn_md_oracle as O
from . import staged as ST

C, H, S = O.C, O.H, O.S
L_OF_LP, L_OF_MP = ST.L_OF_LP, ST.L_OF_MP


def radius_graph_blocked(pos: torch.Tensor, cutoff: float, max_neigh: Optional[int], block: int = 2048):
    """Same edge list as ``escn_md_oracle.radius_graph`` (ordered pairs j -> i with 0 < d <= cutoff, sorted by (target, source),
    nearest ``max_neigh`` sources per target with a stable rank on (d^2, source)), built in row blocks so that 20 000 atoms need
    block x N instead of N x N x 3 temporaries."""
    p = pos.detach().to(torch.float64)
    n = p.shape[0]
    srcs, dsts = [], []
    for s0 in range(0, n, block):
        q = p[s0:s0 + block]
        d2 = ((q[:, None, :] - p[None, :, :]) ** 2).sum(-1)                       # (b, N): row = target
        rows = torch.arange(s0, s0 + len(q))
        mask = d2 <= cutoff * cutoff
        mask[torch.arange(len(q)), rows] = False
        if max_neigh is not None:
            d2m = torch.where(mask, d2, torch.full_like(d2, float("inf")))
            order = torch.argsort(d2m, dim=1, stable=True)
            rank = torch.empty_like(order)
            rank.scatter_(1, order, torch.arange(n).expand(len(q), n))
            mask = mask & (rank < max_neigh)
        ti, sj = torch.nonzero(mask, as_tuple=True)
        dsts.append(ti + s0)
        srcs.append(sj)
    return torch.cat(srcs), torch.cat(dsts)


class ChunkedForces:
    def __init__(self, weights: Dict[str, np.ndarray], dtype=torch.float64, cutoff: float = W.CUTOFF, chunk: int = 16384):
        self.st = ST.Staged(weights, dtype=dtype, cutoff=cutoff)
        self.p = self.st.p
        self.dtype, self.cutoff, self.chunk = dtype, float(cutoff), int(chunk)

    # ---- per-chunk pieces -------------------------------------------------------------------------------------------
    def _radial(self, prefix, gauss, zs, zd):
        p = self.p
        w1 = p[f"{prefix}.fc1.weight"]
        nb = W.NUM_DISTANCE_BASIS
        ts = p["source_embedding.weight"] @ w1[:, nb: nb + W.EDGE_CHANNELS].T
        tt = p["target_embedding.weight"] @ w1[:, nb + W.EDGE_CHANNELS:].T + p[f"{prefix}.fc1.bias"]
        h1 = gauss @ w1[:, :nb].T + ts[zs] + tt[zd]
        a1 = ST.ln_silu_fwd(h1, p[f"{prefix}.ln1.weight"], p[f"{prefix}.ln1.bias"])
        h2 = a1 @ p[f"{prefix}.fc2.weight"].T + p[f"{prefix}.fc2.bias"]
        a2 = ST.ln_silu_fwd(h2, p[f"{prefix}.ln2.weight"], p[f"{prefix}.ln2.bias"])
        return a2 @ p[f"{prefix}.fc3.weight"].T + p[f"{prefix}.fc3.bias"], h1, h2

    def _radial_bwd(self, prefix, g_rad, h1, h2, gauss, dist):
        p = self.p
        g_a2 = g_rad @ p[f"{prefix}.fc3.weight"]
        g_h2 = ST.ln_silu_bwd(g_a2, h2, p[f"{prefix}.ln2.weight"], p[f"{prefix}.ln2.bias"])
        g_a1 = g_h2 @ p[f"{prefix}.fc2.weight"]
        g_h1 = ST.ln_silu_bwd(g_a1, h1, p[f"{prefix}.ln1.weight"], p[f"{prefix}.ln1.bias"])
        g_gauss = g_h1 @ p[f"{prefix}.fc1.weight"][:, : W.NUM_DISTANCE_BASIS]
        dgauss = gauss * (2.0 * self.st.gcoef) * (dist[:, None] - self.st.mu[None, :])
        return (g_gauss * dgauss).sum(-1)

    def _edge_fwd(self, i, xn, g):
        """Forward edge tensors of layer i for the chunk geometry g (dict of chunk-local tensors)."""
        b = f"blocks.{i}"
        xrot = torch.bmm(g["wig"], torch.cat([xn[g["src"]], xn[g["dst"]]], dim=2))
        rad, h1, h2 = self._radial(f"{b}.edge_wise.so2_conv_1.rad_func", g["gauss"], g["zs"], g["zd"])
        hpre, gate = O.so2_conv(self.p, f"{b}.edge_wise.so2_conv_1", xrot, rad, 2 * C, H, W.LMAX * H)
        hid = O.gate_m_primary(gate, hpre)
        msg, _ = O.so2_conv(self.p, f"{b}.edge_wise.so2_conv_2", hid, None, H, C, 0)
        return xrot, rad, h1, h2, hpre, gate, msg

    def _chunks(self, geo):
        ne = len(geo["src"])
        for s in range(0, ne, self.chunk):
            sl = slice(s, min(s + self.chunk, ne))
            yield sl, {k: v[sl] for k, v in geo.items()}

    # ---- whole evaluation -------------------------------------------------------------------------------------------
    def energy_forces(self, z, pos, charge=0, spin=1, task="omol", max_neigh: Optional[int] = W.MAX_NEIGHBORS, log=None):
        """(E_total eV, F (N,3) eV/A float64 numpy) -- same contract as ``Oracle.energy_forces``."""
        p, st = self.p, self.st
        z = torch.as_tensor(np.asarray(z), dtype=torch.long)
        pos = torch.as_tensor(np.asarray(pos), dtype=self.dtype)
        n = pos.shape[0]
        with torch.no_grad():
            src, dst = (O.radius_graph if n <= 4096 else radius_graph_blocked)(pos, self.cutoff, max_neigh)
            vec = pos[src] - pos[dst]
            dist = vec.norm(dim=1)
            nhat = vec / dist[:, None]
            rm = O.edge_rotation(nhat)
            u = dist / self.cutoff
            env = O.envelope(u)
            denv = torch.where(u < 1.0, (-105.0 * u ** 4 + 210.0 * u ** 5 - 105.0 * u ** 6) / self.cutoff, torch.zeros_like(u))
            geo = dict(src=src, dst=dst, dist=dist, wig=O.wigner_m_primary(rm), env=env, denv=denv, zs=z[src], zd=z[dst],
                       gauss=torch.exp(st.gcoef * (dist[:, None] - st.mu[None, :]) ** 2))
            ne = len(src)
            sys_emb = st.o.system_embedding(charge, spin, task)
            # ---------------- forward: node-level tensors kept, edge tensors dropped per chunk
            x = torch.zeros(n, S, C, dtype=self.dtype)
            x[:, 0, :] = p["sphere_embedding.weight"][z] + sys_emb[None]
            for _, g in self._chunks(geo):
                rad0, _, _ = self._radial("edge_degree_embedding.rad_func", g["gauss"], g["zs"], g["zd"])
                emb = torch.cat([rad0.reshape(-1, 3, C), torch.zeros(len(g["src"]), S - 3, C, dtype=self.dtype)], dim=1)
                x.index_add_(0, g["dst"], torch.bmm(g["wig"].transpose(1, 2), emb) * (g["env"] / W.DEG_RESCALE)[:, None, None])
            keep = []
            for i in range(W.NUM_LAYERS):
                b = f"blocks.{i}"
                pa = f"{b}.atom_wise"
                xin = x
                xn = O.rms_norm_sh(xin, p[f"{b}.norm_1.affine_weight"], p[f"{b}.norm_1.affine_bias"])
                xn = torch.cat([xn[:, 0:1, :] + sys_emb[None, None, :], xn[:, 1:, :]], dim=1)
                agg = torch.zeros_like(x)
                for _, g in self._chunks(geo):
                    msg = self._edge_fwd(i, xn, g)[-1]
                    agg.index_add_(0, g["dst"], torch.bmm(g["wig"].transpose(1, 2), msg * g["env"][:, None, None]))
                xmid = xin + agg
                xn2 = O.rms_norm_sh(xmid, p[f"{b}.norm_2.affine_weight"], p[f"{b}.norm_2.affine_bias"])
                o2, saved = ST.atomwise_fwd(p, pa, xn2)
                x = xmid + o2
                keep.append(dict(xin=xin, xn=xn, xmid=xmid, ff=saved, x=x))
                if log:
                    log(f"forward layer {i} done")
            xf = O.rms_norm_sh(x, p["norm.affine_weight"], p["norm.affine_bias"])
            pre1 = xf[:, 0, :] @ p["energy_block.0.weight"].T + p["energy_block.0.bias"]
            pre2 = O.silu(pre1) @ p["energy_block.2.weight"].T + p["energy_block.2.bias"]
            e_model = (O.silu(pre2) @ p["energy_block.4.weight"].T + p["energy_block.4.bias"]).reshape(-1).sum()

            # ---------------- reverse pass (formulas of Staged.backward, per chunk with recomputed forward tensors)
            dedd = torch.zeros(ne, dtype=self.dtype)
            tau = torch.zeros(ne, 3, dtype=self.dtype)
            g_pre2 = p["energy_block.4.weight"].expand(n, H) * ST.silu_grad(pre2)
            g_pre1 = (g_pre2 @ p["energy_block.2.weight"]) * ST.silu_grad(pre1)
            g_xf = torch.zeros(n, S, C, dtype=self.dtype)
            g_xf[:, 0, :] = g_pre1 @ p["energy_block.0.weight"]
            g_x = ST.norm_bwd(g_xf, x, p["norm.affine_weight"])
            for i in reversed(range(W.NUM_LAYERS)):
                b = f"blocks.{i}"
                pa = f"{b}.atom_wise"
                k = keep[i]
                g_xn2 = ST.atomwise_bwd(p, pa, g_x, k["ff"])
                g_xmid = g_x + ST.norm_bwd(g_xn2, k["xmid"], p[f"{b}.norm_2.affine_weight"])
                g_xn = torch.zeros(n, S, C, dtype=self.dtype)
                c2 = 2 * C
                for sl, g in self._chunks(geo):
                    nc = len(g["src"])
                    xrot, rad, rh1, rh2, hpre, gate, msg = self._edge_fwd(i, k["xn"], g)
                    gl = torch.bmm(g["wig"], g_xmid[g["dst"]])
                    dedd[sl] += g["denv"] * (gl * msg).sum(dim=(1, 2))
                    g_msg = gl * g["env"][:, None, None]
                    tau[sl] -= ST.torque(g_msg, msg)
                    g_hid = st.so2_conv_bwd(f"{b}.edge_wise.so2_conv_2", g_msg, H, C, 0)
                    sgt = torch.sigmoid(gate)
                    sgm = sgt.reshape(nc, W.LMAX, H)[:, L_OF_MP[1:] - 1]
                    g_hpre = torch.cat([g_hid[:, 0:1] * ST.silu_grad(hpre[:, 0:1]), g_hid[:, 1:] * sgm], dim=1)
                    pr = g_hid[:, 1:] * hpre[:, 1:]
                    l1 = (L_OF_MP[1:] == 1)
                    g_gate = torch.stack([pr[:, l1].sum(1), pr[:, ~l1].sum(1)], dim=1).reshape(nc, W.LMAX * H) * sgt * (1 - sgt)
                    g_y1 = st.so2_conv_bwd(f"{b}.edge_wise.so2_conv_1", g_hpre, 2 * C, H, W.LMAX * H, g_gate)
                    radx = torch.cat([rad[:, : 3 * c2].reshape(nc, 3, c2), rad[:, 3 * c2: 5 * c2].reshape(nc, 2, c2),
                                      rad[:, 3 * c2: 5 * c2].reshape(nc, 2, c2), rad[:, 5 * c2:].reshape(nc, 1, c2),
                                      rad[:, 5 * c2:].reshape(nc, 1, c2)], dim=1)
                    gx = g_y1 * xrot
                    g_rad = torch.cat([gx[:, 0:3].reshape(nc, -1), (gx[:, 3:5] + gx[:, 5:7]).reshape(nc, -1),
                                       (gx[:, 7:8] + gx[:, 8:9]).reshape(nc, -1)], dim=1)
                    g_xrot = g_y1 * radx
                    tau[sl] += ST.torque(g_xrot, xrot)
                    dedd[sl] += self._radial_bwd(f"{b}.edge_wise.so2_conv_1.rad_func", g_rad, rh1, rh2, g["gauss"], g["dist"])
                    gb = torch.bmm(g["wig"].transpose(1, 2), g_xrot)
                    g_xn.index_add_(0, g["src"], gb[:, :, :C])
                    g_xn.index_add_(0, g["dst"], gb[:, :, C:])
                g_x = g_xmid + ST.norm_bwd(g_xn, k["xin"], p[f"{b}.norm_1.affine_weight"])
                if log:
                    log(f"backward layer {i} done")
            for sl, g in self._chunks(geo):
                nc = len(g["src"])
                rad0, rh1, rh2 = self._radial("edge_degree_embedding.rad_func", g["gauss"], g["zs"], g["zd"])
                gl = torch.bmm(g["wig"], g_x[g["dst"]])
                emb = torch.cat([rad0.reshape(-1, 3, C), torch.zeros(nc, S - 3, C, dtype=self.dtype)], dim=1)
                dedd[sl] += g["denv"] * (gl * emb).sum(dim=(1, 2)) / W.DEG_RESCALE
                g_emb = gl * (g["env"] / W.DEG_RESCALE)[:, None, None]
                tau[sl] -= ST.torque(g_emb, emb)
                dedd[sl] += self._radial_bwd("edge_degree_embedding.rad_func", g_emb[:, 0:3].reshape(nc, 3 * C), rh1, rh2, g["gauss"], g["dist"])
            pole = torch.isclose(nhat[:, 1], torch.ones_like(nhat[:, 1]))
            tloc = torch.stack([tau[:, 2], torch.zeros_like(tau[:, 0]), -tau[:, 0]], dim=1)
            tloc = torch.where(pole[:, None], torch.zeros_like(tloc), tloc)
            gvec = dedd[:, None] * nhat + torch.bmm(rm.transpose(1, 2), tloc[:, :, None])[:, :, 0] / dist[:, None]
            grad = torch.zeros(n, 3, dtype=self.dtype).index_add(0, src, gvec).index_add(0, dst, -gvec)
        rmsd = float(p["normalizer.rmsd"][0])
        e_tot = float(e_model.to(torch.float64)) * rmsd + float(st.o.refs64[z].sum())
        return e_tot, (-grad * rmsd).numpy()
