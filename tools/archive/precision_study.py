"""CPU study: error of split-precision linears (emulated) vs exact float64, forward and reverse pass.
Patches Tensor.__matmul__ (all dense linears of oracle/staged.py go through `@`; rotations use bmm)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from pdb2reaction_amd import weights as W, synth
from oracle.staged import Staged

torch.set_num_threads(8)
MODE = {"fwd": None, "bwd": None, "phase": "fwd"}
orig = torch.Tensor.__matmul__

def split(x, dt, terms, scale=1.0):
    xs = x * scale
    parts, r = [], xs
    for _ in range(terms):
        p = r.to(dt).to(torch.float64)
        parts.append(p)
        r = r - p
    return [p / scale for p in parts]

def mm_f16s(a, b):
    """fp16 hi + fp16 (lo * 2^11), static scale 1: C = ah.bh + 2^-11 (ah.bl + al.bh), fp32 accumulators."""
    ah = a.to(torch.float16).to(torch.float64); al = ((a - ah) * 2048.0).to(torch.float16).to(torch.float64)
    bh = b.to(torch.float16).to(torch.float64); bl = ((b - bh) * 2048.0).to(torch.float16).to(torch.float64)
    hh = orig(ah, bh).to(torch.float32).to(torch.float64)
    cr = (orig(ah, bl) + orig(al, bh)).to(torch.float32).to(torch.float64)
    return hh + cr / 2048.0


def mm_asym(a, b, pa_n, pb_n, order):
    """bf16 split with different plane counts for activations (a) and weights (b); b arrives as W^T (K x N)."""
    BF = torch.bfloat16
    pa, pb = split(a, BF, pa_n), split(b, BF, pb_n)
    out = 0
    for i in range(pa_n):
        for j in range(pb_n):
            if i + j <= order:
                out = out + orig(pa[i], pb[j]).to(torch.float32).to(torch.float64)
    return out


def mm(a, b):
    spec = MODE[MODE["phase"]]
    if isinstance(spec, tuple) and spec and spec[0] == "asym":
        if a.dim() != 2 or b.dim() != 2:
            return orig(a, b)
        return mm_asym(a, b, spec[1], spec[2], spec[3])
    if spec is None or a.dim() != 2 or b.dim() != 2:
        return orig(a, b)
    if spec == "f16s":
        return mm_f16s(a, b)
    if spec == "f32":
        return orig(a.to(torch.float32), b.to(torch.float32)).to(torch.float64)
    dt, nterms, order = spec          # order: max sum of indices of kept cross terms
    sa = float(2.0 ** -torch.floor(torch.log2(a.abs().max().clamp_min(1e-30))).item()) if dt == torch.float16 else 1.0
    sb = float(2.0 ** -torch.floor(torch.log2(b.abs().max().clamp_min(1e-30))).item()) if dt == torch.float16 else 1.0
    pa, pb = split(a, dt, nterms, sa), split(b, dt, nterms, sb)
    out = 0
    for i in range(nterms):
        for j in range(nterms):
            if i + j <= order:
                # fp32 accumulate emulation: round each partial product sum to float32
                out = out + orig(pa[i], pb[j]).to(torch.float32).to(torch.float64)
    return out

torch.Tensor.__matmul__ = mm
w = W.make_synthetic_weights(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
z, pos = synth.make_cluster(n)
pos = pos.astype(np.float32).astype(np.float64)
st = Staged(w)
MODE["phase"] = "fwd"; MODE["fwd"] = None; MODE["bwd"] = None
e0 = float(st.forward(z, pos)); MODE["phase"] = "bwd"; g0 = st.backward().numpy()
F16, BF16 = torch.float16, torch.bfloat16
cfgs = {
    "fwd A2xB3 (5 terms) | bwd bf16x3": (("asym", 2, 3, 2), (BF16, 2, 1)),
    "fwd A3xB3 (6 terms) | bwd bf16x3": (("asym", 3, 3, 2), (BF16, 2, 1)),
    "fwd A2xB3 order1+ (0,2) (4 terms)": (("asym", 2, 3, 2), (BF16, 2, 1)),
    "fp32 matmul (torch CPU)": ("f32", "f32"),
    "fwd fp16x3 scaled-lo static | bwd bf16x3": ("f16s", (BF16, 2, 1)),
    "fwd fp16x3 scaled-lo static | bwd f16s": ("f16s", "f16s"),
    "fwd fp16x3 | bwd fp16x3": ((F16, 2, 1), (F16, 2, 1)),
    "fwd fp16x3 | bwd bf16x3": ((F16, 2, 1), (BF16, 2, 1)),
    "fwd fp16x3 | bwd bf16x2(hi*hi+lo*hi+hi*lo... order0+a_lo)": ((F16, 2, 1), (BF16, 2, 0)),
    "fwd fp16x3 | bwd fp16x1": ((F16, 2, 1), (F16, 1, 0)),
    "fwd fp16x3 | bwd bf16x1": ((F16, 2, 1), (BF16, 1, 0)),
    "fwd bf16x3 | bwd bf16x3": ((BF16, 2, 1), (BF16, 2, 1)),
    "fwd bf16x6 | bwd bf16x3": ((BF16, 3, 2), (BF16, 2, 1)),
    "fwd fp16x1 | bwd fp16x1": ((F16, 1, 0), (F16, 1, 0)),
}
rmsd = 1.5
print(f"N={n} edges={len(st.t['src'])}  E_model={e0:.6f}  max|F|={np.abs(g0).max()*rmsd:.3f}")
for name, (f, b) in cfgs.items():
    MODE["fwd"], MODE["bwd"] = f, b
    MODE["phase"] = "fwd"; e = float(st.forward(z, pos)); MODE["phase"] = "bwd"; g = st.backward().numpy()
    print(f"{name:60s} dE={abs(e-e0)*rmsd:.3e} eV   max dF={np.abs(g-g0).max()*rmsd:.3e} eV/A")
