"""A/B on the GPU: fp32-MFMA vs split-bf16 (bf16x6 forward / bf16x3 reverse) -- accuracy vs the f64 oracle and speed."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from pdb2reaction_amd import weights as W, synth
from pdb2reaction_amd.engine import Engine
from oracle.escn_md_oracle import Oracle

torch.set_num_threads(16)
w = W.make_synthetic_weights(0)
engs = {}
for mode in ("fp32", "split"):
    os.environ["UMX_PRECISION"] = mode
    e = Engine(0); e.load_weights(w); engs[mode] = e
orc = Oracle(w)
for n in [int(a) for a in sys.argv[1:]] or [60, 250]:
    z, pos = synth.make_cluster(n)
    p32 = pos.astype(np.float32)
    t = time.time(); e_ref, f_ref = orc.energy_forces(z, p32.astype(np.float64)); t = time.time() - t
    for mode, eng in engs.items():
        eng.set_system(z)
        e, f = eng.energy_forces(p32)
        print(f"N={n:5d} {mode:7s} dE={abs(e[0]-e_ref):.3e} eV  max dF={np.abs(f[0]-f_ref).max():.3e} eV/A   (oracle {t:.1f}s)", flush=True)
g = np.load("tests/golden/c2_n500_k2.npz")
for mode, eng in engs.items():
    eng.set_system(g["z"]); e, f = eng.energy_forces(g["pos"])
    print(f"c2 golden N=500 {mode:7s} dE={np.abs(e-g['energy']).max():.3e} eV  max dF={np.abs(f-g['forces']).max():.3e} eV/A", flush=True)
z, imgs, _ = synth.make_images(2000, 4)
p = imgs.astype(np.float32)
res = {}
for mode, eng in engs.items():
    eng.set_system(z)
    res[mode] = eng.energy_forces(p)
    t = time.time()
    for _ in range(3): eng.energy_forces(p)
    dt = (time.time() - t) / 3
    eng.profile_enable(True); eng.profile_read(True); eng.energy_forces(p); pr = eng.profile_read(True); eng.profile_enable(False)
    print(f"N=2000 K=4 {mode:7s}: {dt*1e3:.1f} ms/call ({dt/4*1e3:.1f} ms/image)  gemm {pr['gemm_ms']:.1f} ms  {pr['gemm_flops']/pr['gemm_ms']/1e9:.1f} alg-TFLOP/s", flush=True)
for mode in ("split",):
    print(f"N=2000 {mode} vs fp32: dE={np.abs(res[mode][0]-res['fp32'][0]).max():.3e} eV  max dF={np.abs(res[mode][1]-res['fp32'][1]).max():.3e} eV/A")
