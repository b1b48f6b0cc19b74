"""Per-shape GEMM breakdown from UMX_PROFILE_DUMP (one profiled E+F call of N atoms x K images)."""
import os, sys, collections
import numpy as np
sys.path.insert(0, ".")
dump = "gpurun_out/gemm_dump.csv"
if os.path.exists(dump): os.remove(dump)
os.environ["UMX_PROFILE_DUMP"] = dump
from pdb2reaction_amd import weights as W, synth
from pdb2reaction_amd.engine import Engine
n, k = int(sys.argv[1]), int(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else "split"
os.environ["UMX_PRECISION"] = mode
eng = Engine(0); eng.load_weights(W.make_synthetic_weights(0))
z, imgs, _ = synth.make_images(n, k); eng.set_system(z)
p = imgs.astype(np.float32)
eng.energy_forces(p); eng.energy_forces(p)
import time; t = time.time(); eng.energy_forces(p); wall = (time.time() - t) * 1e3
eng.profile_enable(True); eng.profile_read(True); eng.energy_forces(p); pr = eng.profile_read(True)
rows = collections.OrderedDict()
for line in open(dump):
    M, N, K, am, cx, prec, gz, ms, fl = line.strip().split(",")
    key = (int(M), int(N), int(K), int(am), int(cx), int(prec), int(gz))
    r = rows.setdefault(key, [0, 0.0, 0.0]); r[0] += 1; r[1] += float(ms); r[2] += float(fl)
print(f"mode={mode} N={n} K={k}: wall {wall:.1f} ms, gemm total {pr['gemm_ms']:.1f} ms")
print(f"{'M':>8s} {'N':>5s} {'K':>5s} am cx pr gz  calls      ms   TFLOP/s  %gemm")
for key, (c, ms, fl) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print(f"{key[0]:8d} {key[1]:5d} {key[2]:5d} {key[3]:2d} {key[4]:2d} {key[5]:2d} {key[6]:2d} {c:6d} {ms:8.2f} {fl/ms/1e9:8.1f} {100*ms/pr['gemm_ms']:6.1f}")
