"""Per-shape GEMM times from UMX_PROFILE_DUMP files (one line per launch: M,N,K,amode,cplx,prec,gz,ms,flops).
    python3 tools/gemm_shape_times.py a.csv [b.csv]        -> mean ms and executed-equivalent TFLOP/s per (N, K, cplx, prec), side by side"""
import sys
from collections import defaultdict


def load(path):
    acc = defaultdict(lambda: [0, 0.0, 0.0])
    for ln in open(path):
        f = ln.strip().split(",")
        if len(f) != 9:
            continue
        m, n, k, amode, cplx, prec, gz = (int(v) for v in f[:7])
        if m < 100000:
            continue
        key = (n, k, cplx, prec)
        a = acc[key]; a[0] += 1; a[1] += float(f[7]); a[2] += float(f[8])
    return acc


tabs = [load(p) for p in sys.argv[1:]]
keys = sorted(set().union(*[set(t) for t in tabs]), key=lambda q: (q[3] in (2, 3) and q[0] != 0, q))
print("N     K     cplx prec | " + " | ".join(f"{p[-28:]:>28s}" for p in sys.argv[1:]))
tot = [0.0] * len(tabs)
for key in keys:
    cells = []
    for i, t in enumerate(tabs):
        if key in t:
            c, ms, fl = t[key]
            cells.append(f"{c:4d} x {ms / c:7.3f} ms {fl / ms / 1e9:6.0f} TF")
            tot[i] += ms
        else:
            cells.append(" " * 28)
    print(f"{key[0]:5d} {key[1]:5d} {key[2]:4d} {key[3]:4d} | " + " | ".join(cells))
print("total ms: " + "  ".join(f"{v:.1f}" for v in tot))
