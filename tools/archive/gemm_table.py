"""Aggregate a UMX_PROFILE_DUMP CSV (M,N,K,amode,cplx,prec,gz,ms,flops per GEMM launch) into a per-shape table.

    UMX_PROFILE_DUMP=gpurun_out/gemm_dump.csv python bench.py --steps 2 --warmup 1 --no-cpu-baseline
    python tools/gemm_table.py gpurun_out/gemm_dump.csv
"""
import collections
import sys

rows = collections.OrderedDict()
for line in open(sys.argv[1]):
    p = line.strip().split(",")
    if len(p) != 9:
        continue
    M, N, K, amode, cplx, prec, gz = map(int, p[:7])
    ms, fl = float(p[7]), float(p[8])
    if M < (int(sys.argv[2]) if len(sys.argv) > 2 else 100000):
        continue
    k = (M, N, K, amode, cplx, prec, gz)
    r = rows.setdefault(k, [0, 0.0, 0.0])
    r[0] += 1; r[1] += ms; r[2] += fl
tot = sum(r[1] for r in rows.values())
print(f"{'M':>8} {'N':>5} {'K':>5} am cx P gz  calls   ms/call  alg TF/s  exec TF/s  share")
for k, (n, ms, fl) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    M, N, K, amode, cplx, prec, gz = k
    mult = {3: 6, 2: 3, 24: 4, 23: 3}.get(prec, 1)
    print(f"{M:8d} {N:5d} {K:5d} {amode:2d} {cplx:2d} {prec:2d} {gz:2d} {n:6d} {ms/n:9.3f} {fl/ms/1e9:9.1f} {mult*fl/ms/1e9:10.1f} {100*ms/tot:6.1f}%")
