#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/p
cd /tmp && export TMPDIR=/tmp && cd $R
for k in 2 4 8; do
timeout -k 10 300 python bench.py --images $k --steps 10 --warmup 2 --no-cpu-baseline --no-fp32-mode 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print(f\"images=$k: {d['ms_per_step']:.2f} ms/step, GEMM {r['ms_per_step']:.2f}, other-gemm {r['other_gemm_family']['ms_per_step']:.2f}, rest {r['hbm_regime']['ms_per_step']:.2f}\")"
done
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p/kt -o kt -f csv -- python3 bench.py --images 2 --steps 10 --warmup 2 --no-cpu-baseline --no-fp32-mode > $R/gpurun_out/p/kt.log 2>&1
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/p/kt/kt_kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("sum of kernel time per iteration (12 its):", tot/12/1e6, "ms")
tr=list(csv.DictReader(open("gpurun_out/p/kt/kt_kernel_trace.csv")))
tr.sort(key=lambda r:int(r['Start_Timestamp']))
# gaps within last iteration
starts=[i for i,r in enumerate(tr) if 'k_graph_count' in r['Kernel_Name']]
seg=tr[starts[-2]:starts[-1]]
span=(int(seg[-1]['End_Timestamp'])-int(seg[0]['Start_Timestamp']))/1e6
busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in seg)/1e6
print(f"one iteration: span {span:.2f} ms, kernel busy {busy:.2f} ms, {len(seg)} launches, idle {span-busy:.2f} ms")
gaps=sorted(((int(seg[i+1]['Start_Timestamp'])-int(seg[i]['End_Timestamp']))/1e3, seg[i]['Kernel_Name'][:40], seg[i+1]['Kernel_Name'][:40]) for i in range(len(seg)-1))[-8:]
for g in gaps: print("   gap %.1f us after %s before %s"%g)
PY
