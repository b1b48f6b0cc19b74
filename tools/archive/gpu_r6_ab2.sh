#!/bin/bash
# (archive: UMX_LS_NARROW was a round-6 A/B switch of that has been removed again; the numbers are in profiles/r06_ls_ab.txt)
# round 6, second A/B: scope of the aligned planes (all forward products / plain ones only) and the LS form of the 256 x 128 tiles
set -e
mkdir -p gpurun_out/r6b
BIAS_ENVS='[{"UMX_ALIGN_PLANES":"2"},{"UMX_ALIGN_PLANES":"1","UMX_LS_NARROW":"1"},{"UMX_ALIGN_PLANES":"2","UMX_LS_NARROW":"1"}]' python tools/gpu_energy_bias.py w1 c5 g1 perm c3 > gpurun_out/r6b/energy_bias.txt 2>&1
cat gpurun_out/r6b/energy_bias.txt
for rep in 1 2; do
for cfg in "0 2" "1 2" "2 2" "0 1" "1 1" "2 1"; do
  set -- $cfg
  UMX_ALIGN_PLANES=$1 UMX_LS_NARROW=$2 python bench.py --no-shard --no-serial --steps 6 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-fast-mode --driver string > gpurun_out/r6b/bench_al$1_ls$2_$rep.json 2> gpurun_out/r6b/bench_al$1_ls$2_$rep.err || true
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r6b/bench_al$1_ls$2_$rep.json").read().strip().splitlines()[-1]); print("UMX_ALIGN_PLANES=$1 UMX_LS_NARROW=$2 ms_per_step", round(d["ms_per_step"],1), "gemm family ms", round(d["roofline"]["ms_per_step"],1))
except Exception as e: print("bench parse failed", e)
PY
done
done
