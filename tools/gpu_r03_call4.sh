#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c4
mkdir -p $O
cd $R
for m in split-bf16 fp32; do
  for nf in 0 1; do
    echo "=== UMX_PRECISION=$m UMX_NODE_F64=$nf" >> $O/stage.log
    UMX_PRECISION=$m UMX_NODE_F64=$nf timeout -k 10 600 python3 tools/gpu_stage_bias.py 400 >> $O/stage.log 2>&1 || exit 1
  done
done
grep -v amdgpu.ids $O/stage.log | tail -200
