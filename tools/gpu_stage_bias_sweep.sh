#!/bin/bash
# stage-gain sweep over precision modes / dev switches (tools/gpu_stage_bias.py); prints dE and the conv-1 (hg) / conv-2 (msg) gains
N=${1:-600}
for env in "UMX_PRECISION=fp32" "UMX_PRECISION=bf16x3" "UMX_PRECISION=bf16x3 UMX_Q3=0" "UMX_PRECISION=bf16x3 UMX_Q3WIDE=0" "UMX_PRECISION=bf16x3 UMX_MFMA16=0 UMX_Q3=0" "UMX_PRECISION=split" "UMX_PRECISION=bf16x3 UMX_NODE_F64=0"; do
  echo "== $env"
  env $env python tools/gpu_stage_bias.py $N 2>/dev/null | grep -E "^mode|^hg|^msg|^rad\.[0-9]|^e_node|^xmid.3"
done
