#!/bin/bash
# c1 / c2 / 2-image shard latency: default build vs $1
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s
for v in A B A B; do
  if [ $v = A ]; then unset UMX_LIBRARY UMX_ALLOW_STALE; else export UMX_LIBRARY=$GRAFT_REPO_ROOT/$1 UMX_ALLOW_STALE=1; fi
  echo "== $v ${UMX_LIBRARY:-default}"
  python3 tools/gpu_eval_config.py c1 100 2>&1 | grep -v amdgpu.ids
  python3 tools/gpu_eval_config.py c2 20 2>&1 | grep -v amdgpu.ids
  python3 tools/gpu_eval_config.py c3-shard 10 2>&1 | grep -v amdgpu.ids
done
