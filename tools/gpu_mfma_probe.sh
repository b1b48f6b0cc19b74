#!/bin/bash
# Round 6: raw matrix-core results for the offline adder model (tools/mfma_model.py).  Run through gpurun from the repo root.
set -e
mkdir -p gpurun_out/mfma_probe /tmp/mfma_in build
[ -x build/mfma_probe ] || hipcc --offload-arch=gfx950 -O3 -o build/mfma_probe pdb2reaction_amd/csrc/mfma_probe.hip
python tools/mfma_probe_cases.py /tmp/mfma_in
for kind in bf16_32 f16_32 bf16_16 f32_32; do
  for name in single pair tiny16 far16 rand chain; do
    build/mfma_probe $kind /tmp/mfma_in/$name.$kind.in.bin gpurun_out/mfma_probe/$name.$kind.out.bin
  done
done
ls -la gpurun_out/mfma_probe
