#!/bin/bash
# round 6: replay a mismatching GEMM output element instruction by instruction on the probe (tools/mfma_chain_replay.py)
set -e
mkdir -p gpurun_out/r6k /tmp/chain build
[ -x build/mfma_probe ] || hipcc --offload-arch=gfx950 -O3 -o build/mfma_probe pdb2reaction_amd/csrc/mfma_probe.hip
UMX_MODEL_DUMP=/tmp/conv_dump.npz python -m pytest tests/test_gpu_mfma_model.py -q -k so2 > /tmp/so2.log 2>&1 || true
grep -E "differ|AssertionError" /tmp/so2.log | head -14
python tools/mfma_chain_replay.py make /tmp/conv_dump.npz /tmp/chain
for f in /tmp/chain/chain_*.in.bin; do build/mfma_probe bf16_32 $f ${f%.in.bin}.out.bin; done
python tools/mfma_chain_replay.py check /tmp/chain | tee gpurun_out/r6k/chain_replay.txt
