#!/bin/bash
# round 6, evidence on the final build: energy error on every 20 000-atom golden that exists (three precision modes), fuzz parity, determinism
set -e
mkdir -p gpurun_out/r6f
CASES="c3 c5 g1 w1 perm"; [ -f tests/golden/c5_n20000_w2.npz ] && CASES="$CASES w2"; [ -f tests/golden/c5_n20000_w3.npz ] && CASES="$CASES w3"
BIAS_ENVS='[{"UMX_PRECISION":"bf16x3"},{"UMX_PRECISION":"bf16x3","UMX_ALIGN_PLANES":"0"},{"UMX_PRECISION":"bf16x3","UMX_ALIGN_PLANES":"1"},{"UMX_PRECISION":"fp32"},{"UMX_PRECISION":"split"},{"UMX_PRECISION":"bf16x3","UMX_LOW_SEP":"1"}]' \
  python tools/gpu_energy_bias.py $CASES > gpurun_out/r6f/energy_bias.txt 2>&1
grep -v amdgpu.ids gpurun_out/r6f/energy_bias.txt
python tools/gpu_fuzz_parity.py 31 100 > gpurun_out/r6f/fuzz_31_100.txt 2>&1 || { tail -5 gpurun_out/r6f/fuzz_31_100.txt; exit 1; }
tail -3 gpurun_out/r6f/fuzz_31_100.txt
python tools/gpu_repeat_bitwise.py 2000 4 20 > gpurun_out/r6f/repeat_bitwise.txt 2>&1; tail -4 gpurun_out/r6f/repeat_bitwise.txt
