#!/bin/bash
# Timing ablations of the complex quad-row bf16x3 GEMM (round 6): build/overlap_bench_abl<mask> = overlap_bench.hip built with -DUMX_GEMM_ABL=<mask>
# (umx_gemm_q.h: 1 no 2^-16-order products, 2 A read + split at k-tile 0 only, 4 no ring refill, 8 no C stores, 16 B fragments at k-tile 0 only, 32 no barrier).
# Run from the repository root through gpurun; prints the "sum" line of the g3 mode (four-product kernel -> three-product kernel) per mask.
set -o pipefail
mkdir -p gpurun_out
for m in "$@"; do
  echo "== UMX_GEMM_ABL=$m"
  timeout -k 10 120 build/overlap_bench_abl$m g3 3 || exit 1
done 2>&1 | tee gpurun_out/gemm_ablate.txt
