#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/e
timeout -k 10 200 $R/build/gemm_bench 569632 512 512 2>&1 | grep -E "Q3|check Q3" | tee $R/gpurun_out/e/gemm_bench.log &&
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "engine_matches or precision_modes or stage_by_stage or c3_energy" 2>&1 | tail -3 &&
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/e/pmc -o gemm -f csv -- $R/build/overlap_bench pmc 1 > $R/gpurun_out/e/pmc.log 2>&1) &&
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-mode 2>&1 | tail -1 | cut -c1-400
