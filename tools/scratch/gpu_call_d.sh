#!/bin/bash
# GPU call D: PMC counters of the split-bf16 GEMM kernels at the real c3 shapes
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/d
B=$R/build/overlap_bench
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/d/trace -o gemm -f csv -- $B pmc 3 > $R/gpurun_out/d/trace.log 2>&1 &&
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 -d $R/gpurun_out/d/pmc1 -o gemm -f csv -- $B pmc 1 > $R/gpurun_out/d/pmc1.log 2>&1 &&
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $R/gpurun_out/d/pmc2 -o gemm -f csv -- $B pmc 1 > $R/gpurun_out/d/pmc2.log 2>&1
echo "rc=$?"; tail -3 $R/gpurun_out/d/pmc1.log $R/gpurun_out/d/pmc2.log; ls -R $R/gpurun_out/d | head -30
