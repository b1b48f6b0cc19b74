#!/bin/bash
# Round-3 profile collection on the GPU box (run from the repository root through gpurun).  Writes raw rocprofv3 output under
# gpurun_out/prof/ and the judged summaries under profiles/ (copied back by the caller from gpurun_out/prof/profiles_out/).
# Needs build/overlap_bench (the stand-alone GEMM driver of the counter passes), built here in the container with the library's flags:
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I pdb2reaction_amd/csrc pdb2reaction_amd/csrc/overlap_bench.hip -o build/overlap_bench
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; P=$O/profiles_out
mkdir -p $O $P
cd /tmp && export TMPDIR=/tmp && cd $R
BENCH="python3 bench.py --no-cpu-baseline --no-fp32-mode"
echo "== 2. kernel trace + stats (split)" &&
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -f csv -- $BENCH --steps 2 --warmup 1 > $O/kt.log 2>&1 &&
( echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-mode  (MI355X, round 3, default split build (fp16 forward planes), 3 iterations incl. warm-up)"; cat $O/kt/kt_kernel_stats.csv ) > $P/r03_bench_c3_kernel_stats_split.csv &&
echo "== 3. kernel trace + stats (fp32 mode)" &&
UMX_PRECISION=fp32 rocprofv3 --kernel-trace --stats -d $O/kt32 -o kt -f csv -- $BENCH --steps 2 --warmup 1 > $O/kt32.log 2>&1 &&
( echo "# UMX_PRECISION=fp32 rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-mode  (MI355X, round 3, all-fp32-MFMA mode, 3 iterations incl. warm-up)"; cat $O/kt32/kt_kernel_stats.csv ) > $P/r03_bench_c3_kernel_stats_fp32.csv &&
echo "== 4. PMC passes: FETCH_SIZE, WRITE_SIZE (separate, counters only)" &&
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o p -f csv -- $BENCH --steps 1 --warmup 1 > $O/pmc_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o p -f csv -- $BENCH --steps 1 --warmup 1 > $O/pmc_write.log 2>&1 &&
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write 2 $P/r03_pmc_hbm_traffic.json | tee $O/pmc_summary.log &&
echo "== 5. GEMM PMC counters at the real c3 shapes" &&
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS -d $O/g1 -o g -f csv -- $R/build/overlap_bench pmc 1 > $O/g1.log 2>&1 &&
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $O/g2 -o g -f csv -- $R/build/overlap_bench pmc 1 > $O/g2.log 2>&1 &&
python3 tools/gemm_pmc_summary.py $O/g1 $O/g2 $P/r03_gemm_pmc_counters.json | tee $O/gemm_pmc_summary.log &&
echo "== 1. bench line (default build; AFTER the PMC summary so that the traffic figure is the one of this very build)" &&
cp $P/r03_pmc_hbm_traffic.json profiles/r03_pmc_hbm_traffic.json &&
timeout -k 10 600 python3 bench.py --steps 3 --warmup 1 > $O/bench.log 2>&1 && grep '^{' $O/bench.log | tail -1 > $P/r03_bench_c3_n1_split.json &&
echo "== 6. c2 (500 atoms x 12 images: the 1-GPU BASELINE config) and c1 kernel stats" &&
rocprofv3 --kernel-trace --stats -d $O/c2 -o kt -f csv -- python3 tools/gpu_eval_config.py c2 5 > $O/c2.log 2>&1 &&
( echo "# rocprofv3 --kernel-trace --stats -- python3 tools/gpu_eval_config.py c2 5  (MI355X, round 3: c2 = 500 atoms x 12 images, 7 batched E+F evaluations incl. 2 warm-up, default mode)"; cat $O/c2/kt_kernel_stats.csv ) > $P/r03_c2_kernel_stats_split.csv &&
rocprofv3 --kernel-trace --stats -d $O/c1 -o kt -f csv -- python3 tools/gpu_eval_config.py c1 20 > $O/c1.log 2>&1 &&
( echo "# rocprofv3 --kernel-trace --stats -- python3 tools/gpu_eval_config.py c1 20  (MI355X, round 3: c1 = 50 atoms x 8 images, 22 batched E+F evaluations incl. 2 warm-up, default mode)"; cat $O/c1/kt_kernel_stats.csv ) > $P/r03_c1_kernel_stats_split.csv &&
tail -1 $O/c2.log && tail -1 $O/c1.log &&
echo "== 7. c5 (20 000 atoms x 8 images) kernel stats" &&
rocprofv3 --kernel-trace --stats -d $O/c5 -o kt -f csv -- python3 tools/gpu_c5_check.py > $O/c5.log 2>&1 &&
( echo "# rocprofv3 --kernel-trace --stats -- python3 tools/gpu_c5_check.py  (MI355X, round 3: c5 = 20 000 atoms x 8 images, one batched E+F, split mode)"; cat $O/c5/kt_kernel_stats.csv ) > $P/r03_c5_kernel_stats_split.csv &&
tail -3 $O/c5.log &&
echo "== 8. N > 1 rehearsal of bench.py on this one GPU (gloo group, host-staged all-gather; the RCCL run needs the 8-GPU node)" &&
for G in 2 4; do
  UMX_BENCH_BACKEND=gloo UMX_MAX_CHUNK_IMAGES=2 timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $G --master-addr 127.0.0.1 --master-port $((29510 + G)) bench.py --gpus $G --steps 2 --warmup 1 > $O/gloo$G.log 2>&1 || { tail -5 $O/gloo$G.log; exit 1; }
  grep '^{' $O/gloo$G.log | tail -1 > $P/r03_bench_c3_gloo_rehearsal_n$G.json
done &&
echo "== done" && ls -la $P
