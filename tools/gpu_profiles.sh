#!/bin/bash
# Round-4 profile collection on the GPU box (run from the repository root through gpurun).  Writes raw rocprofv3 output under
# gpurun_out/prof/ and the judged summaries under gpurun_out/prof/profiles_out/ (copied into profiles/ by the caller).
# Needs build/overlap_bench (the stand-alone GEMM driver of the counter passes), built in the container with the library's flags:
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I pdb2reaction_amd/csrc pdb2reaction_amd/csrc/overlap_bench.hip -o build/overlap_bench
# Counter passes are their own runs (--pmc only, no trace domains).
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; P=$O/profiles_out
mkdir -p $O $P
cd /tmp && export TMPDIR=/tmp && cd $R
BENCH="python3 bench.py --no-cpu-baseline --no-fp32-mode --no-fast-mode --driver string"
echo "== 1. kernel trace + stats: default mode (bf16x3), fast mode (split), fp32 mode" &&
for M in bf16x3 split fp32; do
  UMX_PRECISION=$M rocprofv3 --kernel-trace --stats -d $O/kt_$M -o kt -f csv -- $BENCH --steps 2 --warmup 1 > $O/kt_$M.log 2>&1 &&
  ( echo "# UMX_PRECISION=$M rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-fast-mode  (MI355X, round 4, 3 iterations incl. warm-up)"; cat $O/kt_$M/kt_kernel_stats.csv ) > $P/r04_bench_c3_kernel_stats_$M.csv || exit 1
done
echo "== 2. PMC passes: FETCH_SIZE, WRITE_SIZE (separate, counters only), default mode and fast mode" &&
for M in bf16x3 split; do
  UMX_PRECISION=$M rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch_$M -o p -f csv -- $BENCH --steps 1 --warmup 1 > $O/pmc_fetch_$M.log 2>&1 &&
  UMX_PRECISION=$M rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write_$M -o p -f csv -- $BENCH --steps 1 --warmup 1 > $O/pmc_write_$M.log 2>&1 &&
  python3 tools/pmc_summary.py $O/pmc_fetch_$M $O/pmc_write_$M 2 $P/r04_pmc_hbm_traffic_$M.json $M | tail -n 3 || exit 1
done
echo "== 3. GEMM PMC counters at the real c3 shapes (bf16x3)" &&
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS -d $O/g1 -o g -f csv -- $R/build/overlap_bench pmc3 1 > $O/g1.log 2>&1 &&
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $O/g2 -o g -f csv -- $R/build/overlap_bench pmc3 1 > $O/g2.log 2>&1 &&
python3 tools/gemm_pmc_summary.py $O/g1 $O/g2 $P/r04_gemm_pmc_counters_bf16x3.json bf16x3 | tee $O/gemm_pmc_summary.log &&
echo "== 4. bench line (AFTER the PMC summaries, so that the traffic figure is the one of this very build)" &&
cp $P/r04_pmc_hbm_traffic_bf16x3.json $P/r04_pmc_hbm_traffic_split.json profiles/ &&
timeout -k 10 900 python3 bench.py --steps 5 --warmup 2 --driver gsm --gsm-cycles 10 > $O/bench.log 2>&1 && grep '^{' $O/bench.log | tail -n 1 > $P/r04_bench_c3_n1.json &&
echo "== 5. the other BASELINE configs: wall time per batched E+F, kernel stats of c1 / c2 / c5" &&
for c in c1 c2 c3-shard c4-string c5; do python3 tools/gpu_eval_config.py $c 2>/dev/null; done | tee $P/r04_config_sweep_bf16x3.txt &&
rocprofv3 --kernel-trace --stats -d $O/c2 -o kt -f csv -- python3 tools/gpu_eval_config.py c2 5 > $O/c2.log 2>&1 &&
( echo "# rocprofv3 --kernel-trace --stats -- python3 tools/gpu_eval_config.py c2 5  (MI355X, round 4: c2 = 500 atoms x 12 images, 7 batched E+F evaluations incl. 2 warm-up, default mode bf16x3)"; cat $O/c2/kt_kernel_stats.csv ) > $P/r04_c2_kernel_stats_bf16x3.csv &&
rocprofv3 --kernel-trace --stats -d $O/c1 -o kt -f csv -- python3 tools/gpu_eval_config.py c1 20 > $O/c1.log 2>&1 &&
( echo "# rocprofv3 --kernel-trace --stats -- python3 tools/gpu_eval_config.py c1 20  (MI355X, round 4: c1 = 50 atoms x 8 images, 22 batched E+F evaluations incl. 2 warm-up, default mode bf16x3)"; cat $O/c1/kt_kernel_stats.csv ) > $P/r04_c1_kernel_stats_bf16x3.csv &&
rocprofv3 --kernel-trace --stats -d $O/c5 -o kt -f csv -- python3 tools/gpu_c5_check.py > $O/c5.log 2>&1 &&
( echo "# rocprofv3 --kernel-trace --stats -- python3 tools/gpu_c5_check.py  (MI355X, round 4: c5 = 20 000 atoms x 8 images, one batched E+F, default mode bf16x3)"; cat $O/c5/kt_kernel_stats.csv ) > $P/r04_c5_kernel_stats_bf16x3.csv &&
tail -n 3 $O/c5.log &&
echo "== 6. N > 1 rehearsal of bench.py on this one GPU (gloo group, host-staged all-gather; the RCCL run needs the 8-GPU node)" &&
for G in 2 4; do
  UMX_BENCH_BACKEND=gloo UMX_MAX_CHUNK_IMAGES=2 timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $G --master-addr 127.0.0.1 --master-port $((29510 + G)) bench.py --gpus $G --steps 2 --warmup 1 > $O/gloo$G.log 2>&1 || { tail -n 5 $O/gloo$G.log; exit 1; }
  grep '^{' $O/gloo$G.log | tail -n 1 > $P/r04_bench_c3_gloo_rehearsal_n$G.json
done &&
echo "== done" && ls -la $P
