#!/bin/bash
# Profile collection of the round (RND, default r06) on the GPU box (run from the repository root through gpurun, AFTER the last code change).  Writes raw rocprofv3
# output under gpurun_out/prof/ and the judged summaries under gpurun_out/prof/profiles_out/ (copied into profiles/ by the caller).
# Needs build/overlap_bench (the stand-alone GEMM driver of the counter passes), built in the container with the library's flags:
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I pdb2reaction_amd/csrc pdb2reaction_amd/csrc/overlap_bench.hip -o build/overlap_bench
# Counter passes are their own runs (--pmc only, no trace domains).  STEP selects a part (default: all): kt | pmc | gemm | bench | cfg
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; P=$O/profiles_out; STEP=${1:-all}; RND=${RND:-r06}
mkdir -p $O $P
cd /tmp && export TMPDIR=/tmp && cd $R
BENCH="python3 bench.py --no-cpu-baseline --no-fp32-mode --no-fast-mode --no-shard --no-serial --driver string"
if [ $STEP = all ] || [ $STEP = kt ]; then
echo "== 1. kernel trace + stats: default mode (bf16x3), fast mode (split), fp32 mode on the default schedule (one lane), and bf16x3 on two lanes" &&
for M in bf16x3 split fp32; do
  UMX_PRECISION=$M rocprofv3 --kernel-trace --stats -d $O/kt_$M -o kt -f csv -- $BENCH --steps 2 --warmup 1 > $O/kt_$M.log 2>&1 &&
  ( echo "# UMX_PRECISION=$M rocprofv3 --kernel-trace --stats -- $BENCH --steps 2 --warmup 1  (MI355X, round 6, 3 iterations incl. warm-up; default schedule = one lane)"; cat $O/kt_$M/kt_kernel_stats.csv ) > $P/${RND}_bench_c3_kernel_stats_$M.csv || exit 1
done
UMX_STREAMS=2 rocprofv3 --kernel-trace --stats -d $O/kt_lanes2 -o kt -f csv -- $BENCH --steps 2 --warmup 1 > $O/kt_lanes2.log 2>&1 &&
( echo "# UMX_STREAMS=2 rocprofv3 --kernel-trace --stats -- $BENCH --steps 2 --warmup 1  (MI355X, round 6, bf16x3, TWO lanes: the large-GEMM segments of one chunk beside the HBM-bound segments of the other, 3 iterations incl. warm-up)"; cat $O/kt_lanes2/kt_kernel_stats.csv ) > $P/${RND}_bench_c3_kernel_stats_bf16x3_two_lanes.csv || exit 1
fi
if [ $STEP = all ] || [ $STEP = pmc ]; then
echo "== 2. PMC passes: FETCH_SIZE, WRITE_SIZE (separate, counters only), default mode and fast mode" &&
for M in bf16x3 split; do
  UMX_PRECISION=$M rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch_$M -o p -f csv -- $BENCH --steps 1 --warmup 1 > $O/pmc_fetch_$M.log 2>&1 &&
  UMX_PRECISION=$M rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write_$M -o p -f csv -- $BENCH --steps 1 --warmup 1 > $O/pmc_write_$M.log 2>&1 &&
  python3 tools/pmc_summary.py $O/pmc_fetch_$M $O/pmc_write_$M 2 $P/${RND}_pmc_hbm_traffic_$M.json $M | tail -n 3 || exit 1
done
fi
if [ $STEP = all ] || [ $STEP = gemm ]; then
echo "== 3. GEMM PMC counters at the real c3 shapes (bf16x3)" &&
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS -d $O/g1 -o g -f csv -- $R/build/overlap_bench pmc3 1 > $O/g1.log 2>&1 &&
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $O/g2 -o g -f csv -- $R/build/overlap_bench pmc3 1 > $O/g2.log 2>&1 &&
python3 tools/gemm_pmc_summary.py $O/g1 $O/g2 $P/${RND}_gemm_pmc_counters_bf16x3.json bf16x3 | tee $O/gemm_pmc_summary.log || exit 1
fi
if [ $STEP = all ] || [ $STEP = bench ]; then
echo "== 4. bench line (AFTER the PMC summaries, so that the traffic figure is the one of this very build)" &&
{ [ ! -f $P/${RND}_pmc_hbm_traffic_bf16x3.json ] || cp $P/${RND}_pmc_hbm_traffic_bf16x3.json $P/${RND}_pmc_hbm_traffic_split.json profiles/; } &&   # (a separate gpurun call: the caller has copied them into profiles/ already)
( TIMEFORMAT="%R s wall"; { time timeout -k 10 900 python3 bench.py --steps 5 --warmup 2 --driver gsm --gsm-cycles 10 > $O/bench.log 2> $O/bench.err; } 2> $P/${RND}_bench_c3_n1.walltime ) &&
grep '^{' $O/bench.log | tail -n 1 > $P/${RND}_bench_c3_n1.json || exit 1
fi
if [ $STEP = all ] || [ $STEP = cfg ]; then
echo "== 5. the other BASELINE configs: bench lines (bench.py --config), kernel stats of c1 / c2 / c5" &&
for c in c1 c2 c4; do timeout -k 10 700 python3 bench.py --config $c --steps 5 --warmup 2 --no-fp32-mode --no-fast-mode --driver string > $O/bench_$c.log 2>&1 && grep '^{' $O/bench_$c.log | tail -n 1 > $P/${RND}_bench_$c.json || exit 1; done
timeout -k 10 500 python3 bench.py --config c5 --steps 2 --warmup 1 --no-fp32-mode --no-fast-mode --driver string > $O/bench_c5.log 2>&1 && grep '^{' $O/bench_c5.log | tail -n 1 > $P/${RND}_bench_c5.json || exit 1
rocprofv3 --kernel-trace --stats -d $O/c2 -o kt -f csv -- python3 tools/gpu_eval_config.py c2 5 > $O/c2.log 2>&1 &&
( echo "# rocprofv3 --kernel-trace --stats -- python3 tools/gpu_eval_config.py c2 5  (MI355X, round 6: c2 = 500 atoms x 12 images, 7 batched E+F evaluations incl. 2 warm-up, default mode bf16x3)"; cat $O/c2/kt_kernel_stats.csv ) > $P/${RND}_c2_kernel_stats_bf16x3.csv &&
rocprofv3 --kernel-trace --stats -d $O/c1 -o kt -f csv -- python3 tools/gpu_eval_config.py c1 20 > $O/c1.log 2>&1 &&
( echo "# rocprofv3 --kernel-trace --stats -- python3 tools/gpu_eval_config.py c1 20  (MI355X, round 6: c1 = 50 atoms x 8 images, 22 batched E+F evaluations incl. 2 warm-up, default mode bf16x3)"; cat $O/c1/kt_kernel_stats.csv ) > $P/${RND}_c1_kernel_stats_bf16x3.csv &&
rocprofv3 --kernel-trace --stats -d $O/c5 -o kt -f csv -- python3 tools/gpu_c5_check.py > $O/c5.log 2>&1 &&
( echo "# rocprofv3 --kernel-trace --stats -- python3 tools/gpu_c5_check.py  (MI355X, round 6: c5 = 20 000 atoms x 8 images, one batched E+F, default mode bf16x3)"; cat $O/c5/kt_kernel_stats.csv ) > $P/${RND}_c5_kernel_stats_bf16x3.csv &&
tail -n 3 $O/c5.log || exit 1
fi
if [ $STEP = all ] || [ $STEP = gloo ]; then
echo "== 6. N > 1 rehearsal of bench.py on this one GPU (gloo group, host-staged all-gather; the RCCL run needs the 8-GPU node; at most SIX processes may have the card open and the launcher is one of them (a 6-rank attempt was killed by the process guard: 7 processes), so 8 ranks cannot be rehearsed here -- 5 ranks = ragged 4,3,3,3,3 shards)" &&
for G in 2 4 5; do
  # (per-rank workspace capped: 5 ranks x 1-image chunks of 20 GB + the graph stay far below the card's 288 GB)
  UMX_BENCH_BACKEND=gloo UMX_MAX_CHUNK_IMAGES=$([ $G = 5 ] && echo 1 || echo 2) UMX_WS_GB=$([ $G = 5 ] && echo 30 || echo 60) timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $G --master-addr 127.0.0.1 --master-port $((29510 + G)) bench.py --gpus $G --steps 2 --warmup 1 > $O/gloo$G.log 2>&1 || { tail -n 5 $O/gloo$G.log; exit 1; }
  grep '^{' $O/gloo$G.log | tail -n 1 > $P/${RND}_bench_c3_gloo_rehearsal_n$G.json
done
fi
echo "== done" && ls -la $P
