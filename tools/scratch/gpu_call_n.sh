#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/n
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== 2-rank gloo rehearsal of bench.py --gpus 2 on one GPU (c3 size)" &&
UMX_BENCH_BACKEND=gloo UMX_WS_GB=100 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 > $R/gpurun_out/n/bench_gloo2.log 2>&1; echo "rc=$?"; grep '^{' $R/gpurun_out/n/bench_gloo2.log | cut -c1-600; tail -3 $R/gpurun_out/n/bench_gloo2.log | cut -c1-300
echo "== 4-rank small" &&
UMX_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 4 --steps 2 --warmup 1 --atoms 300 --images 6 > $R/gpurun_out/n/bench_gloo4.log 2>&1; echo "rc=$?"; grep '^{' $R/gpurun_out/n/bench_gloo4.log | cut -c1-400
