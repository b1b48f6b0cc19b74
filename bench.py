#!/usr/bin/env python3
"""Headline benchmark: string (GSM-style) iterations/s on a synthetic ~2000-atom x 16-image path.

One "step" = one string iteration = batched UMA E+F of every image of the string on its owner GPU
+ one all-gather of [E | F] (RCCL, only when --gpus > 1) + the replicated string update.
The 16 images of the ONE path are sharded contiguously over the ranks (strong scaling: total work
is fixed by BASELINE.json's workload, N=1 evaluates all 16 images on one GPU).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 3 --warmup 1

Rank 0 prints ONE JSON line (contract in the task statement) carrying `roofline` (fp32-MFMA GEMM
family, timed live with HIP events on the launch stream) and `cpu_baseline` (the repo's own CPU
restatement -- the reference's fairchem path cannot run here -- on a bounded sample, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from pdb2reaction_amd import synth, weights as W  # noqa: E402
from pdb2reaction_amd._calculator_base import ANG2BOHR, BOHR2ANG  # noqa: E402
from pdb2reaction_amd.engine import Engine  # noqa: E402
from pdb2reaction_amd.parallel import ShardedImageEvaluator  # noqa: E402
from pdb2reaction_amd.string import string_step  # noqa: E402
from pdb2reaction_amd.uma_pysis import EV2AU, F_EVAA_2_AU  # noqa: E402

FLOP_PER_EDGE = 30.98e6          # algorithmic E+F work per directed edge
PEAK_FP32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense f32 matrix peak
PEAK_BF16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA peak (not the 2:1-sparse headline)


def pmc_traffic_per_launch(split: bool):
    """HBM bytes per launch of the dominant GEMM family from the committed rocprofv3 PMC passes (profiles/), or None.
    bench.py cannot collect PMC counters itself; the figure is for exactly this workload and build."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_hbm_traffic.json")
    try:
        d = json.load(open(path))
        return float(d["dominant_family"]["hbm_bytes_per_launch_avg"]) if split else None
    except Exception:
        return None


def pmc_non_gemm_bytes_per_step(split: bool):
    """HBM bytes per iteration moved by everything EXCEPT the two GEMM families (same committed PMC passes), or None."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_hbm_traffic.json")
    try:
        d = json.load(open(path))
        return (float(d["hbm_bytes_per_iteration"]) - float(d["dominant_family"]["hbm_bytes_per_iteration"])
                - float(d.get("fp32_gemm_family_hbm_bytes_per_iteration", 0.0))) if split else None
    except Exception:
        return None


PEAK_HBM_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E spec peak (about 6.3 TB/s is achievable by a float4 copy)


def usable_cores() -> int:
    """Host cores this process may actually use (affinity and cgroup quota, not the machine total)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("UMX_CPU_BASELINE_THREADS", "16"))))


def cpu_baseline(n_atoms_sample: int, edges_per_iter: float):
    """Time the CPU oracle (float32, all host threads) on ONE image of `n_atoms_sample` atoms and scale
    by directed edges to the benchmark's string iteration."""
    from oracle.escn_md_oracle import Oracle, radius_graph

    cores = usable_cores()
    torch.set_num_threads(cores)
    w = W.make_synthetic_weights(0)
    orc = Oracle(w, dtype=torch.float32)
    z, pos = synth.make_cluster(n_atoms_sample)
    src, _ = radius_graph(torch.as_tensor(pos), W.CUTOFF)
    t0 = time.perf_counter()
    orc.energy_forces(z, pos.astype(np.float32))
    dt = time.perf_counter() - t0
    ne = int(len(src))
    it_s = 1.0 / (dt * edges_per_iter / ne)
    return {
        "value": it_s, "unit": "iterations/s", "cores": cores, "kind": "port",
        "sample": f"own CPU restatement (oracle/, torch float32, {cores} threads), 1 image x {n_atoms_sample} atoms "
                  f"({ne} directed edges) E+F in {dt:.2f} s, scaled by edges to the {int(edges_per_iter)}-edge iteration",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--atoms", type=int, default=2000)
    ap.add_argument("--images", type=int, default=16)
    ap.add_argument("--cpu-sample-atoms", type=int, default=700)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched through torch.distributed.run with N processes")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X GPU (no CPU fallback exists for the engine)")
    backend = os.environ.get("UMX_BENCH_BACKEND", "nccl")        # "gloo" = rehearsal: ranks may share one GPU, host-staged gather
    if backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    n, k = args.atoms, args.images
    z, imgs, frozen = synth.make_images(n, k)
    eng = Engine(local_rank)
    eng.load_weights(W.make_synthetic_weights(0))
    eng.set_system(z, charge=0, spin=1, task="omol")
    frozen_t = torch.as_tensor(frozen, dtype=torch.long, device=dev)

    # string state: coordinates in Bohr, float64, resident on the device
    x = torch.as_tensor(imgs * ANG2BOHR, dtype=torch.float64, device=dev)       # (K,N,3)

    kl_max = -(-k // world)
    pos32 = torch.empty(kl_max, n, 3, dtype=torch.float32, device=dev)
    e_loc = torch.empty(kl_max, dtype=torch.float64, device=dev)
    f_loc = torch.empty(kl_max, n, 3, dtype=torch.float32, device=dev)

    def evaluate_local(c_bohr):
        kl = c_bohr.shape[0]
        pos32[:kl].copy_(c_bohr * BOHR2ANG)                                        # AtomicData.pos is float32 Angstrom
        eng.energy_forces_dev(kl, pos32.data_ptr(), e_loc.data_ptr(), f_loc.data_ptr(),
                              stream=torch.cuda.current_stream().cuda_stream)
        f = f_loc[:kl].to(torch.float64) * F_EVAA_2_AU
        f[:, frozen_t, :] = 0.0                                                    # uma_pysis.py:561-567
        return e_loc[:kl] * EV2AU, f

    ev = ShardedImageEvaluator(evaluate_local, k, n, dev)

    def step(xc):
        e, f = ev(xc)
        xn = string_step(xc.reshape(k, -1), f.reshape(k, -1), max_step=0.1, alpha=0.5, fix_ends=False)
        return xn.reshape(k, n, 3), e

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        x, e = step(x)
    fence()
    eng.profile_enable(True)
    eng.profile_read(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x, e = step(x)
    fence()
    dt = time.perf_counter() - t0
    prof = eng.profile_read(True)
    eng.profile_enable(False)
    ne_local, maxdeg = eng.graph_stats()            # edges of this rank's images in the last step
    tt = torch.tensor([dt, float(ne_local)], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        dt = float(tmax[0])
    edges_iter = float(tt[1])
    ms = dt / args.steps * 1e3
    it_s = args.steps / dt

    if rank == 0:
        pl, f32 = prof["split_bf16"], prof["fp32"]
        split = pl["launches"] > 0
        dom = pl if split else f32
        # dominant kernel family: in the default (split) mode the plane-interleaved bf16 LDS-DMA GEMM.  `achieved` counts the
        # FLOPs the matrix cores executed (6 bf16 MFMA products per fp32-equivalent product in the forward pass, 3 in the
        # reverse pass) against the dense bf16 peak; `achieved_algorithmic` is the fp32-equivalent rate (2*M*N*K per product).
        ach = dom["mfma_flops"] / max(dom["ms"], 1e-9) / 1e9
        peak = PEAK_BF16_MFMA_TFLOPS if split else PEAK_FP32_MFMA_TFLOPS
        out = {
            "metric": "path_opt_string_iterations_per_s", "value": it_s, "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "bf16x3-split (fp32-equivalent 24-bit products fwd, 16-bit reverse pass), fp32 accumulate" if split else "f32",
            "data": "synthetic",
            "image_atom_steps_per_s": k * n * it_s,
            "algorithmic_tflops": FLOP_PER_EDGE * edges_iter * it_s / 1e12,
            "config": {"workload": f"c3: {n}-atom synthetic active-site cluster x {k} images, GSM-style string iteration "
                                   f"(batched UMA-S E+F of all images + string update), UMA-S shapes, synthetic weights",
                       "atoms": n, "images": k, "directed_edges_per_iteration": int(edges_iter), "max_degree": maxdeg,
                       "parallelism": f"images sharded {k}/{world} per GPU, 1 all-gather/iteration" if world > 1 else "single GPU, all images batched"},
            "roofline": {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                         "traffic": pmc_traffic_per_launch(split) if (n == 2000 and k == 16 and world == 1) else None,
                         "kernel": ("umx_gemm_q_kernel<*> / umx_gemm_pl16_kernel<*> / umx_gemm_pl_kernel<*> (split-bf16 LDS-DMA GEMM family: SO(2)/radial linears + transposes, rank 0)" if split
                                    else "umx_gemm_kernel<*> (fp32-MFMA GEMM, rank 0)"),
                         "launches": dom["launches"], "avg_launch_ms": dom["ms"] / max(dom["launches"], 1),
                         "flops_per_launch": dom["mfma_flops"] / max(dom["launches"], 1),
                         "achieved_algorithmic": dom["alg_flops"] / max(dom["ms"], 1e-9) / 1e9,
                         "algorithmic_flops_per_launch": dom["alg_flops"] / max(dom["launches"], 1),
                         "share_of_step": dom["ms"] / (ms * args.steps),
                         "other_gemm_family": {"kernel": "umx_gemm_kernel<*> (fp32 MFMA)" if split else None,
                                               "ms_per_step": f32["ms"] / args.steps if split else 0.0,
                                               "achieved": f32["alg_flops"] / max(f32["ms"], 1e-9) / 1e9 if split else 0.0,
                                               "peak": PEAK_FP32_MFMA_TFLOPS}},
        }
        # second regime: the HBM-bound gather / rotate / gate / segmented-reduce kernels = everything outside the
        # two GEMM families; bytes from the committed PMC passes of this exact workload, time measured live
        nb = pmc_non_gemm_bytes_per_step(split) if (n == 2000 and k == 16 and world == 1) else None
        if nb is not None:
            rest_ms = ms - (dom["ms"] + f32["ms"]) / args.steps
            out["roofline"]["hbm_regime"] = {"bound": "hbm", "kernels": "all non-GEMM kernels (k_gather_rotate_mod_pl, k_modrot_bwd_pl, k_gate_edge_*, k_rotate_back_*, ...)",
                                             "ms_per_step": rest_ms, "traffic_per_step": nb, "achieved": nb / max(rest_ms, 1e-9) / 1e6,
                                             "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": nb / max(rest_ms, 1e-9) / 1e6 / PEAK_HBM_GBPS}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.cpu_sample_atoms, edges_iter)
            except Exception as exc:  # the baseline is informative; never lose the GPU numbers to it
                out["cpu_baseline"] = {"value": None, "unit": "iterations/s", "cores": usable_cores(), "kind": "port",
                                       "sample": f"failed: {type(exc).__name__}: {exc}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
