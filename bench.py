#!/usr/bin/env python3
"""Headline benchmark: string (GSM-style) iterations/s on a synthetic ~2000-atom x 16-image path.

One "step" = one string iteration = batched UMA E+F of every image of the string on its owner GPU
+ one all-gather of [E | F] (RCCL, only when --gpus > 1) + the replicated string update.
The 16 images of the ONE path are sharded contiguously over the ranks (strong scaling: total work
is fixed by BASELINE.json's workload, N=1 evaluates all 16 images on one GPU).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 3 --warmup 1

Rank 0 prints ONE JSON line (contract in the task statement) carrying

* `roofline`: the dominant kernel family = the split-precision LDS-DMA GEMMs (SO(2) / radial linears and their transposes),
  timed live with HIP events on the launch stream.  `achieved` / `frac` are ALGORITHMIC: 2*M*N*K per product (SURVEY.md
  8d / Appendix D) over the measured kernel time, against the dense 16-bit MFMA peak (fp16 = bf16 rate).  The redundant plane
  products of the split (forward: x4 on fp16 planes in the default mode, x6 on bf16 planes in split-bf16; reverse: x3 on bf16
  planes) are emulation overhead, reported separately as `mfma_pipe_util` (executed FLOPs / peak).
  `traffic` (HBM bytes per launch from rocprofv3 PMC passes) is only emitted when the committed summary under profiles/
  was measured on EXACTLY this build (source digest compiled into libumx.so), else null + `traffic_source` says why;
* `fp32_mode`: the same workload on the all-fp32-MFMA build of the engine (`UMX_PRECISION=fp32`, the strict
  same-arithmetic-as-the-reference number), a few steps timed the same way (N=1 only);
* `cpu_baseline`: the repo's own CPU restatement (the reference's fairchem path cannot run here) on a bounded sample.
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
PEAK_BF16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA peak (not the 2:1-sparse headline)
PEAK_FP64_MFMA_TFLOPS = 78.6     # AMD MI355X datasheet: f64 matrix = f64 vector peak (the micro-arch guide has no f64 row)


PMC_SUMMARY = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r03_pmc_hbm_traffic.json")


def pmc_summary(build_digest: str):
    """(summary dict | None, source note).  bench.py cannot collect PMC counters itself (rocprofv3 wraps the process), so HBM
    traffic comes from the committed summary of two PMC passes over this very command -- but ONLY when that summary was
    measured on the build that is running now (`csrc_sha256` == the digest compiled into libumx.so).  A kernel edit makes
    the figure vanish from the line instead of going stale."""
    try:
        with open(PMC_SUMMARY) as f:
            d = json.load(f)
    except Exception as exc:
        return None, f"no PMC summary ({type(exc).__name__})"
    rel = os.path.relpath(PMC_SUMMARY, os.path.dirname(os.path.abspath(__file__)))
    if d.get("csrc_sha256") != build_digest:
        return None, f"{rel} is stale: measured on build {str(d.get('csrc_sha256'))[:12]}, running {build_digest[:12]}"
    return d, f"{rel} @ build {build_digest[:12]} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 x2 FETCH correction)"


PEAK_HBM_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E spec peak (about 6.3 TB/s is achievable by a float4 copy)
MEASURED_COPY_GBPS = 6290.0           # float4 copy on this part (MI355X_MICROARCH.md: 79 % of the 8 TB/s specification)


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


def cpu_baseline(edges_per_iter: float, n_atoms: int, n_images: int, budget_s: float = 240.0):
    """The CPU oracle (``oracle/chunked.py``: the float64-validated hand-derived reverse pass, here in float32 on all usable host
    threads) on the BOUNDED sample SURVEY.md 8d prescribes: c1 in full (50 atoms x 8 images), c2 on 3 of its 12 images (500 atoms;
    stated), and the benchmark's own workload (c3: 2000 atoms) with K = 2 images, scaled x K/2 to the string iteration -- the work
    is linear in images.  A leg that would push the total beyond `budget_s` is cut to one image and says so."""
    from oracle.chunked import ChunkedForces

    cores = usable_cores()
    torch.set_num_threads(cores)
    ch = ChunkedForces(W.make_synthetic_weights(0), dtype=torch.float32)
    t_all = time.perf_counter()

    def timed(n, k_total, k_run):
        z, imgs, _ = synth.make_images(n, k_total)
        t0 = time.perf_counter()
        for k in range(k_run):
            ch.energy_forces(z, imgs[k].astype(np.float32))
        return time.perf_counter() - t0

    t_c1 = timed(50, 8, 8)
    t_c2 = timed(500, 12, 3)
    # c3 leg: two images unless the c2 timing says that would not fit (2000-atom images cost ~4.7x a 500-atom one: edges 142 k vs 30 k)
    k3 = 2 if (time.perf_counter() - t_all) + 2 * 4.7 * (t_c2 / 3) < budget_s else 1
    t_c3 = timed(n_atoms, n_images, k3)
    it_s = 1.0 / (t_c3 / k3 * n_images)
    return {
        "value": it_s, "unit": "iterations/s", "cores": cores, "kind": "port",
        "sample": (f"own CPU restatement (oracle/chunked.py, torch float32, {cores} threads), NOT fairchem -- SURVEY.md 8d legs: "
                   f"c1 in full (50 atoms x 8 images) {t_c1:.2f} s = {1.0 / t_c1:.3f} iterations/s; c2 (500 atoms) 3 of 12 images in {t_c2:.2f} s "
                   f"= {1.0 / (t_c2 / 3 * 12):.4f} iterations/s when scaled x4; c3 ({n_atoms} atoms) K = {k3} image(s) in {t_c3:.2f} s, scaled x{n_images // k3} "
                   f"to the {n_images}-image iteration" + ("" if k3 == 2 else " (K = 2 would not fit the time budget: leg cut to one image)")),
        "c1_iterations_per_s": 1.0 / t_c1, "c2_iterations_per_s": 1.0 / (t_c2 / 3 * 12), "c3_images_timed": k3, "seconds": time.perf_counter() - t_all,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--atoms", type=int, default=2000)
    ap.add_argument("--images", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp32-mode", action="store_true")
    ap.add_argument("--fp32-steps", type=int, default=5)
    ap.add_argument("--fp32-warmup", type=int, default=2)
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
    weights = W.make_synthetic_weights(0)
    frozen_t = torch.as_tensor(frozen, dtype=torch.long, device=dev)
    kl_max = -(-k // world)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(precision: str, steps: int, warmup: int):
        """W untimed + K timed string iterations on a fresh engine in `precision` mode; returns (dt, profile, edges, maxdeg, resolved mode)."""
        os.environ["UMX_PRECISION"] = precision              # read by umx_load_weights
        eng = Engine(local_rank)
        eng.load_weights(weights)
        eng.set_system(z, charge=0, spin=1, task="omol")
        eng.reserve_images(kl_max)                          # a long run of fixed-size batches: workspace for the whole shard, allocated once
        x = torch.as_tensor(imgs * ANG2BOHR, dtype=torch.float64, device=dev)       # string state: Bohr, float64, in HBM
        pos32 = torch.empty(kl_max, n, 3, dtype=torch.float32, device=dev)
        e_loc = torch.empty(kl_max, dtype=torch.float64, device=dev)
        f_loc = torch.empty(kl_max, n, 3, dtype=torch.float32, device=dev)

        def evaluate_local(c_bohr):
            kl = c_bohr.shape[0]
            pos32[:kl].copy_(c_bohr * BOHR2ANG)                                    # AtomicData.pos is float32 Angstrom
            # the engine enqueues on torch's current stream (handle 0 = the legacy default stream): producer (copy above) and
            # consumers (conversion below, the all-gather) are ordered with it by the stream alone
            eng.energy_forces_dev(kl, pos32.data_ptr(), e_loc.data_ptr(), f_loc.data_ptr(),
                                  stream=torch.cuda.current_stream().cuda_stream)
            f = f_loc[:kl].to(torch.float64) * F_EVAA_2_AU
            f[:, frozen_t, :] = 0.0                                                # uma_pysis.py:561-567
            return e_loc[:kl] * EV2AU, f

        # engine=: the device-pointer entry cannot refuse a non-finite energy itself; the evaluator checks the gathered energies
        # every iteration (one scalar read) and would widen the engine on all ranks together (parallel.py)
        ev = ShardedImageEvaluator(evaluate_local, k, n, dev, engine=eng)

        def step(xc):
            e, f = ev(xc)
            xn = string_step(xc.reshape(k, -1), f.reshape(k, -1), max_step=0.1, alpha=0.5, fix_ends=False)
            return xn.reshape(k, n, 3), e

        for _ in range(warmup):
            x, e = step(x)
        fence()
        eng.profile_enable(True)
        eng.profile_read(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            x, e = step(x)
        fence()
        dt = time.perf_counter() - t0
        prof = eng.profile_read(True)
        eng.profile_enable(False)
        ne_local, maxdeg = eng.graph_stats()            # edges of this rank's images in the last step
        if not bool(torch.isfinite(e).all()) or eng.widened:
            raise SystemExit("bench.py: non-finite energies / a precision change inside the timed region")
        resolved = eng.precision_mode()
        eng.close()
        return dt, prof, ne_local, maxdeg, resolved

    mode_req = os.environ.get("UMX_PRECISION", "auto")           # "auto": split-f16 up to 4096 atoms per image, split-bf16 above (include/umx.h)
    dt, prof, ne_local, maxdeg, resolved = run(mode_req, args.steps, args.warmup)
    mode = {"split-f16": "split", "split-bf16": "split-bf16", "bf16x3": "bf16x3", "fp32": "fp32"}[resolved]
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
        from pdb2reaction_amd.engine import load_library

        digest = load_library().umx_build_digest().decode()
        pl, f32 = prof["split_bf16"], prof["fp32"]
        split = pl["launches"] > 0
        dom = pl if split else f32
        peak = PEAK_BF16_MFMA_TFLOPS if split else PEAK_FP32_MFMA_TFLOPS
        alg = dom["alg_flops"] / max(dom["ms"], 1e-9) / 1e9          # algorithmic TFLOP/s of the dominant family (2*M*N*K per product)
        executed = dom["mfma_flops"] / max(dom["ms"], 1e-9) / 1e9    # what the matrix cores executed (x6 forward / x3 reverse split products)
        pmc, pmc_note = pmc_summary(digest) if (n == 2000 and k == 16 and world == 1 and split) else (None, "PMC summary exists for c3 / 1 GPU / split mode only")
        out = {
            "metric": "path_opt_string_iterations_per_s", "value": it_s, "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": ("f16-split" if mode in ("split", "split-f16") else "bf16-split") if split else "f32",
            "dtype_detail": ((("forward GEMMs: 2 fp16 activation planes x 3 exact fp16 weight planes, 4 MFMA products (fp32-level); " if mode in ("split", "split-f16")
                               else "forward GEMMs: 3 x 3 bf16 planes, 6 MFMA products (24-bit); ")
                              + "reverse GEMMs: 2 x 2 bf16 planes, 3 products (16-bit); fp32 accumulate; node-level linears float64-accumulated; everything else fp32") if split
                             else "every GEMM on v_mfma_f32_32x32x2_f32"),
            "precision_mode": mode, "precision_requested": mode_req,
            "data": "synthetic",
            "image_atom_steps_per_s": k * n * it_s,
            "algorithmic_tflops": FLOP_PER_EDGE * edges_iter * it_s / 1e12,
            "build_digest": digest,
            "config": {"workload": f"c3: {n}-atom synthetic active-site cluster x {k} images, GSM-style string iteration "
                                   f"(batched UMA-S E+F of all images + string update), UMA-S shapes, synthetic weights",
                       "atoms": n, "images": k, "directed_edges_per_iteration": int(edges_iter), "max_degree": maxdeg,
                       "parallelism": f"images sharded {k}/{world} per GPU, 1 all-gather/iteration" if world > 1 else "single GPU, all images batched"},
            "roofline": {"bound": "mfma", "achieved": alg, "peak": peak, "unit": "TFLOP/s", "frac": alg / peak,
                         "definition": "achieved = algorithmic FLOPs (2*M*N*K per product, SURVEY.md 8d) of the family / its HIP-event time; "
                                       "mfma_pipe_util counts the plane products actually executed (fwd x4 fp16 / x6 bf16, reverse x3)",
                         "mfma_pipe_util": executed / peak, "executed_tflops": executed,
                         "traffic": float(pmc["dominant_family"]["hbm_bytes_per_launch_avg"]) if pmc else None, "traffic_source": pmc_note,
                         "kernel": ("umx_gemm_q_kernel<*> / umx_gemm_pl16_kernel<*> / umx_gemm_pl_kernel<*> (split-precision LDS-DMA GEMM family: SO(2)/radial linears + transposes, rank 0)" if split
                                    else "umx_gemm_kernel<*> (fp32-MFMA GEMM, rank 0)"),
                         "launches": dom["launches"], "avg_launch_ms": dom["ms"] / max(dom["launches"], 1),
                         "algorithmic_flops_per_launch": dom["alg_flops"] / max(dom["launches"], 1),
                         "executed_flops_per_launch": dom["mfma_flops"] / max(dom["launches"], 1),
                         "ms_per_step": dom["ms"] / args.steps, "share_of_step": dom["ms"] / (ms * args.steps),
                         "vs_fp32_mfma_peak": alg / PEAK_FP32_MFMA_TFLOPS,
                         "other_gemm_family": {"kernel": "k_gemm_f64acc<*> (node-level linears, float64-accumulated on v_mfma_f64_16x16x4_f64; UMX_NODE_F64=0: umx_gemm_kernel<*>, fp32 MFMA)" if split else None,
                                               "ms_per_step": f32["ms"] / args.steps if split else 0.0,
                                               "achieved": f32["alg_flops"] / max(f32["ms"], 1e-9) / 1e9 if split else 0.0,
                                               "peak": PEAK_FP64_MFMA_TFLOPS if os.environ.get("UMX_NODE_F64", "1") != "0" else PEAK_FP32_MFMA_TFLOPS}},
        }
        # second regime: the HBM-bound gather / rotate / gate / segmented-reduce kernels.  Everything outside the two GEMM
        # families is timed as the remainder of the step; the fused radial-MLP kernels (VALU / fp32-MFMA bound, not HBM bound) are timed
        # live as their own family (ABI v7) and taken OUT of the HBM figure, so that `achieved` is not a blend of two bounds.
        rad = prof["radial"]
        rad_ms = rad["ms"] / args.steps
        rest_ms = ms - (dom["ms"] + (f32["ms"] if split else 0.0)) / args.steps
        edge_ms = rest_ms - rad_ms
        hb = {"bound": "hbm", "kernels": "HBM-bound edge / node kernels: k_gather_rotate_mod_q3, k_modrot_bwd_pl, k_gate_edge_*, k_rotate_back_*, norms, graph build "
                                         "(everything outside the GEMM families and the fused radial-MLP kernels)",
              "ms_per_step": edge_ms, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "traffic_per_step": None, "achieved": None, "frac": None,
              "traffic_source": pmc_note,
              "radial": {"kernels": "k_radial_head / k_radial_tail (fused radial-MLP layers: libm-accurate VALU transcendentals + fp32 MFMA)", "bound": "valu/mfma-f32",
                         "ms_per_step": rad_ms, "launches": rad["launches"], "achieved": rad["alg_flops"] / max(rad["ms"], 1e-9) / 1e9, "unit": "TFLOP/s",
                         "peak": PEAK_FP32_MFMA_TFLOPS, "frac": rad["alg_flops"] / max(rad["ms"], 1e-9) / 1e9 / PEAK_FP32_MFMA_TFLOPS, "traffic_per_step": None}}
        if pmc:
            rb = float(pmc.get("radial_hbm_bytes_per_iteration", 0.0))
            nb = (float(pmc["hbm_bytes_per_iteration"]) - float(pmc["dominant_family"]["hbm_bytes_per_iteration"])
                  - float(pmc.get("fp32_gemm_family_hbm_bytes_per_iteration", 0.0)) - rb)
            hb.update(traffic_per_step=nb, achieved=nb / max(edge_ms, 1e-9) / 1e6, frac=nb / max(edge_ms, 1e-9) / 1e6 / PEAK_HBM_GBPS,
                      total_traffic_per_step=float(pmc["hbm_bytes_per_iteration"]))
            hb["radial"]["traffic_per_step"] = rb
            # MI355X_MICROARCH.md: 8.0 TB/s is the specification, a float4 copy measures 6.29 TB/s -- the second fraction says how far the
            # edge kernels are from what the memory system delivers
            hb.update(measured_copy_peak=MEASURED_COPY_GBPS, frac_of_measured_copy_peak=hb["achieved"] / MEASURED_COPY_GBPS)
        out["roofline"]["hbm_regime"] = hb
        if world == 1 and split and not args.no_fp32_mode:
            # the strict same-arithmetic-as-the-reference figure: every GEMM on v_mfma_f32_32x32x2_f32, timed by the same clock
            try:
                dt32, prof32, _, _, _ = run("fp32", args.fp32_steps, args.fp32_warmup)
                g32 = prof32["fp32"]
                out["fp32_mode"] = {"value": args.fp32_steps / dt32, "unit": "iterations/s", "ms_per_step": dt32 / args.fp32_steps * 1e3,
                                    "steps": args.fp32_steps, "warmup": args.fp32_warmup, "dtype": "f32 (all GEMMs on v_mfma_f32_32x32x2_f32)",
                                    "gemm_tflops": g32["alg_flops"] / max(g32["ms"], 1e-9) / 1e9, "gemm_peak": PEAK_FP32_MFMA_TFLOPS,
                                    "gemm_frac": g32["alg_flops"] / max(g32["ms"], 1e-9) / 1e9 / PEAK_FP32_MFMA_TFLOPS}
            except Exception as exc:
                out["fp32_mode"] = {"value": None, "error": f"{type(exc).__name__}: {exc}"}
            os.environ["UMX_PRECISION"] = mode_req
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(edges_iter, n, k)
            except Exception as exc:  # the baseline is informative; never lose the GPU numbers to it
                out["cpu_baseline"] = {"value": None, "unit": "iterations/s", "cores": usable_cores(), "kind": "port",
                                       "sample": f"failed: {type(exc).__name__}: {exc}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
