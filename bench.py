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

* `value` / `dtype`: the DEFAULT precision mode, bf16x3 -- every large GEMM of both passes on 3 x 3 bf16 planes, 6 MFMA plane
  products, >= 24 significant bits per product: the like-for-like arithmetic to the reference's float32 inference
  (`uma_pysis.py:229,246-250`) on the 16-bit matrix cores;
* `roofline`: the dominant kernel family = the split-precision LDS-DMA GEMMs (SO(2) / radial linears and their transposes),
  timed live with HIP events on the launch stream.  `achieved` / `frac` are ALGORITHMIC: 2*M*N*K per product (SURVEY.md
  8d / Appendix D) over the measured kernel time, against the dense 16-bit MFMA peak.  The plane products of the split (x6 in
  both passes) are emulation overhead, reported separately as `mfma_pipe_util` (executed FLOPs / peak).
  `traffic` (HBM bytes per launch from rocprofv3 PMC passes) is only emitted when the committed summary under profiles/
  was measured on EXACTLY this build (source digest compiled into libumx.so), else null + `traffic_source` says why;
* `serial_schedule`: only when the timed region ran on TWO LANES (`UMX_STREAMS=2` / `UMX_LANES_AUTO_EDGES`: two chunks in flight, the
  matrix segments of one beside the HBM-bound segments of the other; bitwise the same results; off by default since the gain went with
  the LS forward kernels).  Kernels of different families then overlap, so "the step is the sum of its families" is measured on a side
  run with `UMX_STREAMS=1`; on the default single lane the families of the timed region itself add up and `roofline.hbm_regime` is theirs;
* `shard`: what ONE rank of the 8-GPU headline run does per iteration -- the 2-image batch through the same evaluator plus the RCCL
  all-gather of [E | status | F] (a one-rank nccl group here: the collective runs, on this GPU alone) -- driver-timed; the it/s derived
  from it is a PROJECTION, not a measurement;
* `--config c1|c2|c4|c5`: the other BASELINE configs through the same evaluator (`config.workload` names them; c4 adds the
  finite-difference Hessian loop of `uma_pysis.py:595-686` on a bounded sample of its 2 x 3N_active displaced geometries);
* `split_bf16_mode`: the same workload with the headline mode's forward pass and the fast mode's 16-bit reverse pass (`UMX_PRECISION=split-bf16`:
  the headline mode's energy bit for bit).
* `fast_mode`: the same workload in the opt-in fast mode (`UMX_PRECISION=split`: 22-23-bit forward activations, 16-bit reverse
  products -- narrower than float32, hence not the headline) and `fp32_mode`: on the fp32 MFMA (`UMX_PRECISION=fp32`), a few
  steps each, timed the same way (N=1 only);
* `gsm` (with `--driver gsm`): cycles of the real driver, `gsm.GrowingStringDriver`, on the fully grown string, and what the
  driver adds on top of the bare evaluation;
* `cpu_baseline`: the repo's own CPU restatement (the reference's fairchem path cannot run here) on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# Runtime environment BEFORE anything can initialise HSA / RCCL (an HSA_* variable set after torch.cuda has come up has no effect).
# dmabuf IPC is the only form this pool's host driver supports; without it RCCL's peer exchange fails with hipIpcGetMemHandle.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from pdb2reaction_amd import synth, weights as W  # noqa: E402
from pdb2reaction_amd._calculator_base import ANG2BOHR  # noqa: E402
from pdb2reaction_amd.engine import Engine  # noqa: E402
from pdb2reaction_amd.parallel import EngineStringEvaluator  # noqa: E402
from pdb2reaction_amd.string import string_step  # noqa: E402

FLOP_PER_EDGE = 30.98e6          # algorithmic E+F work per directed edge
PEAK_FP32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense f32 matrix peak
PEAK_BF16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA peak (not the 2:1-sparse headline)
PEAK_FP64_MFMA_TFLOPS = 78.6     # AMD MI355X datasheet: f64 matrix = f64 vector peak (the micro-arch guide has no f64 row)


MFMA_ONLY_TFLOPS = 1660.0      # executed bf16 plane products / s of the GEMM kernel stripped to its MFMAs (profiles/r06_gemm_ablation_gauss3.txt)
PMC_SUMMARY_GLOB = "r*_pmc_hbm_traffic_{mode}.json"      # profiles/: one per round; the one measured on the running build is taken


def pmc_summary(build_digest: str, mode: str):
    """(summary dict | None, source note).  bench.py cannot collect PMC counters itself (rocprofv3 wraps the process), so HBM
    traffic comes from the committed summary of two PMC passes over this very command -- but ONLY when that summary was
    measured on the build that is running now (`csrc_sha256` == the digest compiled into libumx.so).  A kernel edit makes
    the figure vanish from the line instead of going stale."""
    import glob

    here = os.path.dirname(os.path.abspath(__file__))
    stale = None
    for path in sorted(glob.glob(os.path.join(here, "profiles", PMC_SUMMARY_GLOB.format(mode=mode))), reverse=True):      # newest round first
        try:
            with open(path) as f:
                d = json.load(f)
        except Exception:
            continue
        rel = os.path.relpath(path, here)
        if d.get("csrc_sha256") == build_digest:
            return d, f"{rel} @ build {build_digest[:12]} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 x2 FETCH correction)"
        stale = stale or f"{rel} is stale: measured on build {str(d.get('csrc_sha256'))[:12]}, running {build_digest[:12]}"
    return None, stale or f"no PMC summary for mode {mode}"


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


CONFIGS = {     # BASELINE.json configs: (atoms, images, what it is)
    "c1": (50, 8, "c1: 50-atom small molecule, GSM 8 images (BASELINE: plumbing)"),
    "c2": (500, 12, "c2: ~500-atom active-site cluster, GSM 12 images, 1 GPU"),
    "c3": (2000, 16, "c3: ~2000-atom active-site cluster, GSM 16 images (the config BASELINE's metric is quoted on)"),
    "c4": (2000, 24, "c4: ~2000-atom cluster, DMF path_opt 24 images (+ the freq FD Hessian, `hessian` object)"),
    "c5": (20000, 8, "c5: ~20 000-atom protein-ligand complex, 8 images (one image per chunk on one GPU; one image per GPU on the node)"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c3", help="BASELINE config (default c3, the headline); --atoms / --images override its sizes")
    ap.add_argument("--atoms", type=int, default=None)
    ap.add_argument("--images", type=int, default=None)
    ap.add_argument("--no-shard", action="store_true", help="skip the `shard` leg (2-image batch + one-rank RCCL all-gather)")
    ap.add_argument("--shard-steps", type=int, default=20)
    ap.add_argument("--no-serial", action="store_true", help="skip the `serial_schedule` side run (UMX_STREAMS=1)")
    ap.add_argument("--hessian-sample-atoms", type=int, default=200, help="c4: atoms whose 3 DOF columns the FD-Hessian sample builds (2 x 3 x this many displaced geometries); 0 = the WHOLE Hessian of the config (every active column, device entry only: ~6.5 minutes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp32-mode", action="store_true")
    ap.add_argument("--no-fast-mode", action="store_true")
    ap.add_argument("--driver", choices=["string", "gsm"], default="gsm",
                    help="gsm (default, N=1 only): additionally time cycles of the real driver, gsm.GrowingStringDriver, on the fully grown string "
                         "(device resident) -> `gsm` object; string: skip that leg")
    ap.add_argument("--gsm-cycles", type=int, default=8)
    ap.add_argument("--fp32-steps", type=int, default=5)
    ap.add_argument("--fp32-warmup", type=int, default=2)
    args = ap.parse_args()
    if args.atoms is None:
        args.atoms = CONFIGS[args.config][0]
    if args.images is None:
        args.images = CONFIGS[args.config][1]
    headline = args.config == "c3" and args.atoms == 2000 and args.images == 16

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
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    n, k = args.atoms, args.images
    z, imgs, frozen = synth.make_images(n, k)
    weights = W.make_synthetic_weights(0)
    kl_max = -(-k // world)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def make_engine(precision: str, lanes: str = ""):
        os.environ["UMX_PRECISION"] = precision              # read by umx_load_weights
        if lanes:
            os.environ["UMX_STREAMS"] = lanes                # read by umx_create ("" = the engine's own choice)
        try:
            eng = Engine(local_rank)
        finally:
            if lanes:
                os.environ.pop("UMX_STREAMS", None)
        eng.load_weights(weights)
        eng.set_system(z, charge=0, spin=1, task="omol")
        return eng

    def run(precision: str, steps: int, warmup: int, lanes: str = "", images=None, force_collective: bool = False):
        """W untimed + K timed string iterations on a fresh engine in `precision` mode; returns (dt, profile, edges, maxdeg, resolved mode).
        `images`: another batch of the same system (the 2-image shard); `force_collective`: issue the all-gather in a one-rank group too."""
        eng = make_engine(precision, lanes)
        imgs_run = imgs if images is None else images
        k = len(imgs_run)
        k_full = len(imgs)
        x = torch.as_tensor(imgs * ANG2BOHR, dtype=torch.float64, device=dev).reshape(k_full, -1)   # string state: Bohr, float64, in HBM (the WHOLE string)
        # the device evaluator of parallel.py: float32 Angstrom positions into the engine's device-pointer entry on torch's current stream,
        # frozen rows zeroed, Hartree / Bohr out, the k images sharded over the ranks + ONE all-gather.  check="deferred": the gathered
        # energies are checked on the device and the flag is read behind the engine's own per-call synchronisation (no extra host sync)
        ev = EngineStringEvaluator(eng, n, dev, frozen=frozen, check="deferred", max_images=(kl_max if images is None else k), force_collective=force_collective)

        def step(xc):
            if images is None:
                e, f = ev(xc)
            else:
                # the shard leg: this rank's k images are evaluated (+ the gather), then the REPLICATED update of the whole string runs as on
                # every rank of the 8-GPU job -- with the other ranks' forces stood in for by copies of the local ones
                e, fl = ev(xc[:k])
                f = fl.repeat(-(-k_full // k), 1)[:k_full]
            return string_step(xc, f, max_step=0.1, alpha=0.5, fix_ends=False), e

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
        ev.flush()                                      # the deferred check of the last step
        ne_local, maxdeg = eng.graph_stats()            # edges of this rank's images in the last step
        if not bool(torch.isfinite(e).all()) or eng.widened:
            raise SystemExit("bench.py: non-finite energies / a precision change inside the timed region")
        resolved = eng.precision_mode()
        lanes_used = eng.last_lanes()
        eng.close()
        return dt, prof, ne_local, maxdeg, resolved, lanes_used

    def run_gsm(precision: str, cycles: int, warmup: int):
        """The REAL driver: `gsm.GrowingStringDriver` on a fully grown k-image string (reference construction:
        path_opt.py:959-977, GS_KW :168-185, `--fix-ends` default False :663-668 so all k images move), device resident, through the
        same sharded device evaluator -- `cycles` cycles with the climbing image off, then on (plain string tangent: the Lanczos
        tangent of climb_lanczos adds serial single-image evaluations, not driver work).  Beside it the bare evaluation of the same
        batches: the difference is everything the driver adds per cycle (device math, its one 2K+8-double read, Python)."""
        from pdb2reaction_amd.gsm import GrowingStringDriver

        eng = make_engine(precision)
        ev = EngineStringEvaluator(eng, n, dev, frozen=frozen, check="deferred", max_images=kl_max)
        x0 = (imgs * ANG2BOHR).reshape(k, -1)
        elem = [synth.SYMBOLS[int(v)] for v in z]
        res = {}
        xt = torch.as_tensor(x0, dtype=torch.float64, device=dev)
        for _ in range(warmup):
            ev(xt)
        fence()
        t0 = time.perf_counter()
        for _ in range(cycles):
            ev(xt)
        fence()
        res["evaluation_only_ms"] = (time.perf_counter() - t0) / cycles * 1e3
        # single-image evaluation (what one step of the Lanczos recursion costs: gsm._single_forces)
        for _ in range(2):
            ev(xt[:1])
        fence()
        t0 = time.perf_counter()
        for _ in range(6):
            ev(xt[:1])
        fence()
        res["single_image_ms"] = (time.perf_counter() - t0) / 6 * 1e3
        lz = {"climb": True, "climb_rms": 1e9, "climb_lanczos": True, "climb_lanczos_rms": 1e9}       # the reference's defaults (path_opt.py:179-182), thresholds forced so that the phase runs
        for leg, gs_kw in (("climb_off", {"climb": False}), ("climb_on", {"climb": True, "climb_rms": 1e9, "climb_lanczos": False}),
                           ("climb_lanczos_cold", {**lz, "climb_lanczos_warm_start": False}), ("climb_lanczos", lz)):
            drv = GrowingStringDriver(elem, x0[0], x0[-1], evaluate_device=ev, device=dev, images=x0,
                                      gs_kw={"max_nodes": k - 2, "fix_first": False, "fix_last": False, **gs_kw},
                                      stopt_kw={"max_cycles": cycles + warmup, "thresh": "gau_vtight", "max_step": 0.1, "print_every": 10 ** 9})
            # warm-up cycles are part of the same run (the L-BFGS history must exist): time the LAST `cycles` cycles
            marks = []
            orig = drv._raw_eval

            def timed_eval(xq, _orig=orig, _marks=marks):
                if xq.shape[0] == k:
                    torch.cuda.synchronize()
                    _marks.append(time.perf_counter())
                return _orig(xq)

            drv._raw_eval = timed_eval
            out = drv.run()
            fence()
            t_end = time.perf_counter()
            # marks[i] = start of cycle i+1's evaluation (= end of cycle i); the run ends with one final evaluation of the stepped string
            span = (marks[-1] - marks[warmup]) / max(len(marks) - 1 - warmup, 1)
            res[leg] = {"cycle_ms": span * 1e3, "cycles_timed": len(marks) - 1 - warmup, "redo_steps": out.timing["redo_steps"],
                        "fully_grown": bool(out.fully_grown), "images": int(len(out.coords)), "t_end_minus_last_mark_ms": (t_end - marks[-1]) * 1e3}
            if gs_kw.get("climb_lanczos"):
                calls = max(out.timing["lanczos_calls"], 1.0)
                res[leg].update({"lanczos_evals": int(out.timing["lanczos_evals"]), "lanczos_recursions": int(out.timing["lanczos_calls"]),
                                 "lanczos_evals_per_cycle": out.timing["lanczos_evals"] / calls, "warm_kept": int(out.timing["lanczos_warm_calls"]),
                                 "warm_rejected": int(out.timing["lanczos_warm_rejected"]), "cycles_run": int(out.cycles),
                                 "lowest_ritz_value_and_overlap_with_tangent": [[round(w_, 5), round(o_, 3), n_] for w_, o_, n_, _ in drv.lanczos_log[:12]]})
        ev.flush()
        eng.close()
        shard_ms = res["evaluation_only_ms"] / 8.0
        for leg in ("climb_lanczos_cold", "climb_lanczos"):
            r_ = res[leg]
            # what the serial single-image probes add: on one GPU (measured: cycle_ms), and PROJECTED for one rank of the 8-GPU run -- the probes on
            # one rank while seven wait (gp_singles=False), or graph-parallel over the eight ranks (parallel.EngineStringEvaluator, the default;
            # its 10 all-reduces per probe cannot be timed on one GPU)
            r_["serial_probe_ms_per_cycle"] = r_["lanczos_evals_per_cycle"] * res["single_image_ms"]
            r_["projected_8gpu_cycle_ms_probes_on_one_rank"] = shard_ms + r_["serial_probe_ms_per_cycle"]
            r_["projected_8gpu_cycle_ms_probes_graph_parallel_compute_only"] = shard_ms + r_["serial_probe_ms_per_cycle"] / 8.0
        for leg in ("climb_off", "climb_on"):
            ov = res[leg]["cycle_ms"] - res["evaluation_only_ms"]
            res[leg]["driver_overhead_ms"] = ov
            res[leg]["overhead_share_of_cycle"] = ov / res[leg]["cycle_ms"]
            res[leg]["overhead_vs_2_image_shard"] = ov / shard_ms
        res["note"] = ("gsm.GrowingStringDriver (device resident: tangents, projection, L-BFGS, reparametrisation as torch ops on the GPU, one read of "
                       "2K+8 doubles per cycle) on the fully grown string through parallel.EngineStringEvaluator; evaluation_only = the same batches "
                       "through the evaluator alone; driver_overhead = cycle - evaluation_only; overhead_vs_2_image_shard = overhead / (evaluation_only / 8), "
                       "the share it would have of a cycle of the 8-GPU run (the string update is replicated on every rank).  climb_lanczos(_cold): the "
                       "reference's DEFAULT climbing phase (climb=True, climb_lanczos=True, path_opt.py:179-182; thresholds forced so that it runs on the "
                       "synthetic string): every cycle adds a Lanczos recursion of serial single-image gradients -- started from the string tangent (cold) "
                       "or from last cycle's mode (the default; kept only while its curvature is negative, every 10th recursion cold); "
                       "lowest_ritz_value_and_overlap_with_tangent lists [Ritz value, overlap, gradients] per recursion -- on the SYNTHETIC string with "
                       "random weights the lowest mode at the HEI is orthogonal to the path for the cold recursion too; projected_* are PROJECTIONS from "
                       "shard = evaluation_only / 8")
        return res

    def run_hessian(precision: str, sample_atoms: int):
        """c4's "freq Hessian (3N force batches)" (uma_pysis.py:595-686: 2 central-difference force calls per active DOF, h = 1e-3 A): the batched
        loop `hessian.fd_hessian` on a BOUNDED sample -- the columns of `sample_atoms` atoms (every other atom frozen for the sample, which
        changes the number of columns, not the cost of one: each displaced geometry is a full 2000-atom E+F) -- through BOTH entries: the
        device-resident one `uma_pysis` uses since round 6 (displaced geometries built on the GPU, forces never leave it) and the host one
        (PCIe copies of 64 x N x 3 floats each way per call).  Per-call times give the spread; the extrapolation to the 5940 active columns of
        the config is stated as such."""
        from pdb2reaction_amd.hessian import fd_hessian
        from pdb2reaction_amd.uma_pysis import UMAcore

        eng = make_engine(precision)
        whole = sample_atoms <= 0
        sample = sorted(set(range(n)) - set(int(i) for i in frozen)) if whole else list(range(0, n, max(1, n // sample_atoms)))[:sample_atoms]
        frz = sorted(set(range(n)) - set(sample))
        x0 = imgs[k // 2]
        cols = 3 * len(sample)
        full_cols = 3 * (n - len(frozen))
        core = UMAcore.__new__(UMAcore)              # the calculator core's device batch entry on this engine (no second engine, no weights reload)
        core.engine = eng
        res = {}
        h_ref = None
        for entry in (("device",) if whole else ("device", "host")):
            calls = {"n": 0, "geoms": 0, "edges": 0, "t": []}

            def batch_forces(disp, _c=calls):
                torch.cuda.synchronize(); t = time.perf_counter()
                f = eng.energy_forces(disp)[1]
                _c["t"].append((time.perf_counter() - t) / len(disp) * 1e3)
                _c["n"] += 1; _c["geoms"] += len(disp); _c["edges"] += eng.graph_stats()[0]
                return f

            def batch_forces_dev(pos32, _c=calls):
                torch.cuda.synchronize(); t = time.perf_counter()
                f = core.compute_batch_dev(pos32)
                torch.cuda.synchronize()
                _c["t"].append((time.perf_counter() - t) / len(pos32) * 1e3)
                _c["n"] += 1; _c["geoms"] += len(pos32); _c["edges"] += eng.graph_stats()[0]
                if _c["n"] % 20 == 0:
                    print(f"[hessian] {_c['geoms']} displaced geometries, {_c['t'][-1]:.2f} ms each", file=sys.stderr, flush=True)
                return f

            eng.reserve_images(64)
            batch_forces(np.repeat(x0[None], 64, axis=0))                     # warm-up: workspace for 64-image batches
            calls.update(n=0, geoms=0, edges=0, t=[])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            h = fd_hessian(batch_forces, x0, frz, device=dev, double=True, partial=False, batch=64, engine=eng,
                           batch_forces_dev=batch_forces_dev if entry == "device" else None)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ok = bool(torch.isfinite(h).all())
            if h_ref is None:
                h_ref = h
            per = np.asarray(calls["t"][:-1] if len(calls["t"]) > 1 and calls["geoms"] % 64 else calls["t"])     # (a ragged last call is left out of the spread)
            res[entry] = {"columns": cols, "displaced_geometries": calls["geoms"], "engine_calls": calls["n"], "seconds": dt, "columns_per_s": cols / dt,
                          "ms_per_displaced_geometry": dt / calls["geoms"] * 1e3,
                          "ms_per_displaced_geometry_by_call": {"min": float(per.min()), "median": float(np.median(per)), "max": float(per.max()), "calls": int(per.size)},
                          "finite": ok, "extrapolated_full_hessian_s": full_cols / (cols / dt), "directed_edges": calls["edges"],
                          "algorithmic_tflops": FLOP_PER_EDGE * calls["edges"] / dt / 1e12,
                          "same_columns_as_device_entry": bool(torch.equal(h, h_ref))}
        eng.close()
        d, hst = res["device"], res.get("host")
        out_h = dict(d)
        out_h.update({"entry": "device (uma_pysis.get_hessian's default since round 6: hessian.fd_hessian(batch_forces_dev=UMAcore.compute_batch_dev))",
                      "batch": 64, "full_hessian_columns": full_cols, "sample_share_of_full_hessian": cols / full_cols,
                      "whole_hessian_measured": whole,
                      "host_entry": hst, "host_minus_device_ms_per_geometry": (hst["ms_per_displaced_geometry"] - d["ms_per_displaced_geometry"]) if hst else None,
                      "projected_8gpu_full_hessian_s": d["extrapolated_full_hessian_s"] / 8.0,
                      "columns_per_s_per_rank": d["columns_per_s"],
                      "note": f"hessian.fd_hessian, {cols} columns = {d['displaced_geometries']} displaced {n}-atom geometries in batches of 64 "
                              f"({cols / full_cols:.0%} of the {full_cols} active columns of c4); extrapolated_full_hessian_s scales columns/s to all of them "
                              "(an extrapolation of the same loop, not a second measurement); projected_8gpu_full_hessian_s = that / 8: the columns are dealt "
                              "k mod 8 over the ranks (fd_hessian(shard=True)), every rank builds its share at columns_per_s_per_rank, one all-reduce of the "
                              "(3N)^2 float64 matrix (288 MB) at the end is NOT in it -- a PROJECTION, RCCL has never run on more than one rank here"})
        return out_h

    mode_req = os.environ.get("UMX_PRECISION", "auto")           # "auto" = bf16x3: >= 24-bit products in both passes (include/umx.h)
    dt, prof, ne_local, maxdeg, resolved, lanes_used = run(mode_req, args.steps, args.warmup)
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
    DTYPE = {"bf16x3": ("bf16x3-split", "every large GEMM of BOTH passes on 3 x 3 bf16 planes (exact split of the float32 operands), the 6 plane products of "
                        "order <= 2 on v_mfma_f32_*_bf16, fp32 accumulate: >= 24 significant bits per product, the like-for-like arithmetic to the "
                        "reference's float32; node-level linears float64-accumulated; everything else fp32"),
             "split": ("f16-split", "forward GEMMs: 2 fp16 activation planes x 3 exact fp16 weight planes, 4 MFMA products (22-23 bit activations); reverse GEMMs: "
                       "2 x 2 bf16 planes, 3 products (16-bit) -- NARROWER than the reference's float32; fp32 accumulate; node-level linears float64-accumulated"),
             "split-bf16": ("bf16-split", "forward GEMMs: 3 x 3 bf16 planes, 6 MFMA products (24-bit); reverse GEMMs: 2 x 2 bf16 planes, 3 products (16-bit); "
                            "fp32 accumulate; node-level linears float64-accumulated; everything else fp32"),
             "fp32": ("f32", "every GEMM on v_mfma_f32_32x32x2_f32")}

    if rank == 0:
        from pdb2reaction_amd.engine import load_library

        digest = load_library().umx_build_digest().decode()
        pl, f32 = prof["split_bf16"], prof["fp32"]
        split = pl["launches"] > 0
        dom = pl if split else f32
        peak = PEAK_BF16_MFMA_TFLOPS if split else PEAK_FP32_MFMA_TFLOPS
        alg = dom["alg_flops"] / max(dom["ms"], 1e-9) / 1e9          # algorithmic TFLOP/s of the dominant family (2*M*N*K per product)
        executed = dom["mfma_flops"] / max(dom["ms"], 1e-9) / 1e9    # what the matrix cores executed (the plane products of the split)
        pmc, pmc_note = pmc_summary(digest, mode) if (headline and world == 1 and split) else (None, "PMC summary exists for c3 / 1 GPU / split modes only")
        out = {
            "metric": "path_opt_string_iterations_per_s", "value": it_s, "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": DTYPE[mode][0], "dtype_detail": DTYPE[mode][1],
            "precision_mode": mode, "precision_requested": mode_req,
            "data": "synthetic",
            "image_atom_steps_per_s": k * n * it_s,
            "algorithmic_tflops": FLOP_PER_EDGE * edges_iter * it_s / 1e12,
            "build_digest": digest,
            "config": {"workload": f"{CONFIGS[args.config][2] if (n, k) == CONFIGS[args.config][:2] else args.config + ' (sizes overridden)'}: {n}-atom synthetic "
                                   f"active-site cluster x {k} images, GSM-style string iteration (batched UMA-S E+F of all images + string update), "
                                   f"UMA-S shapes, synthetic weights",
                       "name": args.config, "atoms": n, "images": k, "directed_edges_per_iteration": int(edges_iter), "max_degree": maxdeg,
                       "parallelism": f"images sharded {k}/{world} per GPU, 1 all-gather/iteration" if world > 1 else "single GPU, all images batched"},
            "roofline": {"bound": "mfma", "achieved": alg, "peak": peak, "unit": "TFLOP/s", "frac": alg / peak,
                         "definition": "achieved = algorithmic FLOPs (2*M*N*K per product, SURVEY.md 8d) of the family / its HIP-event time; "
                                       "mfma_pipe_util counts the plane products actually executed (bf16x3: x6 in both passes; fast mode: fwd x4 fp16, reverse x3)",
                         "mfma_pipe_util": executed / peak, "executed_tflops": executed,
                         # (a constant from a committed profile, not measured by this run: the rate of the six-product kernel with everything but
                         #  its MFMAs removed -- what the matrix pipe delivers at the clock this chip holds under its power cap)
                         "mfma_only_rate_under_power_cap": ({"tflops": MFMA_ONLY_TFLOPS, "frac": executed / MFMA_ONLY_TFLOPS,
                                                             "source": "profiles/r06_gemm_ablation_gauss3.txt (UMX_GEMM_ABL=31)"} if split else None),
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
        # Two lanes (UMX_STREAMS=2 / UMX_LANES_AUTO_EDGES; not the default) overlap kernels of different families, so the remainder of the
        # step is no longer "the other kernels": the family breakdown comes from a serial-schedule side run (UMX_STREAMS=1) of the same
        # workload, the headline stays what the product does by default.
        two_lanes = lanes_used == 2
        ser = None
        if world == 1 and two_lanes and not args.no_serial:
            try:
                sdt, sprof, _, _, _, _ = run(mode_req, max(2, min(args.steps, 5)), 1, lanes="1")
                ssteps = max(2, min(args.steps, 5))
                ser = {"ms_per_step": sdt / ssteps * 1e3, "steps": ssteps, "prof": sprof}
            except Exception as exc:
                out["serial_schedule"] = {"error": f"{type(exc).__name__}: {exc}"}
        bprof, bsteps, bms = (ser["prof"], ser["steps"], ser["ms_per_step"]) if ser else (prof, args.steps, ms)
        bdom = bprof["split_bf16"] if split else bprof["fp32"]
        bf32 = bprof["fp32"]
        rad = bprof["radial"]
        rad_ms = rad["ms"] / bsteps
        rest_ms = bms - (bdom["ms"] + (bf32["ms"] if split else 0.0)) / bsteps
        edge_ms = rest_ms - rad_ms
        if ser:
            out["serial_schedule"] = {"ms_per_step": bms, "steps": bsteps, "gemm_family_ms_per_step": bdom["ms"] / bsteps, "node_gemm_ms_per_step": bf32["ms"] / bsteps if split else 0.0,
                                      "radial_ms_per_step": rad_ms, "hbm_regime_ms_per_step": edge_ms, "two_lane_gain_ms": bms - ms,
                                      "gemm_family_tflops": bdom["alg_flops"] / max(bdom["ms"], 1e-9) / 1e9,
                                      "note": "UMX_STREAMS=1 side run of the same workload: one lane, kernels strictly one after another, so the step is the plain sum of its "
                                              "families (roofline.hbm_regime is taken from it); the headline above ran on two lanes (requested through the environment; "
                                              "bitwise the same results)"}
        out["roofline"]["lanes"] = lanes_used
        if ser:
            # the timed region ran on two lanes: a GEMM launch there shares the chip with the other lane's HBM-bound kernels and takes longer than alone
            # (that is the price of the overlap; the step is shorter all the same).  Both figures, so that neither hides the other:
            out["roofline"]["schedule"] = "two lanes (requested through UMX_STREAMS / UMX_LANES_AUTO_EDGES): per-launch times include the slowdown from co-running HBM-bound kernels"
            out["roofline"]["serial_schedule_achieved"] = bdom["alg_flops"] / max(bdom["ms"], 1e-9) / 1e9
            out["roofline"]["serial_schedule_frac"] = out["roofline"]["serial_schedule_achieved"] / peak
        hb = {"bound": "hbm", "kernels": "HBM-bound edge / node kernels: k_gather_rotate_mod_q3, k_modrot_bwd_pl, k_gate_edge_*, k_rotate_back_*, norms, graph build "
                                         "(everything outside the GEMM families and the fused radial-MLP kernels)",
              "ms_per_step": edge_ms, "schedule": "serial side run (UMX_STREAMS=1)" if ser else "the timed region (one lane)",
              "peak": PEAK_HBM_GBPS, "unit": "GB/s", "traffic_per_step": None, "achieved": None, "frac": None,
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

        def side_mode(name: str, steps: int, warmup: int, fam: str, peak_tf: float, detail: str):
            """The same workload in another precision mode: `value` on the engine's default schedule (as the headline); the GEMM family's
            figures from a serial-schedule run when the default schedule overlaps kernels (two lanes), else from the same run."""
            try:
                dts, profs, _, _, res_mode, lanes_s = run(name, steps, warmup)
                res = {"value": steps / dts, "unit": "iterations/s", "ms_per_step": dts / steps * 1e3, "steps": steps, "warmup": warmup,
                       "precision_mode": res_mode, "dtype": detail, "lanes": lanes_s}
                gsteps, gsched = steps, "the timed run (one lane)"
                if lanes_s == 2:
                    gsteps = 2
                    _, profs, _, _, _, _ = run(name, gsteps, 1, lanes="1")
                    gsched = "serial side run (UMX_STREAMS=1, 2 steps)"
                gg = profs[fam]
                res.update({"gemm_schedule": gsched, "gemm_ms_per_step": gg["ms"] / gsteps,
                            "gemm_tflops": gg["alg_flops"] / max(gg["ms"], 1e-9) / 1e9, "gemm_peak": peak_tf,
                            "gemm_frac": gg["alg_flops"] / max(gg["ms"], 1e-9) / 1e9 / peak_tf,
                            "executed_tflops": gg["mfma_flops"] / max(gg["ms"], 1e-9) / 1e9})
                return res
            except Exception as exc:
                return {"value": None, "error": f"{type(exc).__name__}: {exc}"}

        if world == 1 and not args.no_fast_mode and mode != "split":
            # the FAST mode (opt-in, UMX_PRECISION=split): 22-23 bit forward activations, 16-bit reverse products -- narrower than the
            # reference's float32, so it is NOT the headline; tolerances are met with margin (tests/test_gpu_baseline_sizes.py)
            out["fast_mode"] = side_mode("split", args.fp32_steps, args.fp32_warmup, "split_bf16", PEAK_BF16_MFMA_TFLOPS,
                                         "f16-split (fwd 2 x 3 fp16 planes / 4 products, reverse 2 x 2 bf16 planes / 3 products): narrower than float32")
        if world == 1 and not args.no_fast_mode and mode == "bf16x3":
            # split-bf16: the headline mode's forward pass (its ENERGY bit for bit) with the fast mode's 16-bit reverse pass (forces <= 7e-6 eV/A off)
            out["split_bf16_mode"] = side_mode("split-bf16", args.fp32_steps, args.fp32_warmup, "split_bf16", PEAK_BF16_MFMA_TFLOPS,
                                               "bf16-split (fwd as the headline: 3 x 3 bf16 planes / 6 products; reverse 2 x 2 bf16 planes / 3 products): "
                                               "the headline mode's energy, a 16-bit reverse pass")
        if world == 1 and split and not args.no_fp32_mode:
            # every GEMM on v_mfma_f32_32x32x2_f32: the same float32 products as the headline mode's, on the fp32 matrix pipe
            out["fp32_mode"] = side_mode("fp32", args.fp32_steps, args.fp32_warmup, "fp32", PEAK_FP32_MFMA_TFLOPS, "f32 (all GEMMs on v_mfma_f32_32x32x2_f32)")
        os.environ["UMX_PRECISION"] = mode_req
        shard_images = {"c3": 2, "c4": 3, "c5": 1}.get(args.config) if (n, k) == CONFIGS[args.config][:2] else None
        if world == 1 and shard_images and not args.no_shard:
            # what ONE rank of the 8-GPU run does per iteration (BASELINE c3: "16 images sharded 2-images/GPU across 8 x MI355X"): the 2-image batch
            # through the same evaluator + the all-gather, issued for real on a one-rank RCCL group (nccl backend; the identity, but the call, its
            # stream ordering and its latency are there).  Driver-timed; any it/s derived from it is a projection of the 8-GPU run, not a measurement.
            try:
                own_group = False
                if not dist.is_initialized():
                    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{29400 + os.getpid() % 500}", rank=0, world_size=1, device_id=dev)
                    own_group = True
                ssteps = args.shard_steps if args.config != "c5" else max(3, args.shard_steps // 4)
                sdt, _, se, _, _, _ = run(mode_req, ssteps, 3 if args.config != "c5" else 1, images=imgs[:shard_images], force_collective=True)
                sdt0, _, _, _, _, _ = run(mode_req, ssteps, 3 if args.config != "c5" else 1, images=imgs[:shard_images], force_collective=False)
                if own_group:
                    dist.destroy_process_group()
                sms = sdt / ssteps * 1e3
                out["shard"] = {"ms_per_step": sms, "steps": ssteps, "warmup": 3 if args.config != "c5" else 1, "images": shard_images, "directed_edges": int(se),
                                "ms_per_step_without_collective": sdt0 / ssteps * 1e3,
                                "collective": "all_gather_into_tensor of [E | status | F] float64 rows on a ONE-rank nccl (RCCL) group, every step",
                                "projected_8gpu_iterations_per_s": 1e3 / sms,
                                "note": f"the per-rank share of the 8-GPU run of this config ({shard_images} of the {k} images + the gather + the replicated string step), timed like the "
                                        "headline; projected_8gpu_iterations_per_s = 1 / this is a PROJECTION (xGMI all-gather latency of 8 ranks and rank skew "
                                        "are not in it), not a measurement -- RCCL has never run on more than one rank here"}
            except Exception as exc:
                out["shard"] = {"error": f"{type(exc).__name__}: {exc}"}
        if world == 1 and args.config == "c4":
            try:
                out["hessian"] = run_hessian(mode_req, args.hessian_sample_atoms)
            except Exception as exc:
                out["hessian"] = {"error": f"{type(exc).__name__}: {exc}"}
        if world == 1 and args.driver == "gsm":
            try:
                out["gsm"] = run_gsm(mode_req, args.gsm_cycles, 3)
            except Exception as exc:
                out["gsm"] = {"error": f"{type(exc).__name__}: {exc}"}
        if world == 1 and not args.no_cpu_baseline and args.config == "c3":
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
