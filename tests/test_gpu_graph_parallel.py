"""Graph-parallel single-image mode (rows a12 / f4): the reference's ``workers > 1`` semantics -- the graph of ONE image partitioned
over ranks (``uma_pysis.py:220-242``) -- on the engine's exchange-point API (umx_gp_begin / umx_gp_step).

World size 1: the partial-sum path with all edges local must reproduce the ordinary evaluation bit for bit -- also with the all-reduces
actually issued through RCCL (backend nccl, one rank) IN PLACE on the engine's workspace memory.
World size 2 and 3: ranks share the one GPU of the test box (gloo group, host-staged all-reduce: the collective the RCCL path does on
device); every rank must end with the ORACLE's energy and forces (float64 restatement, BASELINE tolerances) and with the same numbers
as every other rank; a rank that owns no atoms / no edges (more ranks than atoms) takes part with zero partial sums."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pdb2reaction_amd import synth, weights as W

pytestmark = pytest.mark.gpu


def _single(z, pos, precision=None):
    from pdb2reaction_amd.engine import Engine

    eng = Engine(0, precision=precision)
    eng.load_weights(W.make_synthetic_weights(0))
    eng.set_system(z)
    e, f = eng.energy_forces(pos)
    ne = eng.graph_stats()[0]
    return eng, e, f, ne


def test_world_size_one_matches_the_ordinary_path():
    """One rank, no collective: the same kernels as the ordinary evaluation except at the ten exchange points, where the rank's edge sums
    are rounded to the float32 that travels between the ranks before the residual is added (the ordinary path adds residual and edge sum
    in double and rounds once, round 5) -- agreement to float32 rounding of the node state, bitwise reproducible, and the engine is bitwise
    its ordinary self afterwards."""
    from pdb2reaction_amd.parallel import GraphParallelEvaluator

    z, imgs, _ = synth.make_images(120, 2, seed=5)
    eng, e, f, _ = _single(z, imgs)
    dev = torch.device("cuda", 0)
    gp = GraphParallelEvaluator(eng, len(z), dev)
    for k in range(2):
        ek, fk = gp(torch.as_tensor(imgs[k], dtype=torch.float32, device=dev))
        assert gp.n_exchanges == 10                                   # edge-degree aggregate + 4 layers x (forward, reverse) + forces
        # (float32 rounding of ten exchanged node aggregates of a -9078 eV system: measured 1.1e-6 eV in round 5, 2.4e-6 with round 6's planes)
        assert abs(float(ek[0]) - e[k]) <= 5e-6 and np.abs(fk.cpu().numpy() - f[k]).max() <= 2e-6
        ek2, fk2 = gp(torch.as_tensor(imgs[k], dtype=torch.float32, device=dev))
        assert float(ek2[0]) == float(ek[0]) and torch.equal(fk2, fk)
    e2, f2 = eng.energy_forces(imgs)                                  # the engine is back in its ordinary mode afterwards
    assert np.array_equal(e2, e) and np.array_equal(f2, f)
    eng.close()


def _worker(rank, world, port, n_atoms, out, precision=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pdb2reaction_amd.engine import Engine
        from pdb2reaction_amd.parallel import GraphParallelEvaluator

        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        z, imgs, _ = synth.make_images(n_atoms, 1, seed=8)
        eng = Engine(0, precision=precision)
        eng.load_weights(W.make_synthetic_weights(0))
        eng.set_system(z)
        gp = GraphParallelEvaluator(eng, n_atoms, dev)
        e, f = gp(torch.as_tensor(imgs[0], dtype=torch.float32, device=dev))
        ne_local = eng.graph_stats()[0]
        eng.close()
        # the calculator boundary: the reference's keyword -- workers = number of ranks switches the graph-parallel mode on by itself
        import importlib
        U = importlib.import_module("pdb2reaction_amd.uma_pysis")
        elem = [synth.SYMBOLS[int(q)] for q in z]
        calc = U.uma_pysis(model="synthetic", freeze_atoms=[1], workers=world, **({"precision": precision} if precision else {}))
        r = calc.get_forces(elem, (imgs[0] * U.ANG2BOHR).reshape(-1))
        assert calc._core._gp is not None and calc._core.parallel_predict
        calc.close()
        out[rank] = (float(e[0]), f.cpu().numpy(), ne_local, (gp.lo, gp.hi), r["energy"] / U.EV2AU, r["forces"].reshape(-1, 3) / U.F_EVAA_2_AU)
    finally:
        dist.destroy_process_group()


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world,n_atoms,precision", [(2, 300, None), (3, 157, None), (2, 120, "split-bf16"), (2, 90, "fp32"), (3, 2, None)])
def test_ranks_partition_one_image(oracle, world, n_atoms, precision):
    z, imgs, _ = synth.make_images(n_atoms, 1, seed=8)
    eng, e_ref, f_ref, ne_all = _single(z, imgs, precision)
    eng.close()
    # the ORACLE is the yardstick (the mode was only ever compared with the single-engine HIP result)
    e_orc, f_orc = oracle.energy_forces(z, np.asarray(imgs[0], dtype=np.float32).astype(np.float64))
    mgr = mp.get_context("spawn").Manager()   # (never FORK a process that has initialised the GPU: the child dies with "Memory in use")
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _port(), n_atoms, out, precision), nprocs=world, join=True)
    assert sorted(out.keys()) == list(range(world))
    assert sum(out[r][2] for r in range(world)) == ne_all               # every directed edge is built by exactly one rank
    assert [out[r][3] for r in range(world)][0][0] == 0 and out[world - 1][3][1] == n_atoms
    if n_atoms < world:
        assert any(out[r][3][0] == out[r][3][1] and out[r][2] == 0 for r in range(world))      # a rank without atoms and edges
    for r in range(world):
        e, f, ne_r, _, e_calc, f_calc = out[r]
        assert abs(e - e_orc) <= 1e-4 and np.abs(f - f_orc).max() <= 1e-3               # BASELINE.json tolerances, per rank
        assert abs(e_calc - e_orc) <= 1e-4 and np.all(f_calc[1] == 0.0)                 # frozen atom zeroed as usual
        act = np.arange(n_atoms) != 1
        if act.any():
            assert np.abs(f_calc[act] - f_orc[act]).max() <= 1e-3
        if n_atoms >= world:
            assert 0 < ne_r < ne_all
        assert abs(e - e_ref[0]) <= 2e-5                                # vs the single engine: summation order of float32 partial sums
        assert np.abs(f - f_ref[0]).max() <= 2e-5
        assert np.array_equal(f, out[0][1]) and e == out[0][0]          # all ranks hold the same reduced result


def _nccl_one_rank(rank, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        from pdb2reaction_amd.engine import Engine
        from pdb2reaction_amd.parallel import GraphParallelEvaluator

        z, imgs, _ = synth.make_images(120, 1, seed=5)
        eng = Engine(0)
        eng.load_weights(W.make_synthetic_weights(0))
        eng.set_system(z)
        e_ord, f_ord = eng.energy_forces(imgs)
        gp0 = GraphParallelEvaluator(eng, len(z), dev)                # one-rank group, no force_collective: exchange points without a collective
        assert not gp0.distributed
        e0t, f0t = gp0(torch.as_tensor(imgs[0], dtype=torch.float32, device=dev))
        e0, f0 = [float(e0t[0])], [f0t.cpu().numpy().copy()]
        out["close"] = bool(abs(e0[0] - e_ord[0]) <= 2e-6 and np.abs(f0[0] - f_ord[0]).max() <= 2e-6)
        gp = GraphParallelEvaluator(eng, len(z), dev, force_collective=True)
        assert gp.distributed and not gp._stage_cpu
        e, f = gp(torch.as_tensor(imgs[0], dtype=torch.float32, device=dev))
        torch.cuda.synchronize()
        out["n_exchanges"] = gp.n_exchanges
        out["bitwise"] = bool(float(e[0]) == e0[0] and np.array_equal(f.cpu().numpy(), f0[0]))
        out["backend"] = dist.get_backend()
        eng.close()
    finally:
        dist.destroy_process_group()


def _nccl_one_rank_gather(rank, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        from pdb2reaction_amd.engine import Engine
        from pdb2reaction_amd.parallel import EngineStringEvaluator
        from pdb2reaction_amd._calculator_base import ANG2BOHR

        z, imgs, _ = synth.make_images(60, 3, seed=6)
        eng = Engine(0)
        eng.load_weights(W.make_synthetic_weights(0))
        eng.set_system(z)
        x = torch.as_tensor(imgs * ANG2BOHR, dtype=torch.float64, device=dev).reshape(3, -1)
        plain = EngineStringEvaluator(eng, len(z), dev, frozen=[2])
        e0, f0 = plain(x)
        for check in ("sync", "deferred"):
            ev = EngineStringEvaluator(eng, len(z), dev, frozen=[2], check=check, force_collective=True)
            e, f = ev(x)
            e2, f2 = ev(x)
            ev.flush()
            torch.cuda.synchronize()
            inner = ev._ev[3]
            out[check] = bool(inner.distributed and not inner._stage_cpu and torch.equal(e, e0) and torch.equal(f, f0) and torch.equal(e2, e0) and torch.equal(f2, f0))
        out["backend"] = dist.get_backend()
        eng.close()
    finally:
        dist.destroy_process_group()


def test_rccl_all_gather_of_the_image_shards():
    """The collective of the image-sharded path: ONE ``all_gather_into_tensor`` of float64 [E | status | F] rows per
    evaluation, here issued by RCCL in a one-rank nccl group (``force_collective``) on the evaluator's device buffers, on torch's current
    stream right behind the engine's kernels -- sync and deferred check; the gathered result must be bitwise the ungathered one."""
    mgr = mp.get_context("spawn").Manager()   # (never FORK a process that has initialised the GPU: the child dies with "Memory in use")
    out = mgr.dict()
    mp.spawn(_nccl_one_rank_gather, args=(_port(), out), nprocs=1, join=True)
    assert out["backend"] == "nccl" and out["sync"] is True and out["deferred"] is True


def test_rccl_all_reduce_in_place_on_engine_memory():
    """The RCCL leg of the graph-parallel mode: with a one-rank nccl group and ``force_collective`` all ten all-reduces are issued by
    RCCL on the caller's stream directly on the engine's workspace buffers (the ``__cuda_array_interface__`` view, no copy).  A
    one-rank all-reduce is the identity, so the result must stay bitwise the graph-parallel evaluation without the collective (and within
    float32 rounding of the ordinary one, test_world_size_one_matches_the_ordinary_path) -- what this proves is that RCCL
    accepts and orders work on memory it did not allocate, between two engine segments, on real hardware."""
    mgr = mp.get_context("spawn").Manager()   # (never FORK a process that has initialised the GPU: the child dies with "Memory in use")
    out = mgr.dict()
    mp.spawn(_nccl_one_rank, args=(_port(), out), nprocs=1, join=True)
    assert out["backend"] == "nccl" and out["n_exchanges"] == 10 and out["bitwise"] is True and out["close"] is True


def _gsm_run(group_world, gp_singles, out_key, out, warm=False):
    """The SPMD growing-string driver on a fully grown 6-image string with the reference's default climbing phase forced on (climb +
    climb_lanczos, path_opt.py:179-182): every cycle = one sharded batched evaluation + a Lanczos recursion of single-image gradients."""
    from pdb2reaction_amd.engine import Engine
    from pdb2reaction_amd.gsm import GrowingStringDriver
    from pdb2reaction_amd.parallel import EngineStringEvaluator

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    n_atoms, k = 60, 6
    z, imgs, frozen = synth.make_images(n_atoms, k, seed=4)
    eng = Engine(0)
    eng.load_weights(W.make_synthetic_weights(0))
    eng.set_system(z)
    ev = EngineStringEvaluator(eng, n_atoms, dev, frozen=frozen, gp_singles=gp_singles)
    x0 = (imgs * 1.8897261259077822).reshape(k, -1)
    elem = [synth.SYMBOLS[int(q)] for q in z]
    drv = GrowingStringDriver(elem, x0[0], x0[-1], evaluate_device=ev, device=dev, images=x0,
                              gs_kw={"max_nodes": k - 2, "fix_first": False, "fix_last": False, "climb": True, "climb_rms": 1e9, "climb_lanczos": True,
                                     "climb_lanczos_rms": 1e9, "climb_lanczos_warm_start": warm},
                              stopt_kw={"max_cycles": 5, "thresh": "gau_vtight", "max_step": 0.05, "print_every": 10 ** 9})
    res = drv.run()
    out[out_key] = (res.coords, res.energies, drv.lanczos_evals, drv.lanczos_calls, ev.gp_single_calls, group_world)
    eng.close()


def _gsm_worker(rank, world, port, out, warm):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _gsm_run(world, True, rank, out, warm)
    finally:
        dist.destroy_process_group()


def _gsm_single(rank, out, warm):
    _gsm_run(1, True, "single", out, warm)


@pytest.mark.parametrize("warm", [False, True])
def test_spmd_driver_evaluates_the_lanczos_probes_graph_parallel(warm):
    """in the sharded run the serial single-image probes of the climbing image's Lanczos recursion no longer run on one
    rank while the others wait -- a one-image batch goes through the graph-parallel evaluator over ALL ranks (row a12: built for K < G).
    Two gloo ranks on the one GPU: both ranks end with the same string bit for bit, every probe was a graph-parallel evaluation, and the
    string agrees with the single-process run to the float32 summation-order differences of the partial edge sums -- tightly with converged
    (cold) recursions; with the default warm-started two-step recursions the 2e-6 eV/A differences of the probes pass through a finite-difference
    Hessian (dx = 5e-3) and five climbing steps, so only the looser bound is asserted there."""
    mgr = mp.get_context("spawn").Manager()   # (never FORK a process that has initialised the GPU)
    out = mgr.dict()
    mp.spawn(_gsm_single, args=(out, warm), nprocs=1, join=True)
    mp.spawn(_gsm_worker, args=(2, _port(), out, warm), nprocs=2, join=True)
    c1, e1, evals1, calls1, gp1, _ = out["single"]
    assert evals1 > 0 and calls1 >= 3 and gp1 == 0                                     # one process: no graph-parallel evaluation
    for r in (0, 1):
        c, e, evals, calls, gp_calls, world = out[r]
        assert world == 2 and gp_calls == evals > 0 and calls == calls1               # every probe went through the graph-parallel evaluator
        tol_c, tol_e = (5e-3, 1e-3) if warm else (2e-3, 2e-5)
        assert np.abs(c - c1).max() < tol_c and np.abs(e - e1).max() < tol_e, (np.abs(c - c1).max(), np.abs(e - e1).max())
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])  # the replicated update stays replicated
