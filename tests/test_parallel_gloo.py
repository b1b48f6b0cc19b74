"""gloo tests (world size 2 and 8) of the image-sharded evaluator (the N>1 path of bench.py), CPU only.
(Managers come from the SPAWN context: a forked child of a process that has initialised the GPU dies with "Memory in use" -- harmless in this
GPU-less file, the same trap if it ever runs inside a process that has touched the GPU)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pdb2reaction_amd.parallel import ShardedImageEvaluator, shard_bounds


def toy(c):
    k = c.shape[0]
    x = c.reshape(k, -1)
    return 0.5 * (x ** 2).sum(1) + x[:, 0], -(x + torch.nn.functional.one_hot(torch.zeros(k, dtype=torch.long), x.shape[1])).reshape(c.shape)


def test_shard_bounds_cover_everything():
    for k in (1, 5, 8, 16, 24):
        for w in (1, 2, 3, 8):
            cuts = [shard_bounds(k, w, r) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == k
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, k, n, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        coords = torch.randn(k, n, 3, dtype=torch.float64, generator=g)
        calls = []

        def local(c):
            calls.append(c.shape[0])
            return toy(c)

        ev = ShardedImageEvaluator(local, k, n, torch.device("cpu"))
        e, f = ev(coords)
        e2, f2 = ev(coords + 1.0)
        out[rank] = (e.numpy(), f.numpy(), e2.numpy(), calls)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("k,world", [(5, 2), (16, 2), (16, 8), (8, 8), (24, 8)])
def test_sharded_equals_single_process(k, world):
    """(16, 8) / (8, 8) / (24, 8): the layouts BASELINE names -- c3 "16 images sharded 2-images/GPU across 8", c5 "8 images one-image-per-GPU",
    c4's 24 DMF images on 8 GPUs -- as eight gloo ranks on the CPU (an 8-rank rehearsal on ONE GPU is not possible: the box admits six GPU
    processes, the launcher included)."""
    n = 7
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, k, n, out), nprocs=world, join=True)
    g = torch.Generator().manual_seed(0)
    coords = torch.randn(k, n, 3, dtype=torch.float64, generator=g)
    e_ref, f_ref = toy(coords)
    for r in range(world):
        e, f, e2, calls = out[r]
        assert np.array_equal(e, e_ref.numpy()) and np.array_equal(f, f_ref.numpy())       # bit-for-bit per image
        assert np.array_equal(e2, toy(coords + 1.0)[0].numpy())
        lo, hi = shard_bounds(k, world, r)
        assert calls == [hi - lo, hi - lo]                                               # each rank evaluated only its shard


def test_single_process_path():
    ev = ShardedImageEvaluator(toy, 3, 4, torch.device("cpu"))
    c = torch.arange(36, dtype=torch.float64).reshape(3, 4, 3)
    e, f = ev(c)
    assert torch.equal(e, toy(c)[0]) and torch.equal(f, toy(c)[1])


def _hess_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pdb2reaction_amd.hessian import fd_hessian

        n = 5
        rng = np.random.default_rng(0)
        m = rng.standard_normal((3 * n, 3 * n))
        a = m @ m.T / (3 * n) + np.eye(3 * n)
        calls = []

        def forces(c):
            calls.append(len(c))
            return (-(c.reshape(len(c), -1) @ a)).reshape(c.shape).astype(np.float32)

        x0 = rng.standard_normal((n, 3))
        h = fd_hessian(forces, x0, [1], device=torch.device("cpu"), double=True, partial=False, batch=4, shard=True)
        n_sharded = sum(calls)
        # default (shard=False): purely local even inside an initialised group -- only rank 0 calls it here, and it must
        # neither hang nor return a partly filled matrix
        h_local = None
        if rank == 0:
            h_local = fd_hessian(forces, x0, [1], device=torch.device("cpu"), double=True, partial=False, batch=4).reshape(3 * n, 3 * n).numpy()
        # ranks that enter the collective with different geometries fail loudly on every rank
        try:
            fd_hessian(forces, x0 + 1e-9 * rank, [1], device=torch.device("cpu"), double=True, partial=False, batch=4, shard=True)
            mismatch = "no error"
        except RuntimeError as exc:
            mismatch = str(exc)
        out[rank] = (h.reshape(3 * n, 3 * n).numpy(), n_sharded, h_local, mismatch)
    finally:
        dist.destroy_process_group()


def test_fd_hessian_columns_are_sharded_over_ranks():
    """c4 'freq Hessian (3N force batches)': active-DOF columns dealt over ranks, one all-reduce assembles the matrix."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_hess_worker, args=(2, port, out), nprocs=2, join=True)
    h0, n0, h_local, msg0 = out[0]
    h1, n1, _, msg1 = out[1]
    assert np.array_equal(h_local, h0)                              # single-rank call == sharded result, bit for bit
    assert "different geometries" in msg0 and "different geometries" in msg1
    assert np.array_equal(h0, h1)                                   # every rank holds the full matrix
    assert n0 == n1 == 12                                           # 12 active DOF / 2 ranks * 2 displacements each
    rng = np.random.default_rng(0)
    m = rng.standard_normal((15, 15))
    a = m @ m.T / 15 + np.eye(15)
    act = [i for i in range(15) if i // 3 != 1]
    assert np.allclose(h0[:, act], a[:, act], atol=2e-3) and np.all(h0[:, 3:6] == 0.0)


# ---- range violations must be agreed on by all ranks ---------------------------------------------------------------
class FakeEngine:
    """The slice of ``engine.Engine`` the collective callers use: ``widened``, ``widen``, ``take_range_error``, ``precision_mode``."""

    def __init__(self, can_widen=True):
        self.widened, self.can_widen, self.widen_calls, self.flag_reads = False, can_widen, 0, 0

    def widen(self, why=""):
        self.widen_calls += 1
        if self.widened or not self.can_widen:
            return False
        self.widened = True
        return True

    def take_range_error(self):
        self.flag_reads += 1
        return False

    def precision_mode(self):
        return "split-bf16" if self.widened else "split-f16"


def _range_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        k, n = 5, 4
        g = torch.Generator().manual_seed(1)
        coords = torch.randn(k, n, 3, dtype=torch.float64, generator=g)
        eng = FakeEngine()
        calls = []

        def local(c):                                         # rank 1's narrow arithmetic overflows on its first image
            calls.append(eng.widened)
            e, f = toy(c)
            if rank == 1 and not eng.widened:
                e = e.clone()
                e[0] = float("nan")
            return e, f

        e, f = ShardedImageEvaluator(local, k, n, torch.device("cpu"), engine=eng)(coords)
        stuck = FakeEngine(can_widen=False)

        def local_bad(c):
            e2, f2 = toy(c)
            if rank == 0:
                e2 = e2.clone()
                e2[1] = float("inf")
            return e2, f2

        try:
            ShardedImageEvaluator(local_bad, k, n, torch.device("cpu"), engine=stuck)(coords)
            msg = "no error"
        except RuntimeError as exc:
            msg = str(exc)
        out[rank] = (e.numpy(), f.numpy(), calls, eng.widened, eng.widen_calls, msg)
    finally:
        dist.destroy_process_group()


def test_sharded_evaluator_widens_on_every_rank_together():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_range_worker, args=(2, port, out), nprocs=2, join=True)
    g = torch.Generator().manual_seed(1)
    coords = torch.randn(5, 4, 3, dtype=torch.float64, generator=g)
    e_ref, f_ref = toy(coords)
    for r in range(2):
        e, f, calls, widened, widen_calls, msg = out[r]
        assert np.array_equal(e, e_ref.numpy()) and np.array_equal(f, f_ref.numpy())
        assert calls == [False, True] and widened and widen_calls == 1          # BOTH ranks repeated their shard in the wide arithmetic
        assert "non-finite energy for image(s) [1]" in msg                      # cannot widen: the same error on every rank


def _hess_range_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pdb2reaction_amd.hessian import fd_hessian

        n = 4
        rng = np.random.default_rng(0)
        m = rng.standard_normal((3 * n, 3 * n))
        a = m @ m.T / (3 * n) + np.eye(3 * n)
        eng = FakeEngine()
        calls = []

        def forces(c):                                        # the narrow arithmetic is measurably different; rank 1 leaves it on its 2nd batch
            calls.append(eng.widened)
            if rank == 1 and len(calls) == 2 and not eng.widened:
                eng.widened = True                            # what Engine._widen does locally on UMX_ERR_RANGE
            scale = 1.0 if eng.widened else 1.01
            return (-(c.reshape(len(c), -1) @ a) * scale).reshape(c.shape).astype(np.float32)

        x0 = rng.standard_normal((n, 3))
        h = fd_hessian(forces, x0, [], device=torch.device("cpu"), double=True, partial=False, batch=4, shard=True, engine=eng)
        wide = FakeEngine()
        wide.widened = True
        h_wide = fd_hessian(lambda c: (-(c.reshape(len(c), -1) @ a)).reshape(c.shape).astype(np.float32), x0, [], device=torch.device("cpu"),
                            double=True, partial=False, batch=4, shard=True, engine=wide)
        out[rank] = (h.numpy(), h_wide.numpy(), calls, eng.widened, wide.widen_calls)
    finally:
        dist.destroy_process_group()


def test_sharded_fd_hessian_never_mixes_two_arithmetics():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_hess_range_worker, args=(2, port, out), nprocs=2, join=True)
    for r in range(2):
        h, h_wide, calls, widened, wide_calls = out[r]
        assert widened and np.array_equal(h, h_wide)                 # every column in the wide arithmetic, on every rank
        assert calls[-3:] == [True, True, True] and len(calls) == 6  # 3 batches (6 DOF / 2 per batch) done twice
        assert wide_calls == 0                                       # already wide everywhere: no repetition, no widen call
    assert np.array_equal(out[0][0], out[1][0])


# ---- a rank-local failure must not leave the peers blocked in the collective ----------------------------------------
def _raise_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import datetime

    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        k, n = 5, 4
        g = torch.Generator().manual_seed(2)
        coords = torch.randn(k, n, 3, dtype=torch.float64, generator=g)
        state = {"n": 0}

        def local(c):                                          # rank 1 fails in its SECOND call only (e.g. the sticky UMX_ERR_RANGE)
            state["n"] += 1
            if rank == 1 and state["n"] == 2:
                raise ValueError("rank-local failure: non-finite position (image 0)")
            return toy(c)

        msgs = []
        for check in ("sync", "deferred"):
            state["n"] = 0
            ev = ShardedImageEvaluator(local, k, n, torch.device("cpu"), check=check)
            e, f = ev(coords)                                   # call 1: fine everywhere
            ok = bool(torch.equal(e, toy(coords)[0]))
            try:
                ev(coords)                                      # call 2: rank 1 raises, rank 0 must come back too
                if check == "deferred":
                    ev(coords)                                  # ... at the latest at the start of the next call
                msgs.append((ok, "no error"))
            except Exception as exc:  # noqa: BLE001
                msgs.append((ok, f"{type(exc).__name__}: {exc}"))
        # a collective after the failures still works: nobody is stuck in, or has skipped, an all-gather
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)
        out[rank] = (msgs, float(t[0]))
    finally:
        dist.destroy_process_group()


def test_rank_local_failure_completes_the_collective_and_raises_everywhere():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_raise_worker, args=(2, port, out), nprocs=2, join=True)
    (m0, t0), (m1, t1) = out[0], out[1]
    assert t0 == t1 == 3.0
    for ok, _ in m0 + m1:
        assert ok
    # sync mode: the failing rank re-raises its own exception, the peer names the failing rank -- in the same call
    assert m1[0][1].startswith("ValueError: rank-local failure") and "raised on rank(s) [1]" in m0[0][1] and "this is rank 0" in m0[0][1]
    # deferred mode: same for the failing rank; the peer learns it at the start of its next call
    assert m1[1][1].startswith("ValueError: rank-local failure") and "raised on a peer rank in call 2" in m0[1][1]


def test_deferred_check_reports_a_non_finite_energy_one_call_later():
    eng = FakeEngine()
    state = {"bad": False}

    def local(c):
        e, f = toy(c)
        if state["bad"]:
            e = e.clone()
            e[1] = float("nan")
        return e, f

    c = torch.arange(36, dtype=torch.float64).reshape(3, 4, 3)
    ev = ShardedImageEvaluator(local, 3, 4, torch.device("cpu"), engine=eng, check="deferred")
    e, f = ev(c)
    assert torch.equal(e, toy(c)[0]) and torch.equal(f, toy(c)[1])
    ev.flush()                                                   # nothing wrong so far
    state["bad"] = True
    e, _ = ev(c)                                                 # returned as computed: no host look in deferred mode
    assert not torch.isfinite(e).all() and eng.widen_calls == 0
    state["bad"] = False
    with pytest.raises(RuntimeError, match="non-finite energy in call 2"):
        ev(c)
    assert eng.flag_reads == 1
    e, _ = ev(c)                                                 # usable again afterwards
    assert torch.isfinite(e).all()
    state["bad"] = True
    ev(c)
    with pytest.raises(RuntimeError, match="deferred check"):
        ev.flush()
    with pytest.raises(ValueError, match="check must be"):
        ShardedImageEvaluator(local, 3, 4, torch.device("cpu"), check="never")


def test_single_rank_fd_hessian_never_mixes_two_arithmetics():
    """the repair of a Hessian whose engine widened half-way applies to the default single-rank case as well."""
    from pdb2reaction_amd.hessian import fd_hessian

    n = 4
    rng = np.random.default_rng(0)
    m = rng.standard_normal((3 * n, 3 * n))
    a = m @ m.T / (3 * n) + np.eye(3 * n)
    eng = FakeEngine()
    calls = []

    def forces(c):
        calls.append(eng.widened)
        if len(calls) == 3 and not eng.widened:
            eng.widened = True
        scale = 1.0 if eng.widened else 1.01
        return (-(c.reshape(len(c), -1) @ a) * scale).reshape(c.shape).astype(np.float32)

    x0 = rng.standard_normal((n, 3))
    h = fd_hessian(forces, x0, [], device=torch.device("cpu"), double=True, partial=False, batch=4, engine=eng)
    wide = FakeEngine()
    wide.widened = True
    h_wide = fd_hessian(lambda c: (-(c.reshape(len(c), -1) @ a)).reshape(c.shape).astype(np.float32), x0, [], device=torch.device("cpu"),
                        double=True, partial=False, batch=4, engine=wide)
    assert torch.equal(h, h_wide) and len(calls) == 12 and calls[6:] == [True] * 6


# ---- round 4: the REAL driver (gsm.GrowingStringDriver) through the sharded evaluator, two ranks ---------------------------------
def _mb_energy_forces(q):
    """Mueller-Brown surface in x, y (+ z^2 / 2), scaled to a Hartree-like range: q (k, 1, 3) -> (E (k,), F (k, 1, 3)), torch float64."""
    A = torch.tensor([-200.0, -100.0, -170.0, 15.0], dtype=torch.float64)
    a = torch.tensor([-1.0, -1.0, -6.5, 0.7], dtype=torch.float64)
    b = torch.tensor([0.0, 0.0, 11.0, 0.6], dtype=torch.float64)
    c = torch.tensor([-10.0, -10.0, -6.5, 0.7], dtype=torch.float64)
    x0 = torch.tensor([1.0, 0.0, -0.5, -1.0], dtype=torch.float64)
    y0 = torch.tensor([0.0, 0.5, 1.5, 1.0], dtype=torch.float64)
    p = q.reshape(-1, 3)
    dx, dy = p[:, 0:1] - x0, p[:, 1:2] - y0
    t = A * torch.exp(a * dx ** 2 + b * dx * dy + c * dy ** 2)
    e = 1e-3 * t.sum(1) + 0.5 * p[:, 2] ** 2
    fx = -1e-3 * (t * (2 * a * dx + b * dy)).sum(1)
    fy = -1e-3 * (t * (b * dx + 2 * c * dy)).sum(1)
    return e, torch.stack([fx, fy, -p[:, 2]], dim=1).reshape(q.shape)


def _gsm_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pdb2reaction_amd.gsm import GrowingStringDriver
        from pdb2reaction_amd.parallel import ShardedStringEvaluator

        calls = []

        def local(c):
            calls.append(int(c.shape[0]))
            return _mb_energy_forces(c)

        ev = ShardedStringEvaluator(local, 1, torch.device("cpu"))
        drv = GrowingStringDriver(["X"], np.array([-0.558224, 1.441726, 0.0]), np.array([0.623499, 0.028038, 0.0]), evaluate_device=ev, device=torch.device("cpu"),
                                  gs_kw={"max_nodes": 9, "perp_thresh": 2e-2, "climb_rms": 5e-3}, stopt_kw={"thresh": "gau", "max_step": 0.05, "max_cycles": 400})
        res = drv.run()
        out[rank] = (res.coords, res.energies, res.converged, res.cycles, res.force_evaluations, calls, drv.lanczos_evals)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_growing_string_driver_runs_spmd_over_the_ranks(world):
    """SURVEY.md 8e with the real driver (two ranks, and the eight of the BASELINE node with ragged shards of the 11-image string): all ranks run ``gsm.GrowingStringDriver`` (device-resident form, CPU tensors here) on the same
    string; every batched evaluation is sharded over the ranks (each evaluates its block only) and gathered; the replicated update keeps
    the ranks bit-identical, and the run equals the single-process run."""
    from pdb2reaction_amd.gsm import GrowingStringDriver

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_gsm_worker, args=(world, port, out), nprocs=world, join=True)
    single = GrowingStringDriver(["X"], np.array([-0.558224, 1.441726, 0.0]), np.array([0.623499, 0.028038, 0.0]),
                                 evaluate_device=lambda x: tuple(t.reshape(x.shape[0], -1) if t.dim() > 1 else t for t in _mb_energy_forces(x.reshape(-1, 1, 3))),
                                 device=torch.device("cpu"), gs_kw={"max_nodes": 9, "perp_thresh": 2e-2, "climb_rms": 5e-3},
                                 stopt_kw={"thresh": "gau", "max_step": 0.05, "max_cycles": 400}).run()
    c0, e0, conv0, cyc0, nev0, calls0, lz0 = out[0]
    assert conv0 and cyc0 == single.cycles and nev0 == single.force_evaluations and lz0 > 0
    assert np.array_equal(c0, single.coords) and np.array_equal(e0, single.energies)  # sharding changes nothing
    for r in range(1, world):
        c1, e1, conv1, cyc1, nev1, calls1, lz1 = out[r]
        assert conv1 and cyc1 == cyc0 and nev1 == nev0 and lz1 == lz0
        assert np.array_equal(c0, c1) and np.array_equal(e0, e1)                     # the ranks never diverge
    # each rank evaluated only its share: together the ranks did the batched evaluations once (+ every Lanczos single on rank 0's block)
    assert sum(sum(out[r][5]) for r in range(world)) == nev0 and all(sum(out[r][5]) < nev0 for r in range(world))
