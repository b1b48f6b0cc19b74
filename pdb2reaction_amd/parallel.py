"""Multi-GPU evaluation of a reaction string: one process per GPU.

1. IMAGE SHARDING (the default): images of one string split into contiguous blocks, ONE all-gather of
   per-image [E | F(3N)] (float64) per string iteration -- :class:`ShardedImageEvaluator`.
2. GRAPH-PARALLEL SINGLE IMAGE (rows a12 / f4; for fewer images than GPUs, e.g. one 20 000-atom structure): the
   reference's ``workers > 1`` semantics (``uma_pysis.py:220-242``) -- the graph of ONE image partitioned by target node over
   the ranks, node-level buffers all-reduced at the engine's exchange points -- :class:`GraphParallelEvaluator`.

The reference evaluates the images serially through one shared calculator (``path_opt.py:949-954``,
``GS_KW["scheduler"] = None`` at ``path_opt.py:184``); its ``workers>1`` knob is graph-parallelism
inside one image (``uma_pysis.py:220-242``).  Images are independent, so the data path needs no
collective; the only exchange is the result gather in front of the (replicated, deterministic)
string update.  Payload at 2000 atoms x 16 images: 768 KB -> latency bound, a single
``all_gather_into_tensor`` over RCCL/xGMI.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_images: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block of images owned by ``rank`` (first ``n_images % world`` ranks get one more)."""
    base, rem = divmod(n_images, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ShardedImageEvaluator:
    """Evaluate a string's images on their owner ranks and return the gathered (E, F) on every rank.

    ``evaluate_local(coords_local[k,N,3]) -> (E[k] float64, F[k,N,3])`` runs on this rank's device.

    Every rank ALWAYS completes the collective.  A rank whose ``evaluate_local`` raises (a non-finite coordinate in its shard, the
    sticky ``UMX_ERR_RANGE`` of the asynchronous device-pointer entry, a HIP error) fills its slot with NaN, sets the STATUS column
    of the slot and enters the all-gather like everybody else; afterwards every rank raises -- the failing rank its own exception,
    the peers a ``RuntimeError`` naming the failing ranks -- instead of the peers blocking in the collective until the backend's
    timeout.
    """

    def __init__(self, evaluate_local: Callable, n_images: int, n_atoms: int, device: torch.device,
                 group: Optional["dist.ProcessGroup"] = None, engine=None, check: str = "sync", force_collective: bool = False):
        """force_collective: issue the all-gather even in a one-rank group (it is the identity there) -- exercises the RCCL call on
        the evaluator's own buffers where only one GPU is available (tests).

        engine: the ``Engine`` behind ``evaluate_local`` when that goes through the asynchronous device-pointer entry
        (``umx_energy_forces_dev`` cannot refuse a non-finite energy itself).  With it, the gathered energies are checked:
        they are the same on every rank, so every rank takes the same decision without a further collective.

        check = "sync" (default): looked at right after the gather (one scalar read per call = one host synchronisation); a non-finite
        energy widens the engine to bf16 forward planes on all ranks (``Engine.widen``, fast split-f16 mode only) and evaluates again,
        or raises.  check = "deferred": no host synchronisation of its own -- the flag of call i is computed on the device, copied to
        pinned host memory on the same stream, and read at the START of call i+1 behind the engine's own per-call synchronisation
        (the D2H read of the edge counts), or by ``flush()``; a non-finite energy then raises on every rank (the caller has already
        consumed the bad energies, so there is nothing to repeat).  For loops that never leave the device (bench.py, device string
        updates): at the 52-ms 2-image shard of the 8-GPU run a forced synchronisation per iteration is no longer free."""
        if check not in ("sync", "deferred"):
            raise ValueError(f"check must be 'sync' or 'deferred', got {check!r}")
        self.evaluate_local = evaluate_local
        self.engine = engine
        self.check = check
        self.n_images, self.n_atoms, self.device, self.group = n_images, n_atoms, device, group
        initialised = dist.is_available() and dist.is_initialized()
        self.distributed = initialised and (dist.get_world_size(group) > 1 or bool(force_collective))
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.rank = dist.get_rank(group) if self.distributed else 0
        self.lo, self.hi = shard_bounds(n_images, self.world, self.rank)
        self.width = 2 + 3 * n_atoms                             # [E | status | F(3N)]
        # equal-size slots so a single all_gather_into_tensor works for ragged shards
        self.slot = -(-n_images // self.world)
        self._send = torch.zeros(self.slot, self.width, dtype=torch.float64, device=device)
        self._recv = torch.zeros(self.world * self.slot, self.width, dtype=torch.float64, device=device)
        # gloo has no device collectives: stage through the host (rehearsal of the N>1 path on boxes without RCCL peers)
        self._stage_cpu = self.distributed and device.type != "cpu" and dist.get_backend(group) == "gloo"
        self._calls = 0
        self._pending = None                                     # (call number, event behind the flag copy) of the deferred check
        if check == "deferred":
            pin = device.type == "cuda"
            self._flag_host = torch.zeros(2, dtype=torch.float64, pin_memory=pin)
            self._flag_dev = torch.zeros(2, dtype=torch.float64, device=device)

    def _raise_status(self, failed_ranks, own_exc):
        if own_exc is not None:
            raise own_exc
        raise RuntimeError(f"sharded string evaluation: evaluate_local raised on rank(s) {failed_ranks} (this is rank {self.rank}); "
                           "their error text is on those ranks")

    def _take_deferred(self):
        """Look at the flags of the previous call (already in pinned host memory: the engine synchronised the stream they were copied on)."""
        if self._pending is None:
            return
        n, ev = self._pending
        self._pending = None
        if ev is not None and not ev.query():                    # normally complete: the engine call of this iteration began with a
            ev.synchronize()                                     # stream synchronisation (a rank without images, or a caller without an engine, waits here)
        bad_e, failed = float(self._flag_host[0]), float(self._flag_host[1])
        if failed > 0.0:
            raise RuntimeError(f"sharded string evaluation: evaluate_local raised on a peer rank in call {n}")
        if bad_e > 0.0:
            if self.engine is not None:
                self.engine.take_range_error()
            raise RuntimeError(f"non-finite energy in call {n} of a sharded string evaluation (precision mode "
                               f"{self.engine.precision_mode() if self.engine is not None else '?'}; found by the deferred check, the "
                               "caller has consumed it): non-finite coordinates, or an activation beyond the operand range of the fast split-f16 mode")

    def flush(self):
        """Deferred mode: synchronise and check the last call's flags (call once after a loop)."""
        if self.check == "deferred" and self._pending is not None:
            if self.device.type == "cuda":
                torch.cuda.current_stream(self.device).synchronize()
            self._take_deferred()

    def __call__(self, coords: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        k = self.hi - self.lo
        own_exc = None
        self._calls += 1
        if k > 0:
            try:
                e, f = self.evaluate_local(coords[self.lo:self.hi])
                # (the engine call above began with its own stream synchronisation: the previous call's flags have landed)
                self._send[:k, 0] = e.to(torch.float64)
                self._send[:k, 1] = 0.0
                self._send[:k, 2:] = f.reshape(k, -1).to(torch.float64)
            except Exception as exc:      # noqa: BLE001 -- whatever it is, the peers are waiting in the collective
                own_exc = exc
                self._send[:k] = float("nan")
                self._send[:k, 1] = 1.0
        if self.check == "deferred" and own_exc is None:
            self._take_deferred()
        if not self.distributed:
            out = self._send[: self.n_images]
        else:
            if self._stage_cpu:
                recv = torch.empty(self._recv.shape, dtype=torch.float64)
                dist.all_gather_into_tensor(recv, self._send.cpu(), group=self.group)
                self._recv.copy_(recv)
            else:
                dist.all_gather_into_tensor(self._recv, self._send, group=self.group)
            rows = []
            for r in range(self.world):
                lo, hi = shard_bounds(self.n_images, self.world, r)
                rows.append(self._recv[r * self.slot: r * self.slot + (hi - lo)])
            out = torch.cat(rows, dim=0)
        if own_exc is not None:                                  # the collective is complete: now this rank may leave
            self._raise_status([self.rank], own_exc)
        if self.check == "deferred":
            self._flag_dev[0] = (~torch.isfinite(out[:, 0])).any().to(torch.float64)
            self._flag_dev[1] = (out[:, 1] > 0).any().to(torch.float64)
            self._flag_host.copy_(self._flag_dev, non_blocking=True)
            ev = None
            if self.device.type == "cuda":
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
            self._pending = (self._calls, ev)
            return out[:, 0].clone(), out[:, 2:].reshape(self.n_images, self.n_atoms, 3).clone()
        status = out[:, 1]
        if bool((status > 0).any()):                             # identical on every rank
            failed = sorted({r for r in range(self.world)
                             if bool((status[shard_bounds(self.n_images, self.world, r)[0]: shard_bounds(self.n_images, self.world, r)[1]] > 0).any())})
            self._raise_status(failed, None)
        if self.engine is not None and not bool(torch.isfinite(out[:, 0]).all()):
            self.engine.take_range_error()                      # collect (clear) the sticky flag of this rank's engine, if it was the one
            if self.engine.widen("non-finite energy in a sharded string evaluation"):
                return self(coords)                             # every rank is here with the same gathered energies: all widen, all repeat
            bad = torch.nonzero(~torch.isfinite(out[:, 0])).flatten().tolist()
            raise RuntimeError(f"non-finite energy for image(s) {bad} (precision mode {self.engine.precision_mode()}): non-finite "
                               "coordinates, or an overflow that wider forward planes cannot cure")
        return out[:, 0].clone(), out[:, 2:].reshape(self.n_images, self.n_atoms, 3).clone()


class _DeviceView:
    """A float32 device buffer owned by the engine, exposed through ``__cuda_array_interface__`` so that torch can wrap it
    without a copy (the all-reduce must happen IN PLACE in the engine's workspace)."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


class GraphParallelEvaluator:
    """E + F of ONE image with its graph partitioned over the ranks of ``group`` (reference ``workers > 1``).

    Every rank calls ``evaluator(pos)`` with the same float32 positions ``(N, 3)`` on its device.  Rank r owns the target
    nodes ``shard_bounds(N, world, r)`` and builds / processes only their incoming edges; the engine pauses at 10 exchange
    points (edge-degree aggregate, one node aggregate and one node gradient per layer, forces) whose buffers are summed
    over the ranks in place -- RCCL all-reduce over xGMI on the caller's stream (``backend="nccl"``), or a host-staged
    all-reduce when the group is gloo (rehearsal of the N > 1 path with several ranks on one GPU).  Energies are complete
    on every rank (node-level work is replicated), forces after the last all-reduce.  Payload per evaluation:
    9 x N x 1152 x 4 B + N x 12 B (92 MB per all-reduce at 20 000 atoms).
    """

    def __init__(self, engine, n_atoms: int, device: torch.device, group: Optional["dist.ProcessGroup"] = None,
                 force_collective: bool = False):
        """force_collective: issue the all-reduces even in a one-rank group (they are the identity there) -- exercises the in-place
        RCCL call on the engine's own workspace memory where only one GPU is available."""
        self.engine, self.n_atoms, self.device, self.group = engine, int(n_atoms), device, group
        initialised = dist.is_available() and dist.is_initialized()
        self.distributed = initialised and (dist.get_world_size(group) > 1 or bool(force_collective))
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.rank = dist.get_rank(group) if self.distributed else 0
        self.lo, self.hi = shard_bounds(self.n_atoms, self.world, self.rank)
        self._stage_cpu = self.distributed and dist.get_backend(group) == "gloo"
        self.n_exchanges = 0
        self._e = torch.zeros(1, dtype=torch.float64, device=device)
        self._f = torch.zeros(self.n_atoms, 3, dtype=torch.float32, device=device)

    def __call__(self, pos_ang: torch.Tensor, _retry: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
        pos = pos_ang.to(device=self.device, dtype=torch.float32).contiguous()
        if pos.shape != (self.n_atoms, 3):
            raise ValueError(f"positions must be ({self.n_atoms}, 3), got {tuple(pos.shape)}")
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self.engine.gp_begin(pos.data_ptr(), self.lo, self.hi, self._e.data_ptr(), self._f.data_ptr(), stream)
        self.n_exchanges = 0
        while True:
            ptr, count, done = self.engine.gp_step()
            if done:
                break
            self.n_exchanges += 1
            if not self.distributed:
                continue
            buf = torch.as_tensor(_DeviceView(ptr, count), device=self.device)
            if self._stage_cpu:
                host = buf.cpu()
                dist.all_reduce(host, group=self.group)
                buf.copy_(host)
            else:
                dist.all_reduce(buf, group=self.group)
        if not bool(torch.isfinite(self._e).all()):
            # energies are complete on every rank (node-level work is replicated): every rank sees the same value and takes the same path
            self.engine.take_range_error()
            if not _retry and self.engine.widen("non-finite energy in a graph-parallel evaluation"):
                return self(pos_ang, _retry=True)
            raise RuntimeError(f"non-finite energy in graph-parallel mode (precision mode {self.engine.precision_mode()}): non-finite "
                               "coordinates, or an overflow that wider forward planes cannot cure")
        return self._e.clone(), self._f.clone()


class ShardedStringEvaluator:
    """``evaluator(x[k, 3N]) -> (E[k], F[k, 3N])`` for ANY batch size k (a growing string changes it): the k images are sharded contiguously
    over the ranks of `group` through a :class:`ShardedImageEvaluator` cached per k; ``local_fn(x_local[kl, N, 3]) -> (E[kl], F[kl, N, 3])``
    evaluates this rank's block on `device`.  The device contract of ``gsm.GrowingStringDriver(evaluate_device=...)``."""

    def __init__(self, local_fn: Callable, n_atoms: int, device: torch.device, group: Optional["dist.ProcessGroup"] = None, engine=None,
                 check: str = "sync", force_collective: bool = False):
        self.local_fn, self.n_atoms, self.device, self.group, self.engine, self.check = local_fn, int(n_atoms), device, group, engine, check
        self.force_collective = bool(force_collective)
        self._ev: dict = {}

    def __call__(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        k = x.shape[0]
        ev = self._ev.get(k)
        if ev is None:
            ev = self._ev[k] = ShardedImageEvaluator(self.local_fn, k, self.n_atoms, self.device, group=self.group, engine=self.engine, check=self.check,
                                                     force_collective=self.force_collective)
        e, f = ev(x.reshape(k, self.n_atoms, 3))
        return e, f.reshape(k, -1)

    def flush(self):
        for ev in self._ev.values():
            ev.flush()


class EngineStringEvaluator(ShardedStringEvaluator):
    """Device evaluator of string images for a device-resident driver (``gsm.GrowingStringDriver(evaluate_device=...)``, bench.py):
    ``evaluator(x[k, 3N] Bohr float64 on the engine's GPU) -> (E[k] Hartree float64, F[k, 3N] Hartree/Bohr float64)`` on the same
    device, exactly what ``uma_pysis.get_forces_batch`` returns per image (``uma_pysis.py:695-706``: float32 Angstrom positions into
    the model, frozen-atom force rows zeroed in eV/A, unit conversion in float64) -- but nothing crosses PCIe: positions are converted
    on the device, the engine runs through its device-pointer entry on torch's current stream, and with more than one rank the k
    images are sharded contiguously over the ranks with ONE all-gather of [E | F] (``ShardedImageEvaluator``; one cached per batch
    size, since a growing string changes k)."""

    def __init__(self, engine, n_atoms: int, device: torch.device, frozen: Sequence[int] = (), group: Optional["dist.ProcessGroup"] = None,
                 check: str = "sync", max_images: int = 0, force_collective: bool = False, gp_singles: bool = True):
        """gp_singles: with more than one rank, a ONE-image batch (the serial probes of the climbing image's Lanczos recursion,
        ``gsm._single_forces``; reference ``GS_KW["climb_lanczos"]``, ``path_opt.py:181-182``) is evaluated GRAPH-PARALLEL over all ranks
        (:class:`GraphParallelEvaluator`: the image's edges partitioned by target node, 10 all-reduces) instead of on rank 0 while the
        others wait in the all-gather -- every rank calls the evaluator with the same geometry anyway (SPMD driver)."""
        from ._calculator_base import BOHR2ANG
        from .hessian import EV_PER_ANG_TO_AU, EV_TO_HARTREE

        super().__init__(self._local, n_atoms, device, group=group, engine=engine, check=check, force_collective=force_collective)
        self._b2a, self._e2h, self._f2au = float(BOHR2ANG), float(EV_TO_HARTREE), float(EV_PER_ANG_TO_AU)
        self._frozen = torch.as_tensor(sorted(set(int(i) for i in frozen)), dtype=torch.long, device=device)
        self._cap = 0
        self._pos32 = self._e = self._f = None
        initialised = dist.is_available() and dist.is_initialized()
        self._world = dist.get_world_size(group) if initialised else 1
        self._gp_singles = bool(gp_singles) and self._world > 1
        self._gp: Optional[GraphParallelEvaluator] = None
        self.gp_single_calls = 0
        if max_images:
            self._reserve(int(max_images))
            engine.reserve_images(int(max_images))

    def _reserve(self, k: int):
        if k > self._cap:
            self._pos32 = torch.empty(k, self.n_atoms, 3, dtype=torch.float32, device=self.device)
            self._e = torch.empty(k, dtype=torch.float64, device=self.device)
            self._f = torch.empty(k, self.n_atoms, 3, dtype=torch.float32, device=self.device)
            self._cap = k

    def __call__(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        if self._gp_singles and x.shape[0] == 1:
            if self._gp is None:
                self._gp = GraphParallelEvaluator(self.engine, self.n_atoms, self.device, group=self.group)
            self.flush()                                                             # a deferred flag of the batched calls is read before the mode changes
            e, f = self._gp((x.reshape(self.n_atoms, 3) * self._b2a).to(torch.float32))
            self.gp_single_calls += 1
            f = f.to(torch.float64) * self._f2au
            if self._frozen.numel():
                f[self._frozen, :] = 0.0
            return e.to(torch.float64) * self._e2h, f.reshape(1, -1)
        return super().__call__(x)

    def _local(self, c_bohr: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        kl = c_bohr.shape[0]
        self._reserve(kl)
        self._pos32[:kl].copy_(c_bohr.reshape(kl, self.n_atoms, 3) * self._b2a)      # AtomicData.pos is float32 Angstrom
        # the engine enqueues on torch's current stream: producer (the copy above) and consumers (conversion below, the all-gather)
        # are ordered with it by the stream alone
        self.engine.energy_forces_dev(kl, self._pos32.data_ptr(), self._e.data_ptr(), self._f.data_ptr(),
                                      stream=torch.cuda.current_stream(self.device).cuda_stream)
        f = self._f[:kl].to(torch.float64) * self._f2au
        if self._frozen.numel():
            f[:, self._frozen, :] = 0.0                                              # uma_pysis.py:561-567
        return self._e[:kl] * self._e2h, f
