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

from typing import Callable, Optional, Tuple

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
    """

    def __init__(self, evaluate_local: Callable, n_images: int, n_atoms: int, device: torch.device,
                 group: Optional["dist.ProcessGroup"] = None, engine=None):
        """engine: the ``Engine`` behind ``evaluate_local`` when that goes through the asynchronous device-pointer entry
        (``umx_energy_forces_dev`` cannot refuse a non-finite energy itself).  With it, the gathered energies are checked after
        every call: they are the same on every rank, so every rank takes the same decision -- widen the engine to bf16 forward
        planes (``Engine.widen``) and evaluate again, or raise -- without a further collective."""
        self.evaluate_local = evaluate_local
        self.engine = engine
        self.n_images, self.n_atoms, self.device, self.group = n_images, n_atoms, device, group
        self.distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.rank = dist.get_rank(group) if self.distributed else 0
        self.lo, self.hi = shard_bounds(n_images, self.world, self.rank)
        self.width = 1 + 3 * n_atoms
        # equal-size slots so a single all_gather_into_tensor works for ragged shards
        self.slot = -(-n_images // self.world)
        self._send = torch.zeros(self.slot, self.width, dtype=torch.float64, device=device)
        self._recv = torch.zeros(self.world * self.slot, self.width, dtype=torch.float64, device=device)
        # gloo has no device collectives: stage through the host (rehearsal of the N>1 path on boxes without RCCL peers)
        self._stage_cpu = self.distributed and device.type != "cpu" and dist.get_backend(group) == "gloo"

    def __call__(self, coords: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        k = self.hi - self.lo
        if k > 0:
            e, f = self.evaluate_local(coords[self.lo:self.hi])
            self._send[:k, 0] = e.to(torch.float64)
            self._send[:k, 1:] = f.reshape(k, -1).to(torch.float64)
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
        if self.engine is not None and not bool(torch.isfinite(out[:, 0]).all()):
            self.engine.take_range_error()                      # collect (clear) the sticky flag of this rank's engine, if it was the one
            if self.engine.widen("non-finite energy in a sharded string evaluation"):
                return self(coords)                             # every rank is here with the same gathered energies: all widen, all repeat
            bad = torch.nonzero(~torch.isfinite(out[:, 0])).flatten().tolist()
            raise RuntimeError(f"non-finite energy for image(s) {bad} (precision mode {self.engine.precision_mode()}): non-finite "
                               "coordinates, or an overflow that wider forward planes cannot cure")
        return out[:, 0].clone(), out[:, 1:].reshape(self.n_images, self.n_atoms, 3).clone()


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
