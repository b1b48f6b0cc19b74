"""RCCL API rehearsal with ONE rank (this pool has one-GPU boxes): the collectives bench.py / parallel.py issue, same dtypes and shapes."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
send = torch.randn(2, 6001, dtype=torch.float64, device=dev); recv = torch.empty(2, 6001, dtype=torch.float64, device=dev)
dist.all_gather_into_tensor(recv, send); assert torch.equal(recv, send)
t = torch.tensor([1.5, 2.0], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.all_reduce(t, op=dist.ReduceOp.SUM)
f = torch.randn(2000, 1152, dtype=torch.float32, device=dev); g = f.clone(); dist.all_reduce(g); assert torch.equal(f, g)
dist.barrier(); torch.cuda.synchronize()
from pdb2reaction_amd.parallel import ShardedImageEvaluator
ev = ShardedImageEvaluator(lambda c: (c.sum((1, 2)), -c), 4, 10, dev)
e, fo = ev(torch.ones(4, 10, 3, dtype=torch.float64, device=dev)); assert e.shape == (4,) and fo.shape == (4, 10, 3)
dist.destroy_process_group(); print("RCCL single-rank rehearsal OK")
