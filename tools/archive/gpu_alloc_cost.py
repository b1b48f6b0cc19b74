"""Cost of hipMalloc / hipFree by size on this driver (dev): the workspace policy (how large a chunk to allocate) depends on it."""
import time
import torch

torch.zeros(1, device="cuda")
for rep in range(2):
    for gib in (4, 16, 64, 128, 200):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x = torch.empty(gib << 30, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        x[:: 1 << 21].fill_(1)                       # touch every 2 MiB page
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        del x
        torch.cuda.empty_cache()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        y = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")      # does the NEXT allocation pay for the release?
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        del y
        torch.cuda.empty_cache()
        print(f"{gib:4d} GiB: hipMalloc {1e3 * (t1 - t0):8.1f} ms  first touch {1e3 * (t2 - t1):7.1f} ms  hipFree {1e3 * (t3 - t2):8.1f} ms  next 1 GiB malloc {1e3 * (t4 - t3):7.1f} ms", flush=True)
