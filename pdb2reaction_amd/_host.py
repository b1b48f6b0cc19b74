"""Host-side thread hygiene for the drivers around the engine.

The drivers (growing string, batched L-BFGS, staged scan, FD Hessian) do a few milliseconds of numpy work on (K, 3N) arrays between
two engine calls.  numpy's OpenBLAS starts one worker per core of the MACHINE (64 on the 256-core hosts of the MI355X boxes) and lets
them spin after every matmul; inside a container with a CPU quota (16 cores on a one-GPU box) those spinning workers burn the cgroup's
quota and the kernel throttles the whole process for the rest of the 100 ms scheduling period -- including the thread that feeds the
GPU.  Measured on a 500-atom, 10-image string: 97 ms per batched E+F call instead of 57 ms, with the GPU idle for the difference
(tools/archive/gpu_batch_latency2.py).  The arrays are far too small to profit from BLAS threads, so the drivers run their host math on
``UMX_HOST_THREADS`` (default 1) threads via threadpoolctl; the limit is lifted again when the driver returns.
"""
from __future__ import annotations

import contextlib
import functools
import os


def usable_cores() -> int:
    """Host cores this process may actually use: CPU affinity and the cgroup-v2 quota, not the machine total."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


@contextlib.contextmanager
def small_host_math(threads=None):
    """Run the enclosed host-side numpy / scipy code on a few BLAS / OpenMP threads (see the module docstring)."""
    n = int(threads if threads is not None else os.environ.get("UMX_HOST_THREADS", "1"))
    if n <= 0:                                   # 0: leave the pools alone
        yield
        return
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:                          # optional dependency: without it the pools stay as they are
        yield
        return
    with threadpool_limits(limits=min(n, usable_cores())):
        yield


_CAPPED = None


def cap_pools_to_usable_cores() -> int:
    """Once per process: BLAS / OpenMP pools that are larger than the cores this process may use are cut down to that number, for good.
    A pool sized for the whole machine inside a CPU-quota container is the mis-configuration described in the module docstring; a
    limit <= the quota measured 56 ms per call where 32 and 64 threads gave 95 (``tools/archive/gpu_batch_latency2.py``).  Called when a
    calculator creates its engine, so that callers this package does not control (an external optimiser stepping through
    ``get_forces``) are covered too.  ``UMX_HOST_THREADS=0`` disables it.  Returns the cap applied (0: nothing done)."""
    global _CAPPED
    if _CAPPED is not None or os.environ.get("UMX_HOST_THREADS", "1") == "0":
        return 0
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
    except ImportError:
        return 0
    cores = usable_cores()
    if not any(p.get("num_threads", 0) > cores for p in threadpool_info()):
        return 0
    _CAPPED = threadpool_limits(limits=cores)    # kept alive: the limits stay until the process ends
    return cores


def with_small_host_math(fn):
    """Decorator form of :func:`small_host_math` for a driver's entry point."""
    @functools.wraps(fn)
    def wrapper(*a, **kw):
        with small_host_math():
            return fn(*a, **kw)
    return wrapper
