"""Host-side plumbing of the drivers around the engine (no GPU)."""
import os

import numpy as np
import pytest


def test_driver_host_math_runs_on_few_blas_threads(monkeypatch):
    """pdb2reaction_amd._host: inside a driver the BLAS / OpenMP pools are limited to UMX_HOST_THREADS (default 1) -- spinning OpenBLAS
    workers otherwise burn a container's CPU quota and the throttled process starves the GPU -- and restored afterwards."""
    threadpoolctl = pytest.importorskip("threadpoolctl")
    from pdb2reaction_amd import _host

    assert 1 <= _host.usable_cores() <= (os.cpu_count() or 1)
    a = np.ones((64, 64))
    a @ a                                                       # make sure the BLAS library is loaded
    before = [p["num_threads"] for p in threadpoolctl.threadpool_info()]
    seen = {}

    @_host.with_small_host_math
    def driver():
        seen["inside"] = [p["num_threads"] for p in threadpoolctl.threadpool_info()]
        return 7

    assert driver() == 7 and seen["inside"] and all(n == 1 for n in seen["inside"])
    assert [p["num_threads"] for p in threadpoolctl.threadpool_info()] == before
    monkeypatch.setenv("UMX_HOST_THREADS", "0")                 # 0: hands off
    driver()
    assert seen["inside"] == before
    monkeypatch.setenv("UMX_HOST_THREADS", "2")
    driver()
    assert all(n == min(2, b) for n, b in zip(seen["inside"], before))
    from pdb2reaction_amd import gsm, lbfgs, hessian, prestep
    for fn in (gsm.GrowingStringDriver.run, lbfgs.BatchedLBFGS.run, hessian.fd_hessian, prestep.scan_toward_target, prestep.align_and_refine_sequence):
        assert hasattr(fn, "__wrapped__")                      # every driver entry point is covered


def test_pools_are_capped_to_the_usable_cores_once(monkeypatch):
    """cap_pools_to_usable_cores: pools larger than what affinity / cgroup quota allow are cut to that number, once, for good; smaller
    pools are left alone; UMX_HOST_THREADS=0 switches it off."""
    threadpoolctl = pytest.importorskip("threadpoolctl")
    from pdb2reaction_amd import _host

    a = np.ones((32, 32))
    a @ a
    monkeypatch.setattr(_host, "_CAPPED", None)
    monkeypatch.setenv("UMX_HOST_THREADS", "0")
    assert _host.cap_pools_to_usable_cores() == 0
    monkeypatch.delenv("UMX_HOST_THREADS")
    before = [p["num_threads"] for p in threadpoolctl.threadpool_info()]
    monkeypatch.setattr(_host, "usable_cores", lambda: 10 ** 6)
    assert _host.cap_pools_to_usable_cores() == 0                    # nothing exceeds the allowance: hands off
    assert [p["num_threads"] for p in threadpoolctl.threadpool_info()] == before
    if max(before) > 1:
        monkeypatch.setattr(_host, "usable_cores", lambda: 1)
        try:
            assert _host.cap_pools_to_usable_cores() == 1
            assert all(p["num_threads"] == 1 for p in threadpoolctl.threadpool_info())
            assert _host.cap_pools_to_usable_cores() == 0            # once per process
        finally:
            _host._CAPPED.restore_original_limits()
        assert [p["num_threads"] for p in threadpoolctl.threadpool_info()] == before
