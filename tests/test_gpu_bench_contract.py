"""bench.py prints ONE JSON line with the contract's keys (driver contract) -- checked on a tiny workload."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_line_contract():
    out = subprocess.run([sys.executable, "bench.py", "--atoms", "60", "--images", "4", "--steps", "2", "--warmup", "1", "--driver", "gsm", "--gsm-cycles", "4"],
                         cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 1e-6
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "mfma_pipe_util"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma") and 0.0 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # frac is ALGORITHMIC (2*M*N*K per product / time / peak); the executed split products are the separate, larger figure
    assert abs(r["achieved"] * 1e12 - r["algorithmic_flops_per_launch"] / (r["avg_launch_ms"] * 1e-3)) <= 1e-6 * r["achieved"] * 1e12
    assert r["mfma_pipe_util"] == pytest.approx(6.0 * r["frac"], rel=1e-9) and r["peak"] == 2500.0   # bf16x3: six plane products in BOTH passes
    assert r["traffic"] is None and r["hbm_regime"]["traffic_per_step"] is None       # no PMC summary exists for this tiny workload
    hb = r["hbm_regime"]                                                             # edge kernels and fused radial kernels timed apart
    assert hb["bound"] == "hbm" and hb["ms_per_step"] > 0 and hb["radial"]["ms_per_step"] > 0 and hb["radial"]["launches"] == 2 * 10
    assert hb["ms_per_step"] + hb["radial"]["ms_per_step"] + r["ms_per_step"] + r["other_gemm_family"]["ms_per_step"] == pytest.approx(d["ms_per_step"], rel=1e-9)
    # the headline is the like-for-like mode: >= 24-bit products in both passes
    assert d["precision_requested"] == "auto" and d["precision_mode"] == "bf16x3" and d["dtype"] == "bf16x3-split"
    f = d["fp32_mode"]                                                               # the fp32-MFMA figure, same clock
    assert f["value"] > 0 and f["dtype"].startswith("f32") and 0.0 < f["gemm_frac"] < 1.0 and f["steps"] >= 5 and f["warmup"] >= 2 and f["lanes"] == 1
    fm = d["fast_mode"]                                                              # the opt-in narrower mode, reported beside, never as `value`
    assert fm["value"] > 0 and fm["precision_mode"] == "split-f16" and "narrower" in fm["dtype"]
    assert r["lanes"] == 1 and "serial_schedule" not in d and "shard" not in d       # a tiny batch runs on one lane; the shard leg belongs to the headline sizes
    assert d["config"]["name"] == "c3" and "sizes overridden" in d["config"]["workload"]
    g = d["gsm"]                                                                     # the real driver on the fully grown string
    assert "error" not in g, g
    for leg in ("climb_off", "climb_on"):
        assert g[leg]["fully_grown"] and g[leg]["images"] == 4 and g[leg]["cycles_timed"] >= 3 and g[leg]["cycle_ms"] > 0
        assert g[leg]["driver_overhead_ms"] == pytest.approx(g[leg]["cycle_ms"] - g["evaluation_only_ms"], rel=1e-9)
    assert len(d["build_digest"]) == 64
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("reference", "port") and c["value"] and c["value"] > 0 and c["cores"] >= 1
    assert "c1 in full" in c["sample"] and "3 of 12 images" in c["sample"] and c["c3_images_timed"] in (1, 2)


def test_bench_other_configs_and_the_hessian_leg():
    """`--config`: the BASELINE configs the headline line does not carry -- c1 in full, and c4's FD-Hessian leg on a
    reduced size so the test stays short (the loop, its accounting and the extrapolation are what is checked; profiles/r05_bench_c4.json
    has the real size)."""
    out = subprocess.run([sys.executable, "bench.py", "--config", "c1", "--steps", "3", "--warmup", "1", "--driver", "string", "--no-fp32-mode", "--no-fast-mode"],
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["config"]["name"] == "c1" and d["config"]["atoms"] == 50 and d["config"]["images"] == 8 and d["config"]["workload"].startswith("c1: 50-atom")
    assert "cpu_baseline" not in d and d["roofline"]["frac"] > 0
    out = subprocess.run([sys.executable, "bench.py", "--config", "c4", "--atoms", "120", "--images", "6", "--steps", "2", "--warmup", "1", "--driver", "string",
                          "--no-fp32-mode", "--no-fast-mode", "--hessian-sample-atoms", "4"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    h = d["hessian"]
    assert "error" not in h, h
    assert h["columns"] == 12 and h["displaced_geometries"] == 24 and h["engine_calls"] == 1 and h["finite"] and h["columns_per_s"] > 0
    assert h["full_hessian_columns"] == 3 * (120 - 12) and h["extrapolated_full_hessian_s"] == pytest.approx(h["full_hessian_columns"] / h["columns_per_s"], rel=1e-9)
