"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same bench command) into
profiles/rNN_pmc_hbm_traffic.json: HBM bytes per kernel launch and per iteration.

    python tools/pmc_summary.py <fetch_dir> <write_dir> <iterations_in_each_run> <out.json> [mode label]

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of wide coalesced reads, WRITE_SIZE is
exact; both are in KiB:  hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pdb2reaction_amd.build import source_digest  # noqa: E402  (the build the passes were run on = the tree this runs in)


def load(d, counter):
    f = (glob.glob(f"{d}/*counter_collection.csv") + glob.glob(f"{d}/*/*counter_collection.csv"))[0]
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter or "umx::" not in r["Kernel_Name"]:
            continue
        a = agg.setdefault(r["Kernel_Name"], [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
iters = int(sys.argv[3])
kernels, total, fam_b, fam_n, f32_b, rad_b = [], 0.0, 0.0, 0, 0.0, 0.0
for name, (n, fkb) in fetch.items():
    wn, wkb = write.get(name, (n, 0.0))
    b = (2.0 * fkb + wkb) * 1024.0
    total += b
    if "umx_gemm_pl" in name or "umx_gemm_q_kernel" in name:
        fam_b += b
        fam_n += n
    elif "umx_gemm_kernel" in name or "k_gemm_f64acc" in name:
        f32_b += b
    elif "k_radial_head" in name or "k_radial_tail" in name:
        rad_b += b
    kernels.append({"kernel": name, "launches_per_iteration": n / iters, "fetch_size_kb_raw_per_launch": fkb / n,
                    "write_size_kb_per_launch": wkb / max(wn, 1), "hbm_bytes_per_launch": b / n})
kernels.sort(key=lambda k: -k["hbm_bytes_per_launch"] * k["launches_per_iteration"])
out = {
    "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline (two separate passes)",
    "correction": "gfx950: FETCH_SIZE reports half of wide coalesced reads -> hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
    "workload": "c3: 2000 atoms x 16 images, 1 GPU, precision mode " + (sys.argv[5] if len(sys.argv) > 5 else "auto (= bf16x3)"),
    "csrc_sha256": source_digest(),
    "hbm_bytes_per_iteration": total / iters,
    "dominant_family": {"kernel": "umx_gemm_q_kernel<*> + umx_gemm_pl16_kernel<*> + umx_gemm_pl_kernel<*>", "launches_per_iteration": fam_n / iters,
                        "hbm_bytes_per_launch_avg": fam_b / max(fam_n, 1), "hbm_bytes_per_iteration": fam_b / iters},
    "fp32_gemm_family_hbm_bytes_per_iteration": f32_b / iters,
    "radial_hbm_bytes_per_iteration": rad_b / iters,
    "kernels": kernels[:24],
}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(f"HBM bytes/iteration {total / iters / 1e12:.3f} TB; GEMM family {fam_b / iters / 1e12:.3f} TB over {fam_n / iters:.0f} launches "
      f"({fam_b / max(fam_n, 1) / 1e9:.2f} GB/launch)")
for k in kernels[:12]:
    print(f"  {k['kernel'][:80]:80s} {k['launches_per_iteration']:6.1f} x {k['hbm_bytes_per_launch'] / 1e9:7.2f} GB")
