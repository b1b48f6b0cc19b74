"""Summarise two rocprofv3 PMC passes over `build/overlap_bench pmc 1` (one launch of every large split-bf16 GEMM of a c3 layer at
the real shapes, 8-image chunk = 1.14 M edges) into profiles/rNN_gemm_pmc_counters.json.

    python tools/gemm_pmc_summary.py <pass1_dir> <pass2_dir> <out.json>

pass 1: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
pass 2: SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE
Derived: MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles), cycles = GRBM_GUI_ACTIVE / 8 XCDs (MI355X_MICROARCH.md,
DVFS give-back); in-kernel clock = cycles / duration; SQ_WAIT_* are quad-cycle counters relative to SQ_WAVE_CYCLES."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pdb2reaction_amd.build import source_digest  # noqa: E402

M = 1139068
# last column: planes code -- 24 = forward fp16 form (2 activation x 3 weight planes, 4 products); 2 = reverse bf16 form (2 x 2 planes, 3 products)
SHAPES = [("conv1 m0", 0, 640, 768, 24), ("conv1 m1", 1, 256, 512, 24), ("conv1 m2", 1, 128, 256, 24), ("conv2 m0", 0, 384, 384, 24),
          ("conv2 m1", 1, 256, 256, 24), ("conv2 m2", 1, 128, 128, 24), ("radial fc3", 0, 1536, 128, 24),
          ("conv2^T m0", 0, 384, 384, 2), ("conv2^T m1", 1, 256, 256, 2), ("conv2^T m2", 1, 128, 128, 2),
          ("conv1^T m0", 0, 768, 640, 2), ("conv1^T m1", 1, 512, 256, 2), ("conv1^T m2", 1, 256, 128, 2), ("radial fc3^T", 0, 128, 1536, 2)]


def load(d):
    f = glob.glob(f"{d}/*counter_collection.csv") + glob.glob(f"{d}/*/*counter_collection.csv")
    out = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        k = int(r["Dispatch_Id"])
        e = out.setdefault(k, {"kernel": r["Kernel_Name"], "us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                               "vgpr": int(r["VGPR_Count"]), "lds": int(r["LDS_Block_Size"])})
        e[r["Counter_Name"]] = float(r["Counter_Value"])
    return out


if len(sys.argv) > 4 and sys.argv[4] == "bf16x3":      # the default mode since round 4: three bf16 planes, 6 products in both passes
    SHAPES = [(n, c, nn, kk, 3) for (n, c, nn, kk, _) in SHAPES]
p1, p2 = load(sys.argv[1]), load(sys.argv[2])
rows = []
for (_, a), (_, b), (name, cplx, n, k, p) in zip(p1.items(), p2.items(), SHAPES):
    alg = (8.0 if cplx else 2.0) * M * n * k
    ex = alg * {3: 6, 24: 4, 2: 3}[p]
    cyc = b["GRBM_GUI_ACTIVE"] / 8.0
    wc = a["SQ_WAVE_CYCLES"]
    rows.append({
        "gemm": name, "kernel": a["kernel"].replace("void umx::", "").split("(")[0], "complex": bool(cplx), "N": n, "K": k, "planes": "2x3 fp16" if p == 24 else "2x2 bf16" if p == 2 else "3x3 bf16", "products": {3: 6, 24: 4, 2: 3}[p],
        "vgpr": a["vgpr"], "lds_bytes": a["lds"], "duration_us": a["us"], "algorithmic_tflops": alg / a["us"] / 1e6, "executed_tflops": ex / a["us"] / 1e6,
        "mfma_busy_cycles": a["SQ_VALU_MFMA_BUSY_CYCLES"], "mfma_busy_expected": ex / 32768.0 * 32.0,
        "mfma_util": a["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), "clock_ghz": cyc / b["us"] / 1e3,
        "wait_any_frac": a["SQ_WAIT_ANY"] / wc, "wait_inst_any_frac": a["SQ_WAIT_INST_ANY"] / wc, "active_inst_frac": a["SQ_ACTIVE_INST_ANY"] / wc,
        "wait_inst_lds_frac": a["SQ_WAIT_INST_LDS"] / wc,
        "lds_idx_active_per_cu_cycle": b["SQ_LDS_IDX_ACTIVE"] / (256.0 * cyc), "lds_bank_conflict_frac": b["SQ_LDS_BANK_CONFLICT"] / max(b["SQ_LDS_IDX_ACTIVE"], 1.0),
    })
x3 = len(sys.argv) > 4 and sys.argv[4] == "bf16x3"
out = {"command": f"rocprofv3 --pmc <pass counters> -- build/overlap_bench {'pmc3' if x3 else 'pmc'} 1   (two passes; csrc/overlap_bench.hip)",
       "workload": ("every large GEMM of one c3 layer in the bf16x3 mode: 3 x 3 bf16 planes, 6 products, forward and conv^T reverse on the quad-row layout, fc3^T PL" if x3 else
                    "every large split-bf16 GEMM of one c3 layer, forward (quad-row layout, fp16: 2 activation x 3 weight planes, 4 products) and reverse (PL layout, bf16 2 x 2 planes, 3 products)")
                   + ", M = 1 139 068 edges (8-image chunk), random finite planes",
       "csrc_sha256": source_digest(), "gemms": rows}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(f"{'gemm':14s} {'us':>7s} {'exec TF/s':>9s} {'util':>5s} {'GHz':>5s} {'wait':>5s} {'w_inst':>6s} {'w_lds':>5s} {'lds/cyc':>7s} {'conf':>5s}")
for r in rows:
    print(f"{r['gemm']:14s} {r['duration_us']:7.0f} {r['executed_tflops']:9.0f} {r['mfma_util']:5.2f} {r['clock_ghz']:5.2f} {r['wait_any_frac']:5.2f} "
          f"{r['wait_inst_any_frac']:6.2f} {r['wait_inst_lds_frac']:5.2f} {r['lds_idx_active_per_cu_cycle']:7.3f} {r['lds_bank_conflict_frac']:5.2f}")
